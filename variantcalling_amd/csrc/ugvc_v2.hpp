// v2 fast path: featurize/quantise kernel (K1) + LDS-resident forest kernel (K2).
//
// Tree walks are the dominant cost of the hot path (40 trees x depth 8 = 320 dependent node
// visits per variant).  Global/L2 gathers of 16-byte nodes are bound by the per-CU L1 rate
// (64 B/clk); LDS serves 128-256 B/clk, so v2 keeps one variant-type group's whole forest in
// the LDS of a workgroup.  To make a forest fit (<= ~140 KB):
//   * thresholds are replaced by their RANK among the sorted unique thresholds of that
//     (group, feature): x <= thr_j  <=>  code(x) <= j  with code(x) = #{thr_k < x}  (exact);
//     for XGBoost-style `<`: code(x) = #{thr_k <= x}.  A node is one dword:
//         rank[0:16) | byte offset of the feature's code plane[16:32)
//   * every tree is padded to a complete binary tree (level-order index, no child pointers,
//     padded nodes always go left), so a walk is a fixed `depth` steps with no leaf test;
//   * RF leaf payloads (p_class0, p_class1 as f64 - both needed to reproduce scikit-learn's
//     argmax bit-for-bit) are de-duplicated; a leaf slot holds a u16 index into the unique table.
// A variant's features are quantised once by K1 into <= 96 bits of packed rank codes; K1
// appends a 16-byte record {code0, code1, code2, variant index} to its group's list, K2
// workgroups are split over the groups in proportion to work and walk their list.
#pragma once
#include "ugvc_device.hpp"

namespace ugvc {

constexpr int kJoinArrays = 2 * (UGVC_MAX_TRACKS + 1) + 1;   // runs s/e, tracks s/e, blacklist
constexpr int kStageCap = 320;                                // staged entries per array per block
constexpr int kK2Threads = 1024;
constexpr int kLdsBudget = 150 * 1024;
constexpr int kShards = 256;          // record lists are sharded by (block & 255): spreads the atomics
constexpr int kCounterStride = 16;    // shard counters sit 64 bytes apart
// v3 (kernels_v3.hip)
constexpr int kJoin3 = UGVC_MAX_TRACKS + 2;   // searched arrays: starts of runs + 5 tracks, blacklist keys
constexpr int kPool3 = 2048;                  // dwords of LDS for the staged side-table slices of one tile
constexpr int kThr3 = 3584;                   // floats of LDS for the float-feature thresholds of all groups

struct __attribute__((aligned(16))) FeatDesc {
    uint32_t lut;    // lut_off[0:20) | kind[30:32)   kind: 0 unused, 1 LUT(+search), 2 search only
    uint32_t lut_len;
    uint32_t thr;    // thr_off[0:20) | thr_len[20:32)
    uint32_t pack;   // dword[0:2) | bit_off[2:7) | width[7:11)
};

struct PackedGroupView {
    int ok, kind, T, D;
    float base;
    int n_pairs, n_planes;
    const uint32_t* plane_desc;   // per code plane: dword[0:2) | bit_off[2:7) | width[7:11)
    const uint32_t* nodes;        // T * 2^D, 1-based heap order: rank[0:16) | plane byte offset[16:32)
    const uint16_t* leaf_idx;     // RF: T * 2^D
    const double2* pairs;         // RF: n_pairs
    const float* leaf_f32;        // GBT: T * 2^D
    // RF single-sum layout (forest3_kernel<true>): levels 0..D-2 as a u32 heap of 2^(D-1) dwords per
    // tree, level D-1 as {node word, left payload index | right payload index << 16} pairs, p1[] =
    // class-1 probability of every distinct payload.  Usable when every payload has p0 + p1 == 1 to
    // 1e-9 (then pr1 > pr0 is decided by the class-1 sum alone outside a narrow band around T/2).
    int fast4;
    double band;                  // |class-1 sum - T/2| <= band: both sums are recomputed (exact argmax)
    double inv_T;                 // 1.0 / T (IEEE, from the host): the score's quotient by a multiply (walk_forest)
    const uint32_t* hi4;          // T * 2^(D-1)
    const uint2* last4;           // T * 2^(D-1)
    const double* p1;             // n_pairs
    const uint32_t* roots;        // v5 (round 4): the T root node words, 64-byte aligned, padded to whole batches of 16
};

struct V2Args {
    FilterArgs f;
    PackedGroupView pg[UGVC_N_GROUPS];
    const FeatDesc* desc;         // [3][kMaxFeatures]
    const uint16_t* lut;
    const float* thr;             // search-only features first: [0, thr_lds_len) is copied to LDS
    int thr_lds_len;
    const uint8_t* css_lut;       // 256 entries, index l1<<6 | ref<<4 | alt<<2 | r1 (bases - 1)
    int32_t* brackets;            // [(n_blocks + 1)][kJoinArrays]
    uint4* records[UGVC_N_GROUPS];
    uint32_t* counters;           // [UGVC_N_GROUPS][kShards] x kCounterStride, zeroed every launch
    int n_blocks;
    int shard_cap;                // records per shard region
    // v3
    const uint2* desc3;           // [3][kMaxFeatures]: {off[0:20) | kind[30:32), len[0:16) | dword[16:18) | bit_off[18:23)}
    int thr_bits4[4];             // descent depth of the float-feature searches (qual, sor, vaf, gc)
    uint8_t dw3[kMaxFeatures + 2];// group-uniform dword of every feature's code
    uint32_t boolmask3[UGVC_N_GROUPS];   // which of the seven 0/1 features (bit f-15) a group's model tests below 1
    int32_t* brackets3;           // [(n_blocks + 1)][8]
    int na3[8];                   // lengths of the searched arrays
    unsigned long long* prof;     // 8 phase-clock accumulators (profiling aid), or null
};

// ---- v5 (kernels_v5.hip) ---------------------------------------------------------------------
// Codes are 16-bit values addressed by FEATURE index (plane f = feature f, 128 bytes per wave):
//   * float features (qual, sor, vaf, gc): rank among the group's sorted thresholds, as above;
//   * every other feature is a non-negative integer: code(x) = x < 0 ? 0 : min(x, cap) + 1 with
//     cap = floor(largest threshold) + 1, node rank = floor(thr) + 1, so code > rank <=> x > thr (exact) and
//     no code table is needed - one clamp per feature.
// Node word = rank[0:16) | (f * 128)[16:32); forests in the single-sum layout (hi / last / p1).
constexpr int kRec5Dwords = 12;               // raw record: 20 codes (u16, by feature) | pad | variant index (dword 11)
constexpr int kTile5 = 64;                    // variants per tile = one wave
constexpr int kMinRowsWg5 = 1024;             // a workgroup of the fused kernel owns at least this many rows
constexpr int kJoin5 = UGVC_MAX_TRACKS + 2;   // runs, tracks, blacklist
constexpr int kGtabBytes = 640;          // clamps | float-slice descriptors | per-wave class counts | contig-boundary words | second-contig state (fused5_kernel)
constexpr int kBlCap5 = 64;                   // staged blacklist keys per SNP tile

struct V5Args {
    FilterArgs f;
    PackedGroupView pg[UGVC_N_GROUPS];   // hi4 / last4 hold the raw-code node tables; n_planes = kMaxFeatures
    const float* thr;                    // float-feature thresholds, group 0's four slices first
    const uint2* desc3;                  // [3][kMaxFeatures] {off, len | ...} of the float slices
    int thr_lds_len;                     // floats staged by the indel path (all groups)
    int thr0_len;                        // floats of group 0 (prefix of thr)
    int thr_bits;                        // descent depth of the float-feature searches of the indel groups (sorted slices)
    const float* eyt;                    // group 0: qual / sor / vaf thresholds as complete search trees (Eytzinger order, +inf padded)
    int eyt_off[3], eyt_bits[3], eyt_len;
    const uint16_t* gcr;                 // [3][121] rank of gc_content = count / len among the group's thresholds (384 entries)
    int cap5[UGVC_N_GROUPS][kMaxFeatures];
    uint32_t used5[UGVC_N_GROUPS];       // features a group's forest tests
    const uint8_t* css_lut;
    uint32_t* snp_idx;                   // [workgroup][list_stride] row indices of the workgroup's substitutions, in order; ~0u = padding
    uint32_t* indel_idx;                 // ... of its indels
    int rows_wg;                         // consecutive rows a workgroup owns
    int list_stride;                     // entries per workgroup list (rows_wg rounded up to 64, + 64 of padding)
    int run_forest;                      // the pass has indel records to walk (indels resident and a model for them)
    int indel_one_round;                 // indel tiles: every narrow slice and the blacklist keys staged together (scratch_indel holds them)
    uint32_t iwide;                      // bit t: interval table t is dense enough for the six-rows-per-lane slice under an indel tile
    uint4* rec5[UGVC_N_GROUPS];
    uint32_t* counters;                  // [UGVC_N_GROUPS][kShards] x kCounterStride: this pass's record counts
    uint32_t* counters_next;             // the other set: zeroed by this pass's forest kernel for the next pass
    int shard_cap5;
    int jcap[kJoin5];                    // staged elements per table in the SNP path: 64 or 128 (blacklist: kBlCap5)
    int joff[kJoin5];                    // dword offset of the staged starts (ends follow at + jcap); keys: 2 dwords each
    int na[kJoin5];                      // table lengths
    int scratch_bytes;                   // per-wave LDS scratch of the fused kernel's SNP waves
    int scratch_indel;                   // ... of its indel waves (48-byte window rows)
    int n_waves;
    int n_indel_waves;                   // waves of a fused workgroup that work on indel tiles
    int indel_w;                         // cost of an indel tile relative to an SNP tile, in 1/256 (the wave role split follows it)
    unsigned long long* wave_clk;        // profiling (UGVC_WAVE_CLK): per workgroup and wave {kernel entry, first tile, end, tiles | indel tiles << 32} in s_memtime ticks, or null
    unsigned short snp_cum[17];          // profiling (UGVC_SNP_W): cumulative tile weights of the SNP waves, or all zero (equal consecutive shares)
    int forest_lds_tail;                 // forest5_kernel: byte offset of its shard-offset / total words in the dynamic LDS
};

int v5_fill_args(ugvc_ctx* ctx, V5Args& v, const FilterArgs& a, bool scoring = true, int wg_per_cu = 1);

int pack_model_group(ugvc_ctx* ctx, int g, const int32_t* feature, const float* threshold,
                     const int32_t* left, const int32_t* right, int n_nodes, const int32_t* tree_root,
                     int n_trees, const double* leaf_value, int n_leaves, int n_features, int kind,
                     float base, int depth);
int clear_model_group(ugvc_ctx* ctx, int g);
int finalize_pack(ugvc_ctx* ctx);
int build_css_lut(ugvc_ctx* ctx);
bool v2_available(ugvc_ctx* ctx);
int launch_filter_v3(ugvc_ctx* ctx, const FilterArgs& a);
int launch_filter_v5(ugvc_ctx* ctx, const FilterArgs& a);
int launch_feature_matrix_v5(ugvc_ctx* ctx, const FilterArgs& a);
bool fm5_available(ugvc_ctx* ctx);
bool v5_available(ugvc_ctx* ctx);
int launch_forest3(ugvc_ctx* ctx, const V2Args& v, const FilterArgs& a);
int v2_fill_args(ugvc_ctx* ctx, V2Args& v, int64_t n, int n_tiles = 0);
bool v3_available(ugvc_ctx* ctx);
void v2_destroy(ugvc_ctx* ctx);
const char* v2_reason(ugvc_ctx* ctx);
int v2_phase_clocks(ugvc_ctx* ctx, uint64_t out[8], int reset);

}  // namespace ugvc
