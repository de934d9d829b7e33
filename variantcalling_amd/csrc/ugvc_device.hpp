// Shared device-side views and host context for libugvc_mi355x.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/ugvc_mi355x.h"

namespace ugvc {

constexpr int kBlock = 256;                                   // variants per workgroup
constexpr int kMaxFeatures = UGVC_N_BASE_FEATURES + UGVC_MAX_TRACKS;
constexpr int kMotif = 5;
constexpr int kGcWindow = 10;
constexpr int kRefFrontPad = 64;                              // zero bytes in front of the reference buffer (v4 window loads)

// 16-byte tree node.  Leaves self-loop (left = right = self) so a fixed-depth walk needs no leaf
// test; the bits of a leaf's `thr` are its payload row.
struct __attribute__((aligned(16))) Node {
    float thr;
    int feat;
    int left;
    int right;
};

struct ForestView {
    const Node* nodes;
    const int* roots;
    const double2* leaves;     // RF: (p_class0, p_class1); GBT: (margin, 0)
    int n_trees;
    int depth;                 // max root->leaf edges
    int kind;
    float base;
    // dense layout (every tree padded to a complete binary tree of `depth` levels, BFS order):
    const float2* dense;       // {thr, feat-as-int-bits}, (2^depth - 1) per tree
    const double2* dense_leaves;  // 2^depth per tree
};

struct TrackView {
    const int32_t* starts;
    const int32_t* ends;
    const int32_t* ptr;        // CSR by contig
    const int32_t* coarse;     // starts[64 k]: a 1/64 sample that stays in L2 (K0's bracket search), or null
};

struct FilterArgs {
    int64_t n;
    const uint16_t* contig;
    const int32_t* pos;
    const uint16_t* ref_len;
    const uint16_t* alt_len;
    const uint32_t* ref_off;
    const uint32_t* alt_off;
    const uint8_t* alleles;
    const float* qual;
    const float* sor;
    const int32_t* dp;
    const int32_t* ad_ref;
    const int32_t* ad_alt;
    const uint8_t* gq;
    // reference genome
    const uint8_t* ref;
    const int64_t* contig_off;
    // side tables
    TrackView runs;
    int has_runs, hpol_dist, mark_hpol;
    TrackView tracks[UGVC_MAX_TRACKS];
    int n_tracks;
    const uint64_t* bl;
    const uint64_t* bl_coarse; // bl[64 k], or null
    int64_t n_bl;
    ForestView forest[UGVC_N_GROUPS];
    uint8_t flow[4];
    // outputs
    float* score;
    uint8_t* filter;
    uint8_t* flags;
    float* X;                  // optional N x F
    uint8_t* group;            // optional
    int n_contigs;
    int ablate;                // debug/profiling only: bit0 no forest walk, bit1 no side-table joins,
                               // bit2 no cycle-skip, bit3 no reference-derived features
};

struct DeviceBuf {
    void* p = nullptr;
    size_t cap = 0;
    template <class T> T* as() const { return static_cast<T*>(p); }
};

}  // namespace ugvc

struct ugvc_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int n_cus = 0;
    // resident tables
    ugvc::DeviceBuf ref, contig_off;
    int n_contigs = 0;
    int64_t ref_len = 0;
    ugvc::DeviceBuf runs_s, runs_e, runs_p, runs_c;
    int has_runs = 0, hpol_dist = 10, mark_hpol = 1;
    ugvc::DeviceBuf trk_s[UGVC_MAX_TRACKS], trk_e[UGVC_MAX_TRACKS], trk_p[UGVC_MAX_TRACKS], trk_c[UGVC_MAX_TRACKS];
    int track_set[UGVC_MAX_TRACKS] = {0, 0, 0, 0, 0};
    int n_tracks = 0;
    int64_t runs_n = 0, trk_n[UGVC_MAX_TRACKS] = {0, 0, 0, 0, 0};
    // v3 needs: runs disjoint and sorted; tracks with non-decreasing starts AND ends per contig
    int runs_fast = 1, trk_fast[UGVC_MAX_TRACKS] = {1, 1, 1, 1, 1};
    ugvc::DeviceBuf bl, bl_c;          // bl_c: every 64th key
    int64_t n_bl = 0;
    uint8_t flow[4] = {4, 3, 2, 1};   // TGCA
    struct Model {
        ugvc::DeviceBuf nodes, roots, leaves, dense, dense_leaves;
        int n_trees = 0, depth = 0, kind = 0, n_features = 0, set = 0, has_dense = 0;
        float base = 0.f;
    } model[UGVC_N_GROUPS];
    // resident variants + results
    int64_t n = 0;
    int64_t n_indel = 0;          // rows with ref_len != alt_len (counted at upload)
    ugvc::DeviceBuf v_contig, v_pos, v_rl, v_al, v_ro, v_ao, v_alleles, v_qual, v_sor, v_dp,
        v_adr, v_ada, v_gq;
    ugvc::DeviceBuf r_score, r_filter, r_flags, x_mat, x_group;
    int scored = 0;                    // the resident result columns hold a scoring pass over the resident variants
    // pileup
    int64_t pl_n = 0, pl_obs = 0, pl_span = 0;   // pl_span: longest staged span of a 256-locus workgroup (picks the LDS capacity)
    ugvc::DeviceBuf pl_off, pl_off32, pl_obsb, pl_out;
    int pl_compact = 0;              // device layout of the pileup table (kernels_aux.hip)
    // SEC database (kernels_sec.hip): sorted locus keys, every 64th key, k expected counts per locus
    ugvc::DeviceBuf sec_keys, sec_coarse, sec_exp, sec_lgtab;
    int64_t n_sec = 0;
    int sec_k = 0;
    // gather
    ugvc::DeviceBuf g_score[2], g_filter[2], g_flags[2];   // double-buffered: pass i+1 writes one while the collective of pass i reads the other
    int g_cur = 0;                                         // buffer the last gather used
    void* comm = nullptr;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_res_ready = nullptr, ev_gather_done[2] = {nullptr, nullptr};
    int gather_pending[2] = {0, 0};
    int rank = 0, world = 1;
    int kernel_variant = 0;
    void* bounce = nullptr;         // ugvc::Bounce (devmem.hip): the two pinned slots every host <-> device copy goes through
    int clk_probe = 0;              // the next scoring passes leave their clock words (ugvc_pass_clock)
    unsigned long long clk_rt = 0, clk_sh = 0;   // 100 MHz ticks / shader-clock ticks across workgroup 0's first wave of the last probed pass
    ugvc::DeviceBuf wclk_buf;       // the clock words of a probed pass (ugvc_pass_clock, UGVC_WAVE_CLK): per context, freed with it
    int step_events = 1;            // ugvc_timed_steps: an event pair around every step (0: one pair around the run)
    std::vector<float> step_ms;     // per-step kernel times of the last ugvc_timed_steps
    void* v2 = nullptr;             // ugvc::V2State (model_pack.hip)
    void* pipe = nullptr;           // ugvc::PipeState (pipeline.hip): host pool, pinned staging, copy streams
    int64_t density_n = 0;          // when a pass covers a row range of a larger callset: that callset's size (table-density estimates)
};

namespace ugvc {
void set_error(const std::string& msg);
int fail(const std::string& msg);
int dev_alloc(void** out, size_t bytes);          // devmem.hip: hipMalloc, or the guarded / poisoned debug mappings (UGVC_GUARD, UGVC_POISON)
void dev_free(void* p);
bool guard_on();                                  // UGVC_GUARD is set (tests)
void launch_note(const char* name, hipStream_t stream);   // breadcrumb ring (+ name on stderr under UGVC_DEBUG_SYNC; LDS poison under UGVC_POISON)
void launch_done(const char* name, hipStream_t stream);
hipError_t copy_in(ugvc_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);    // devmem.hip: through the context's pinned slots,
hipError_t copy_out(ugvc_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);   // never a DMA on the caller's pageable memory
void bounce_destroy(ugvc_ctx* ctx);
int lds_probe(ugvc_ctx* ctx, uint32_t* host_out, int n_wg);   // devmem.hip: 8 words of unwritten LDS per workgroup
int ensure(DeviceBuf& b, size_t bytes);
int upload(ugvc_ctx* ctx, DeviceBuf& b, const void* src, size_t bytes);
int build_args(ugvc_ctx* ctx, FilterArgs& a, bool want_x);
int launch_filter(ugvc_ctx* ctx, const FilterArgs& a, bool score, bool write_x);
int launch_pileup(ugvc_ctx* ctx);
int launch_sec(ugvc_ctx* ctx, const int32_t* d_actual, const int32_t* d_expected, int64_t n, int k,
               double* d_lik, double* d_ratio);
}  // namespace ugvc

#define UGVC_HIP(expr)                                                                     \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess)                                                              \
            return ugvc::fail(std::string(#expr) + ": " + hipGetErrorString(_e));          \
    } while (0)

// Every kernel launch of the library: the name goes to the breadcrumb ring (devmem.hip); UGVC_DEBUG_SYNC=1 names the launch on
// stderr and waits for it, so that the last name printed before a GPU memory fault is the kernel that faulted.
#define UGVC_LAUNCH(kern, grid, block, lds, stream, ...)                            \
    do {                                                                            \
        ugvc::launch_note(#kern, stream);                                                 \
        hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);            \
        ugvc::launch_done(#kern, stream);                                           \
    } while (0)
