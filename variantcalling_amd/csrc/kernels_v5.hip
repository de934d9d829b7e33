// v5 scoring pass (gfx950): featurize -> lookup -> score -> FILTER as four launches.
//
// What v3 measured (profiles/r01_*): its featurize kernel (K1, 378 us per 5 M variants) waits on a chain of
// dependent memory round trips per 256-variant tile at 4 waves/SIMD, with the vector ALU mostly idle; its forest
// kernel (K2, 321 us) is bound by the LDS pipeline and runs AFTER K1; the two exchange a 16-byte record per
// variant through HBM (160 MB per pass) and K2 stores its results scattered.  v5 puts both on the same waves:
//
//   compact5_kernel   every 1024-variant block splits its rows by variant class (ref_len == alt_len: the SNP
//                     forest; else an indel) into tiles of 64 row indices; a tile is one wave's work and is pure
//                     in class, so every lane of a wave walks the SAME forest (no lane is idle in a walk).  Tile
//                     slots come from 64 sharded counters (one atomic per block and class).
//   bracket5_kernel   per tile, the lower bound of its first (indel tiles: and last) variant in every searched
//                     side table (two-level search, L2-resident 1/64 sample first).
//   fused5_kernel     one 16-wave workgroup per CU holds the SNP forest in LDS (rank-coded complete trees,
//                     single-sum layout) and the float thresholds.  A wave is autonomous - no workgroup barrier
//                     after the prologue: it loads its tile's columns, an 11-base reference window per lane (one
//                     16-byte load, realigned with v_alignbyte), stages the side-table slices its tile can touch
//                     in wave-private LDS (sentinel padded: the lock-step descents carry no bounds test and no
//                     branch), derives the features, writes 16-bit codes straight into its code planes and walks
//                     the forest; score / FILTER / flags leave in variant order (coalesced).  While one wave waits
//                     for memory the other fifteen walk: the featurize latency that bounded K1 disappears under
//                     the LDS-bound walk.  Indel tiles (18 % of a WGS callset) are featurised by the same waves
//                     between SNP tiles (48-byte window, homopolymer logic, bracketed searches on the L2-resident
//                     tables) and leave as 48-byte raw-code records in their variant-type group's list.
//   forest5_kernel    walks the indel groups' forests over those records (the v3 forest kernel on raw records).
//
// Codes: floats (qual, sor, vaf, gc) are ranked against the group's sorted thresholds (exact, as v3); every
// other feature is a non-negative integer and is used as it stands, clamped to one past the largest threshold
// the forest tests (ugvc_v2.hpp) - no code tables, no gathers.
// The number of annotation tracks is a template parameter: the per-table code is straight-line.
// Semantics are those of the oracle (oracle/oracle.py); parity tests run v5, v3 and v1 against it.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "ugvc_walk.hpp"

namespace ugvc {

#define UGVC_CONST __attribute__((address_space(4)))
// Loads through the constant address space: data written by an EARLIER launch, wave-uniform address -> s_load.
template <class T> __device__ __forceinline__ T cload(const T* p) {
    return *(const UGVC_CONST T*)(uintptr_t)p;
}
__device__ __forceinline__ uint2 cload2(const uint2* p) {
    const u32x2_t x = *(const UGVC_CONST u32x2_t*)(uintptr_t)p;
    return make_uint2(x.x, x.y);
}
__device__ __forceinline__ void lds_st32(uint32_t a, int32_t x) { *(UGVC_LDS int32_t*)(uintptr_t)a = x; }
__device__ __forceinline__ void lds_st64(uint32_t a, uint64_t x) { *(UGVC_LDS uint64_t*)(uintptr_t)a = x; }
__device__ __forceinline__ void lds_st16(uint32_t a, uint32_t x) { *(UGVC_LDS uint16_t*)(uintptr_t)a = (uint16_t)x; }
__device__ __forceinline__ uint32_t lds_u8(uint32_t a) { return *(UGVC_LDS const uint8_t*)(uintptr_t)a; }

constexpr int kGcRank = 121;         // gc_content takes 121 values (count / len, len and count in 0..10): its rank code is a table
constexpr int kGcRankBytes = 768;    // 3 groups x 121 u16, padded
constexpr int kWinRowB = kWinStride * 4;

// ---- Kc: variant classes -> tiles of 64 row indices ---------------------------------------------------
// A workgroup of four waves owns kCBlock5 = 1024 consecutive rows (256 per wave, four rounds of 64): small
// workgroups keep thousands of them in flight, so the one returning atomic each needs is hidden.
__global__ __launch_bounds__(256) void compact5_kernel(const V5Args v) {
    __shared__ unsigned ws[4], wi[4];
    __shared__ unsigned base_s, base_i;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int64_t g = (int64_t)blockIdx.x * 256 + tid; g < UGVC_N_GROUPS * kShards; g += (int64_t)gridDim.x * 256)
        v.counters[g * kCounterStride] = 0;                            // the record lists of this pass start empty
    const int64_t i0 = (int64_t)blockIdx.x * kCBlock5 + wave * 256 + lane;
    unsigned long long ms[4], mi[4];
    unsigned cs = 0, ci = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t i = i0 + r * 64;
        bool snp = false, ind = false;
        if (i < v.f.n) {
            snp = v.f.ref_len[i] == v.f.alt_len[i];
            ind = !snp;
        }
        ms[r] = __ballot(snp);
        mi[r] = __ballot(ind);
        cs += (unsigned)__popcll(ms[r]);
        ci += (unsigned)__popcll(mi[r]);
    }
    if (lane == 0) { ws[wave] = cs; wi[wave] = ci; }
    __syncthreads();
    unsigned ps = 0, pi = 0, ts = 0, ti = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const unsigned a = ws[w], b = wi[w];
        ps += w < wave ? a : 0u;
        pi += w < wave ? b : 0u;
        ts += a;
        ti += b;
    }
    const unsigned nts = (ts + 63) >> 6, nti = (ti + 63) >> 6;
    // tile slots: 64 shards of `shard_tiles` slots per class, one atomic per block and class on the block's shard
    // (a single counter per class serialises ~5000 same-address atomics: 116 us per 5 M variants)
    const unsigned shard = blockIdx.x & (kTileShards5 - 1);
    if (tid == 0) base_s = nts ? atomicAdd(&v.tile_cnt[shard * kTileCntStride5], nts) : 0u;
    if (tid == 64) base_i = nti ? atomicAdd(&v.tile_cnt[(kTileShards5 + shard) * kTileCntStride5], nti) : 0u;
    __syncthreads();
    const unsigned bs = shard * (unsigned)v.shard_tiles + base_s, bi = shard * (unsigned)v.shard_tiles + base_i;
    const unsigned long long below = (1ull << lane) - 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t i = (uint32_t)(i0 + r * 64);
        if ((ms[r] >> lane) & 1) v.snp_idx[(size_t)bs * 64 + ps + (unsigned)__popcll(ms[r] & below)] = i;
        if ((mi[r] >> lane) & 1) v.indel_idx[(size_t)bi * 64 + pi + (unsigned)__popcll(mi[r] & below)] = i;
        ps += (unsigned)__popcll(ms[r]);
        pi += (unsigned)__popcll(mi[r]);
    }
    if ((unsigned)tid < nts * 64 - ts) v.snp_idx[(size_t)bs * 64 + ts + tid] = ~0u;          // padding of the last tile
    if ((unsigned)tid < nti * 64 - ti) v.indel_idx[(size_t)bi * 64 + ti + tid] = ~0u;
    if ((unsigned)tid < nts) v.tile_n[bs + tid] = (uint8_t)((unsigned)tid + 1 < nts ? 64u : ts - 64u * (nts - 1));
    if ((unsigned)tid < nti) v.tile_n[(size_t)v.max_tiles + bi + tid] = (uint8_t)((unsigned)tid + 1 < nti ? 64u : ti - 64u * (nti - 1));
}

// ---- K0: per tile, lower bounds of its first (indel tiles: and last) variant in the searched tables ---
// One thread per (REAL tile, searched table): the tile counters of both classes are scanned in LDS, a thread finds
// its tile by its rank among the real tiles - no lane idles on an empty slot or an absent table (the searches are
// chains of ~22 dependent loads: what counts is how many waves the launch needs, not their instruction count).
__global__ __launch_bounds__(256) void bracket5_kernel(const V5Args v) {
    __shared__ unsigned incl[2][kTileShards5];
    const int tid = threadIdx.x;
    if (tid < 2 * kTileShards5) {
        const int cls = tid >> 6, sh = tid & 63;
        unsigned x = v.tile_cnt[(cls * kTileShards5 + sh) * kTileCntStride5];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned y = __shfl_up(x, d);
            if (sh >= d) x += y;
        }
        incl[cls][sh] = x;
    }
    __syncthreads();
    const FilterArgs& f = v.f;
    // active tables, in order: runs (if any), tracks, blacklist (if any)
    const int n_act = (f.has_runs ? 1 : 0) + f.n_tracks + (f.n_bl > 0 ? 1 : 0);
    const int n_thr = n_act > 0 ? n_act : 1;                            // (no table at all: the SNP tiles still get their contig word)
    const int64_t ns = incl[0][kTileShards5 - 1], ni = incl[1][kTileShards5 - 1];
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + tid;
    int64_t rank;                       // rank of the tile among the real tiles of its class
    int k, last;
    bool is_indel;
    if (gid < ns * n_thr) { rank = gid / n_thr; k = (int)(gid - rank * n_thr); last = 0; is_indel = false; }
    else {
        const int64_t g2 = gid - ns * n_thr;
        if (n_act == 0 || g2 >= ni * 2 * n_act) return;
        rank = g2 / (2 * n_act);
        const int r = (int)(g2 - rank * 2 * n_act);
        last = r >= n_act;
        k = last ? r - n_act : r;
        is_indel = true;
    }
    int t = k + (f.has_runs ? 0 : 1);                                   // table index: 0 runs, 1.. tracks, kJoin5 - 1 blacklist
    if (t > f.n_tracks) t = kJoin5 - 1;
    const unsigned* inc = incl[is_indel ? 1 : 0];
    int lo = 0, len = kTileShards5;
    while (len > 0) {                                                   // first shard whose inclusive count exceeds the rank
        const int half = len >> 1;
        const bool le = inc[lo + half] <= (unsigned)rank;
        lo = le ? lo + half + 1 : lo;
        len = le ? len - half - 1 : half;
    }
    const int shard = lo;
    const int64_t tile = (int64_t)shard * v.shard_tiles + (rank - (shard > 0 ? inc[shard - 1] : 0u));
    const uint32_t* list = is_indel ? v.indel_idx : v.snp_idx;
    const int slot = last ? (int)v.tile_n[(size_t)v.max_tiles + tile] - 1 : 0;
    const uint32_t i = list[tile * 64 + slot];
    const int c = f.contig[i], pos = f.pos[i];
    int out = 0;
    if (n_act == 0) {
    } else if (t == kJoin5 - 1) out = lb_two_level_g<uint64_t>(f.bl, f.bl_coarse, 0, (int)f.n_bl, ((uint64_t)c << 32) | (uint32_t)pos);
    else {
        const TrackView& tv = table_view(f, t);
        out = lb_two_level_g<int32_t>(tv.starts, tv.coarse, tv.ptr[c], tv.ptr[c + 1], pos);
    }
    if (is_indel) {
        // indel tile record (kRecI5 ints): [t] / [8 + t] lower bounds of the first / last variant per table,
        // [16 + 2t], [17 + 2t] the contig's row range, [28] contig (bit 31: the tile spans contigs)
        int32_t* rec = v.br_indel + tile * kRecI5;
        rec[(last ? 8 : 0) + t] = out;
        if (!last && t != kJoin5 - 1) {
            const TrackView& tv = table_view(f, t);
            rec[16 + 2 * t] = tv.ptr[c];
            rec[17 + 2 * t] = tv.ptr[c + 1];
        }
        if (!last && k == 0) {
            const int n = (int)v.tile_n[(size_t)v.max_tiles + tile];
            const int c_last = f.contig[list[tile * 64 + (n > 0 ? n - 1 : 0)]];
            rec[28] = c | (c_last != c ? INT32_MIN : 0);
        }
    } else {
        // SNP tile record (kRecS5 ints): [t] lower bound per table, [7] contig of the tile (bit 31: the tile spans
        // contigs), [8 + 2t], [9 + 2t] the contig's row range of table t, [20..23] its span of the reference - everything the fused kernel needs to fetch
        // the tile's slices BEFORE it has seen the tile's columns
        int32_t* rec = v.br_snp + tile * kRecS5;
        if (n_act > 0) rec[t] = out;
        if (n_act > 0 && t != kJoin5 - 1) {
            const TrackView& tv = table_view(f, t);
            rec[8 + 2 * t] = tv.ptr[c];
            rec[9 + 2 * t] = tv.ptr[c + 1];
        }
        if (k == 0) {
            const int n = (int)v.tile_n[tile];
            const int c_last = f.contig[list[tile * 64 + (n > 0 ? n - 1 : 0)]];
            rec[7] = c | (c_last != c ? INT32_MIN : 0);
            const int64_t clo = f.contig_off[c], chi = f.contig_off[c + 1];      // [20..23]: the contig's span of the reference
            rec[20] = (int32_t)(uint32_t)clo; rec[21] = (int32_t)(clo >> 32);
            rec[22] = (int32_t)(uint32_t)chi; rec[23] = (int32_t)(chi >> 32);
        }
    }
}

#ifdef UGVC_PHASE_CLOCK
struct PhaseClk { uint64_t last; uint64_t acc[8]; };
#define CLK(pc, k) do { const uint64_t now_ = __builtin_readcyclecounter(); (pc).acc[k] += now_ - (pc).last; (pc).last = now_; } while (0)
#else
struct PhaseClk {};
#define CLK(pc, k) do { } while (0)
#endif

// ---- joins ---------------------------------------------------------------------------------------
struct JoinOut {
    bool inside_run, close_run, cohort;
    uint32_t trk;                    // bit t: inside an interval of annotation track t
};

// After the rank sg = #starts < pos of a table (global index): membership / proximity from a few reads of
// the starts and ends around it.  S / E return starts[i] / ends[i] for any i the guards allow.
template <class GetS, class GetE>
__device__ __forceinline__ void interval_verdict(int t, int sg, int plo, int phi, int pos, int D, GetS S, GetE E, JoinOut& o) {
    const bool valid = sg > plo;
    const int e1v = E(sg - 1);
    if (t == 0) {
        // runs are disjoint: #ends < pos is sg-1 or sg
        const bool ins_run = valid && e1v >= pos;
        if (phi > plo) {
            const int eg = valid ? sg - 1 + (e1v < pos ? 1 : 0) : sg;
            auto near = [&](int x) { const int d = pos - x; return (d < 0 ? -d : d) < D; };
            const int s1v = S(sg - 1), s0v = S(sg);
            bool cd = (valid && near(s1v)) || (sg <= phi - 1 && near(s0v));
            const int ee = E(eg), em = E(eg - 1);
            cd = cd || (eg - 1 >= plo && near(em)) || (eg <= phi - 1 && near(ee));
            o.inside_run = ins_run;
            o.close_run = cd && !ins_run;
        }
    } else {
        const int e2v = E(sg - 2);
        const bool in = valid && e1v >= pos && (sg - 1 == plo || e2v < pos);
        o.trk |= in ? 1u << (t - 1) : 0u;
    }
}

// One table searched in HBM (rows [lo, hi) per lane): the rare paths - a tile that spans contigs, a slice that
// outgrew its staging area.  Kept out of line so the common path stays small.
__device__ __forceinline__ void join_one_global(const FilterArgs* ap, int t, int lo, int hi, int plo, int phi, int pos, uint64_t key,
                                             JoinOut* op) {
    const FilterArgs& a = *ap;
    JoinOut o = *op;
    if (t == kJoin5 - 1) {
        const int r = lb_u64_g(a.bl, lo, hi, key);
        o.cohort = r < (int)a.n_bl && a.bl[r] == key;
    } else if (phi > plo) {
        const TrackView& tv = table_view(a, t);
        const int sg = lb_i32_g(tv.starts, lo, hi, pos);
        const int top = phi - 1;
        auto S = [&](int i) { return tv.starts[i < plo ? plo : (i > top ? top : i)]; };     // clamped into the contig's rows;
        auto E = [&](int i) { return tv.ends[i < plo ? plo : (i > top ? top : i)]; };       // interval_verdict's guards discard them
        if (t == 0) { o.inside_run = o.close_run = false; }
        else o.trk &= ~(1u << (t - 1));
        interval_verdict(t, sg, plo, phi, pos, a.hpol_dist, S, E, o);
    }
    *op = o;
}

// Indel tiles cover ~5x the span of an SNP tile, so their slices are staged per table: rows [lo - 2, hi + 2) between
// the lower bounds of the tile's first and last variant (K0's record) - at most kIndelRows - in registers first
// (fetched from the record alone, under the column / window chain of the tile), then into the wave's scratch two
// tables at a time once the window rows are dead.  A bracket wider than that is searched in HBM (join_one_global).
// (Measured and dropped: the same searches as dependent gathers on the resident tables - binary: nine round trips per
// tile; 8-ary: three, but 105 scattered 64-lane gathers instead of 45, no faster.)
constexpr int kIndelRows = 128;                     // staged rows per table (two per lane); 127 searchable + sentinel
constexpr uint32_t kIndelSlotB = kIndelRows * 8;    // starts | ends, or 128 blacklist keys

template <int NT>
struct IndelPre {
    int sv[NT][2], ev[NT][2];
    uint64_t bl[2];
};

// rank of `pos` among the 127 searchable staged starts at A (ascending, sentinel padded): seven branch-free steps
__device__ __forceinline__ uint32_t staged_rank(uint32_t A, int pos) {
    uint32_t p = A - 4u;
#pragma unroll
    for (int sb = 256; sb >= 4; sb >>= 1) {
        const uint32_t cand = p + (uint32_t)sb;
        p = lds_i32(cand) < pos ? cand : p;
    }
    return (p + 4u - A) >> 2;
}

__device__ __forceinline__ void stage_rows(uint32_t slot_b, int L0, int plo, int phi, const int (&sv)[2], const int (&ev)[2], int lane) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int gi = L0 + 64 * h + lane;
        lds_st32(slot_b + 256u * h + 4u * lane, gi < plo ? INT32_MIN : (gi >= phi ? INT32_MAX : sv[h]));
        lds_st32(slot_b + 512u + 256u * h + 4u * lane, ev[h]);
    }
}

__device__ __forceinline__ void staged_verdict(const FilterArgs& a, uint32_t slot_b, int t, int L0, int plo, int phi, int pos, JoinOut& o) {
    if (phi <= plo) return;
    const int sg = L0 + (int)staged_rank(slot_b, pos);
    auto S = [&](int gi) { return lds_i32(slot_b + 4u * (uint32_t)(gi - L0)); };
    auto E = [&](int gi) { return lds_i32(slot_b + 512u + 4u * (uint32_t)(gi - L0)); };
    interval_verdict(t, sg, plo, phi, pos, a.hpol_dist, S, E, o);
}

// A table too dense for the two-rows-per-lane slice (a 3 M-interval track under a 220 kb indel tile: ~210 rows):
// six rows per lane, the whole scratch, a round of its own; the descent clamps its probes to the last staged row
// (rows past the searched range compare like the padding would).
constexpr int kWideChunks = 6, kWideRows = 64 * kWideChunks;          // 384 rows: starts | ends = 3072 B

__device__ __forceinline__ void wide_load(const TrackView& tv, int L0, int top, int lane, int (&wv)[kWideChunks], int (&we)[kWideChunks]) {
#pragma unroll
    for (int h = 0; h < kWideChunks; ++h) {
        const uint32_t gs = (uint32_t)max(min(L0 + 64 * h + lane, top), 0);
        wv[h] = tv.starts[gs]; we[h] = tv.ends[gs];
    }
}

__device__ __forceinline__ void wide_verdict(const FilterArgs& a, uint32_t A, int t, int L0, int plo, int phi, int pos, int lane,
                                             const int (&wv)[kWideChunks], const int (&we)[kWideChunks], JoinOut& o) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int h = 0; h < kWideChunks; ++h) {
        const int gi = L0 + 64 * h + lane;
        lds_st32(A + 256u * h + 4u * lane, gi < plo ? INT32_MIN : (gi >= phi ? INT32_MAX : wv[h]));
        lds_st32(A + 4u * kWideRows + 256u * h + 4u * lane, we[h]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (phi <= plo) return;
    uint32_t p = A - 4u;
    const uint32_t last = A + 4u * (kWideRows - 1);
#pragma unroll
    for (int sb = 1024; sb >= 4; sb >>= 1) {
        const uint32_t cand = min(p + (uint32_t)sb, last);
        p = lds_i32(cand) < pos ? cand : p;
    }
    const int sg = L0 + (int)((p + 4u - A) >> 2);
    auto S = [&](int gi) { return lds_i32(A + 4u * (uint32_t)(gi - L0)); };
    auto E = [&](int gi) { return lds_i32(A + 4u * kWideRows + 4u * (uint32_t)(gi - L0)); };
    interval_verdict(t, sg, plo, phi, pos, a.hpol_dist, S, E, o);
}

__device__ __forceinline__ uint32_t raw_code(int x, int cap) {          // x < 0 ? 0 : min(x, cap) + 1
    return (uint32_t)(max(min(x, cap), -1) + 1);
}

__device__ __forceinline__ bool any_zero_byte(uint32_t x) { return ((x - 0x01010101u) & ~x & 0x80808080u) != 0; }
// A or T bytes (codes 1, 4) of a packed base word -> 0x01 per byte
__device__ __forceinline__ uint32_t at_bytes(uint32_t x) { return ((x & ~(x >> 1)) | (x >> 2)) & 0x01010101u; }

struct Scratch {                    // LDS byte addresses
    uint32_t base;                  // wave-private: window rows / staged slices / code planes, one after the other in time
    uint32_t eyt_b;                 // group 0's qual / sor / vaf thresholds, level order
    uint32_t thr_b;                 // the indel groups' sorted threshold slices (skewed)
    uint32_t gcr_b;                 // gc rank codes [group][len * 11 + count], u16
    uint32_t css_b;
    uint32_t gtab_b;                // indel groups: clamps and float-slice descriptors
};

// Indel tiles: lock-step descents of qual / sor / vaf over the lane's group's sorted threshold slices (slice of
// feature k: elements [off, off + len) of the staged table); rank = #thresholds < x, NaN -> len (compares false:
// always the right branch).  The table is SKEWED in LDS - element j sits at dword j + (j >> 5): the candidates of a
// power-of-two descent step are congruent modulo the step, i.e. on ONE bank in a plain layout.
__device__ __forceinline__ void rank3_sorted(const float (&fx)[3], uint32_t thr_b, const uint32_t (&off)[3], const uint32_t (&len)[3], int bits,
                                             uint32_t (&cd)[3]) {
    uint32_t q[3];                                               // thresholds known to lie below x
#pragma unroll
    for (int e = 0; e < 3; ++e) q[e] = 0;
    for (int s = bits - 1; s >= 0; --s) {
        uint32_t cand[3];
        float t[3];
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            cand[e] = q[e] + (1u << s);                          // the cand-th threshold of the slice = element off + cand - 1
            const uint32_t j = off[e] + cand[e] - 1u;
            t[e] = lds_f32(thr_b + 4u * (j + (j >> 5)));
        }
#pragma unroll
        for (int e = 0; e < 3; ++e) q[e] = (cand[e] <= len[e] && t[e] < fx[e]) ? cand[e] : q[e];
    }
#pragma unroll
    for (int e = 0; e < 3; ++e) cd[e] = fx[e] != fx[e] ? len[e] : q[e];
}

// SNP tiles: the same ranks from group 0's level-order trees (model_pack.hip): i = 2 i + (t < x), `bits` levels, the
// leaf index is the rank; three VALU per level and feature, reads of one level on consecutive LDS words.
__device__ __forceinline__ void rank3_eyt(const float (&fx)[3], const uint32_t (&base)[3], const int (&bits)[3], const uint32_t (&len)[3],
                                          uint32_t (&cd)[3]) {
    uint32_t i[3] = {1u, 1u, 1u};
    const int bmin = min(bits[0], min(bits[1], bits[2])), bmax = max(bits[0], max(bits[1], bits[2]));
    for (int s = 0; s < bmin; ++s) {
        float t[3];
#pragma unroll
        for (int e = 0; e < 3; ++e) t[e] = lds_f32(base[e] + 4u * i[e]);
#pragma unroll
        for (int e = 0; e < 3; ++e) i[e] = 2u * i[e] + (t[e] < fx[e] ? 1u : 0u);
    }
    for (int s = bmin; s < bmax; ++s) {
#pragma unroll
        for (int e = 0; e < 3; ++e)
            if (s < bits[e]) {
                const float t = lds_f32(base[e] + 4u * i[e]);
                i[e] = 2u * i[e] + (t < fx[e] ? 1u : 0u);
            }
    }
#pragma unroll
    for (int e = 0; e < 3; ++e) cd[e] = fx[e] != fx[e] ? len[e] : i[e] - (1u << bits[e]);
}

struct SnpCols {                    // the columns of one substitution (fetched one tile ahead of their use)
    int c, pos, rl;
    uint32_t ro, ao;
    float qual, sor;
    int dp, adr, ada, gq;
};

__device__ __forceinline__ SnpCols load_snp_cols(const FilterArgs& a, uint32_t i) {
    SnpCols k;
    k.c = a.contig[i]; k.pos = a.pos[i]; k.rl = a.ref_len[i];
    k.ro = a.ref_off[i]; k.ao = a.alt_off[i];
    k.qual = a.qual[i]; k.sor = a.sor[i];
    k.dp = a.dp[i]; k.adr = a.ad_ref[i]; k.ada = a.ad_alt[i]; k.gq = a.gq[i];
    return k;
}

// The side-table slices of one SNP tile, in registers: fetched from the tile record alone, one tile ahead (they are
// in flight during the previous tile's walk and are written to the wave's LDS scratch when that walk has finished
// with its code planes).
template <int NT>
struct SlicePre {
    int sv[NT][2], ev[NT][2];
    uint64_t bl;
};

template <int NT>
__device__ __forceinline__ void issue_slices(const V5Args& v, uint32_t rec, int lane, SlicePre<NT>& s) {
    const FilterArgs& a = v.f;
    s.bl = ~0ull;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        s.sv[t][0] = s.sv[t][1] = s.ev[t][0] = s.ev[t][1] = 0;
        if (t == 0 && !a.has_runs) continue;
        const TrackView& tv = table_view(a, t);
        const int Lt = __builtin_amdgcn_readlane((int)rec, t) - 2;
        const int top = max(v.na[t] - 1, 0);
        {
            const uint32_t gs = (uint32_t)max(min(Lt + lane, top), 0);
            s.sv[t][0] = tv.starts[gs]; s.ev[t][0] = tv.ends[gs];
        }
        if (v.jcap[t] > 64) {
            const uint32_t gs = (uint32_t)max(min(Lt + 64 + lane, top), 0);
            s.sv[t][1] = tv.starts[gs]; s.ev[t][1] = tv.ends[gs];
        }
    }
    if (a.n_bl > 0) {
        const int64_t gi = (int64_t)__builtin_amdgcn_readlane((int)rec, kJoin5 - 1) + lane;
        if (gi < a.n_bl) s.bl = a.bl[gi];
    }
}

// ---- SNP / MNP tile: features of 64 substitutions (ref_len == alt_len) ---------------------------------
// Writes the flags column, leaves the 16-bit codes of the group-0 forest in the wave's code planes.
template <int NTRK>
__device__ __forceinline__ void featurize_snp_tile(const V5Args& v, const Scratch& sc, int64_t tile, int lane, uint32_t i, bool live, bool has_model,
                                                   const SnpCols& k, uint32_t rec, const SlicePre<1 + NTRK>& pre, PhaseClk& pc) {
    constexpr int NT = 1 + NTRK;                               // interval tables: runs + tracks
    const FilterArgs& a = v.f;
    const int c = k.c, pos = k.pos, rl = k.rl;
    const uint32_t ro = k.ro, ao = k.ao;
    const int c0 = rfl(c);
    const bool uni = __builtin_amdgcn_readlane((int)rec, 7) >= 0;   // one contig (all but a handful of tiles): from K0's record
    int64_t clo, chi;
    if (uni) {                                                  // from the record: no round trip in front of the window
        clo = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)rec, 21) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)rec, 20));
        chi = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)rec, 23) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)rec, 22));
    }
    else { clo = a.contig_off[c]; chi = a.contig_off[c + 1]; }
    const uint32_t clen = (uint32_t)(chi - clo);
    const uint32_t p0 = (uint32_t)(pos - 1);
    const int64_t g0 = clo + p0;
    // 11 bases pos-5 .. pos+5: one dword-aligned 16-byte load, realigned per lane (the buffer is padded by 64
    // bytes at both ends)
    const int64_t wa = (g0 - 5) & ~(int64_t)3;
    const uint32_t sh = (uint32_t)(g0 - 5) & 3u;
    const uint4 xw = *reinterpret_cast<const uint4*>(a.ref + wa);
    const uint32_t rbase = a.alleles[ro], abase = a.alleles[ao];

    // ---- side-table slices of this tile -> wave-private LDS (sentinel padded)
    const int n_live = (int)__popcll(__ballot(live));
    const int pos_max = __builtin_amdgcn_readlane(pos, n_live - 1);
    const uint64_t key = ((uint64_t)(uint32_t)c << 32) | (uint32_t)pos;
    const bool joins_on = !(a.ablate & 524288);
    int L[NT], plo[NT], phi[NT];
    const bool stage = uni && joins_on;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        L[t] = __builtin_amdgcn_readlane((int)rec, t) - 2;
        plo[t] = __builtin_amdgcn_readlane((int)rec, 8 + 2 * t);
        phi[t] = __builtin_amdgcn_readlane((int)rec, 9 + 2 * t);
    }
    CLK(pc, 6);
    const float qual = k.qual, sor = k.sor;
    const int dp = k.dp, adr = k.adr, ada = k.ada, gq = k.gq;
    const float vaf = dp > 0 ? __fdiv_rn((float)ada, (float)dp) : 0.0f;
    uint32_t cd[3] = {0, 0, 0};
    if (has_model) {
        const float fx[3] = {qual, sor, vaf};
        const int fj[3] = {0, 1, 5};
        uint32_t base[3], len[3];
        int bits[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            base[q] = sc.eyt_b + 4u * (uint32_t)v.eyt_off[q];
            bits[q] = v.eyt_bits[q];
            len[q] = cload2(v.desc3 + fj[q]).y & 0xFFFFu;             // group 0
        }
        rank3_eyt(fx, base, bits, len, cd);
    }
    CLK(pc, 7);
    if (stage) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t == 0 && !a.has_runs) continue;
            const int cap = v.jcap[t];
            const uint32_t dS = sc.base + 4u * (uint32_t)v.joff[t], dE = dS + 4u * (uint32_t)cap;
            {
                const int gi = L[t] + lane;
                lds_st32(dS + 4u * lane, gi < plo[t] ? INT32_MIN : (gi >= phi[t] ? INT32_MAX : pre.sv[t][0]));
                lds_st32(dE + 4u * lane, pre.ev[t][0]);
            }
            if (cap > 64) {
                const int gi = L[t] + 64 + lane;
                lds_st32(dS + 256u + 4u * lane, gi < plo[t] ? INT32_MIN : (gi >= phi[t] ? INT32_MAX : pre.sv[t][1]));
                lds_st32(dE + 256u + 4u * lane, pre.ev[t][1]);
            }
        }
        if (a.n_bl > 0) lds_st64(sc.base + 4u * (uint32_t)v.joff[kJoin5 - 1] + 8u * lane, pre.bl);
    }
    CLK(pc, 0);

    // ---- joins
    JoinOut jo{false, false, false, 0u};
    if (!joins_on) {
    } else if (uni) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // rank among the staged starts of every table: seven descent steps in lock-step, no bounds test (sentinels)
        // and no branch (a table staged with 64 entries makes a step of zero at 64)
        uint32_t p[NT], A[NT], maskB[NT];
        const uint32_t Ab = sc.base + 4u * (uint32_t)v.joff[kJoin5 - 1];
        uint32_t pb = Ab - 8u;
        const uint64_t key_max = ((uint64_t)(uint32_t)c0 << 32) | (uint32_t)pos_max;
        bool miss = false;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            A[t] = sc.base + 4u * (uint32_t)v.joff[t];
            p[t] = A[t] - 4u;
            maskB[t] = 4u * (uint32_t)(v.jcap[t] - 1);
            // a staged slice that does not reach the tile's last variant: that table is searched in HBM (dense stretches)
            if (t > 0 || a.has_runs) miss |= lds_i32(A[t] + maskB[t]) < pos_max;
        }
        if (a.n_bl > 0) miss |= lds_u64(Ab + 8u * (kBlCap5 - 1)) < key_max;
#pragma unroll
        for (int sb = 256; sb >= 4; sb >>= 1) {
            uint32_t cand[NT];
            int x[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                cand[t] = p[t] + ((uint32_t)sb & maskB[t]);
                x[t] = lds_i32(cand[t]);
            }
            const uint32_t cb = pb + (uint32_t)((2 * sb) & (8 * (kBlCap5 - 1)));
            const uint64_t xk = lds_u64(cb);
#pragma unroll
            for (int t = 0; t < NT; ++t) p[t] = x[t] < pos ? cand[t] : p[t];
            pb = xk < key ? cb : pb;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t == 0 && !a.has_runs) continue;
            const uint32_t dE = 4u * (uint32_t)v.jcap[t];
            const int sg = L[t] + (int)((p[t] + 4u - A[t]) >> 2);   // staged starts below pos
            const uint32_t ps = p[t];                               // LDS address of starts[sg - 1]
            auto S = [&](int gi) { return lds_i32(ps + 4u * (uint32_t)(gi - sg + 1)); };
            auto E = [&](int gi) { return lds_i32(ps + dE + 4u * (uint32_t)(gi - sg + 1)); };
            interval_verdict(t, sg, plo[t], phi[t], pos, a.hpol_dist, S, E, jo);
        }
        if (a.n_bl > 0 && lds_u64(pb + 8u) == key) jo.cohort = true;   // the staged key at the rank (sentinel beyond the table)
        if (__ballot(miss) != 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t == 0 && !a.has_runs) continue;
                if (__ballot(lds_i32(A[t] + maskB[t]) < pos_max) != 0) join_one_global(&a, t, plo[t], phi[t], plo[t], phi[t], pos, key, &jo);
            }
            if (a.n_bl > 0 && __ballot(lds_u64(Ab + 8u * (kBlCap5 - 1)) < key_max) != 0)
                join_one_global(&a, kJoin5 - 1, 0, (int)a.n_bl, 0, 0, pos, key, &jo);
        }
    } else {
        // a tile that spans contigs: every lane searches its own contig's rows
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t == 0 && !a.has_runs) continue;
            const TrackView& tv = table_view(a, t);
            const int pl = tv.ptr[c], ph = tv.ptr[c + 1];
            join_one_global(&a, t, pl, ph, pl, ph, pos, key, &jo);
        }
        if (a.n_bl > 0) join_one_global(&a, kJoin5 - 1, 0, (int)a.n_bl, 0, 0, pos, key, &jo);
    }
    uint8_t flags = (uint8_t)(jo.trk << UGVC_FLAG_TRACK0_SHIFT);
    if (jo.cohort) flags |= UGVC_FLAG_COHORT_FP;
    if (a.mark_hpol && (jo.inside_run || jo.close_run)) flags |= UGVC_FLAG_HPOL_RUN;
    if (live) a.flags[i] = flags;
    CLK(pc, 2);
    if (!has_model) return;
    // (the window gather is consumed here, behind the joins: its round trip hides under the rank / join work)
    // ---- window: bases pos-5 .. pos+5 in bytes 0..10 of (w0, w1, w2)
    uint32_t w0 = __builtin_amdgcn_alignbyte(xw.y, xw.x, sh);
    uint32_t w1 = __builtin_amdgcn_alignbyte(xw.z, xw.y, sh);
    uint32_t w2 = __builtin_amdgcn_alignbyte(xw.w, xw.z, sh) & 0x00FFFFFFu;
    uint32_t gc_len = kGcWindow;
    if (__ballot(p0 < 5u || p0 + 6u > clen) != 0) {             // a lane near a contig edge: bases outside read as N
        uint32_t m[3] = {0, 0, 0};
        gc_len = 0;
#pragma unroll
        for (int q = 0; q < 11; ++q) {
            const bool inb = (uint32_t)(p0 - 5u + (uint32_t)q) < clen;        // wraps below 0 -> fails
            m[q >> 2] |= inb ? 0xFFu << (8 * (q & 3)) : 0u;
            if (q >= 1) gc_len += inb ? 1u : 0u;
        }
        w0 &= m[0]; w1 &= m[1]; w2 &= m[2];
    }
    // get_motif_around (5): left = pos-5 .. pos-1, right = pos+1 .. pos+5 (substitutions), base-5 codes
    const uint32_t b4 = w1 & 0xFFu, b6 = (w1 >> 16) & 0xFFu;
    const int lm = (int)(__builtin_amdgcn_udot4(w0, 0x00010519u, 0u, false) * 25u + __builtin_amdgcn_udot4(w0, 0x05000000u, b4, false));
    const int rm = (int)(__builtin_amdgcn_udot4(w1, 0x05190000u, w2 & 0xFFu, false) * 25u + __builtin_amdgcn_udot4(w2, 0x00010500u, 0u, false));
    const bool motif_n = any_zero_byte(w0) || any_zero_byte(w1 | 0x0000FF00u) || any_zero_byte(w2 | 0xFF000000u);
    // gc_content (10): bases pos-4 .. pos+5; everything that is not A / T counts (N included, as the reference's string test)
    const uint32_t n_at = (uint32_t)__popc(at_bytes(w0) & 0x01010100u) + (uint32_t)__popc(at_bytes(w1)) + (uint32_t)__popc(at_bytes(w2) & 0x00010101u);
    const uint32_t gc_code = lds_u16(sc.gcr_b + 2u * (gc_len * 11u + (gc_len - n_at)));       // group 0
    // cycle skip
    int css = 0;
    if (!(motif_n || rbase == 0 || abase == 0)) css = (int)lds_u8(sc.css_b + (((b4 - 1) << 6) | ((rbase - 1) << 4) | ((abase - 1) << 2) | (b6 - 1)));
    if (__ballot(rl > 1) != 0) {                                 // MNPs: the full flow-space walk
        if (rl > 1) {
            const uint8_t* __restrict__ apool = a.alleles;
            bool has_n = motif_n;
            for (int q = 0; q < rl; ++q) has_n |= apool[ro + q] == 0 || apool[ao + q] == 0;
            if (has_n) css = 0;
            else {
                auto wbyte = [&](int q) -> int { return (int)(((q < 4 ? w0 : (q < 8 ? w1 : w2)) >> (8 * (q & 3))) & 0xFFu); };
                auto seq_r = [&](int q) -> int {
                    if (q < kMotif) return wbyte(q);
                    if (q < kMotif + rl) return apool[ro + q - kMotif];
                    return wbyte(q - rl + 1);
                };
                auto seq_a = [&](int q) -> int {
                    if (q < kMotif) return wbyte(q);
                    if (q < kMotif + rl) return apool[ao + q - kMotif];
                    return wbyte(q - rl + 1);
                };
                css = cycle_skip_walk(rl + 2 * kMotif, a.flow, seq_r, seq_a);
            }
        }
    }

    CLK(pc, 1);

    // ---- codes -> the wave's code planes (the staged slices are dead: LDS executes a wave's accesses in order)
    __builtin_amdgcn_wave_barrier();
    const int hslot = ((lane & 31) << 1) | (lane >> 5);
    const uint32_t pl_b = sc.base + 2u * (uint32_t)hslot;
    lds_st16(pl_b + 128u * 0, cd[0]);
    lds_st16(pl_b + 128u * 1, cd[1]);
    lds_st16(pl_b + 128u * 2, raw_code(dp, v.cap5[0][2]));
    lds_st16(pl_b + 128u * 3, raw_code(adr, v.cap5[0][3]));
    lds_st16(pl_b + 128u * 4, raw_code(ada, v.cap5[0][4]));
    lds_st16(pl_b + 128u * 5, cd[2]);
    lds_st16(pl_b + 128u * 6, raw_code(gq, v.cap5[0][6]));
    if (v.used5[0] & 0x780u) {                                       // classify, indel_length, hmer length / base: 0
        lds_st16(pl_b + 128u * 7, 1u); lds_st16(pl_b + 128u * 8, 1u); lds_st16(pl_b + 128u * 9, 1u); lds_st16(pl_b + 128u * 10, 1u);
    }
    lds_st16(pl_b + 128u * 11, raw_code(lm, v.cap5[0][11]));
    lds_st16(pl_b + 128u * 12, raw_code(rm, v.cap5[0][12]));
    lds_st16(pl_b + 128u * 13, gc_code);
    lds_st16(pl_b + 128u * 14, raw_code(css, v.cap5[0][14]));
    lds_st16(pl_b + 128u * 15, jo.inside_run ? 2u : 1u);
    lds_st16(pl_b + 128u * 16, jo.close_run ? 2u : 1u);
#pragma unroll
    for (int t = 0; t < UGVC_MAX_TRACKS; ++t) lds_st16(pl_b + 128u * (17 + t), (jo.trk >> t) & 1u ? 2u : 1u);
    CLK(pc, 3);
}

// ---- indel tile: features of 64 length-changing variants -> raw-code records of groups 1 / 2 -------------
template <int NTRK>
__device__ __forceinline__ void featurize_indel_tile(const V5Args& v, const Scratch& sc, int64_t tile, int lane, uint32_t i, bool live, PhaseClk& pc) {
    constexpr int NT = 1 + NTRK;
    const FilterArgs& a = v.f;
    const uint8_t* __restrict__ apool = a.alleles;
    // ---- the tile's record and its table slices (independent of the columns: issued first)
    const bool joins_on = !(a.ablate & 524288);
    const int32_t* trec = v.br_indel + tile * kRecI5;
    const bool uni = cload(trec + 28) >= 0;
    int lo_[kJoin5], hi_[kJoin5], pl[NT], ph[NT];
    IndelPre<NT> pre;
    if (uni && joins_on) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            lo_[t] = hi_[t] = pl[t] = ph[t] = 0;
            pre.sv[t][0] = pre.sv[t][1] = pre.ev[t][0] = pre.ev[t][1] = 0;
            if (t == 0 && !a.has_runs) continue;
            const TrackView& tv = table_view(a, t);
            lo_[t] = cload(trec + t); hi_[t] = cload(trec + 8 + t);
            pl[t] = cload(trec + 16 + 2 * t); ph[t] = cload(trec + 17 + 2 * t);
            const int top = max(v.na[t] - 1, 0);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t gs = (uint32_t)max(min(lo_[t] - 2 + 64 * h + lane, top), 0);
                pre.sv[t][h] = tv.starts[gs]; pre.ev[t][h] = tv.ends[gs];
            }
        }
        lo_[kJoin5 - 1] = cload(trec + kJoin5 - 1); hi_[kJoin5 - 1] = cload(trec + 8 + kJoin5 - 1);
        pre.bl[0] = pre.bl[1] = ~0ull;
        if (a.n_bl > 0) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int64_t gi = (int64_t)lo_[kJoin5 - 1] + 64 * h + lane;
                if (gi < a.n_bl) pre.bl[h] = a.bl[gi];
            }
        }
    }
    const int c = a.contig[i], pos = a.pos[i], rl = a.ref_len[i], al = a.alt_len[i];
    const uint32_t ro = a.ref_off[i], ao = a.alt_off[i];
    // the model's own columns: needed last, issued first (every round trip of this tile is a dependent one)
    const float qual = a.qual[i], sor = a.sor[i];
    const int dp = a.dp[i], adr = a.ad_ref[i], ada = a.ad_alt[i], gq = a.gq[i];
    const bool ins = rl < al;
    const int classify = ins ? 1 : 2;
    const int indel_length = ins ? al - rl : rl - al;
    const int64_t clo = a.contig_off[c], chi = a.contig_off[c + 1];
    const uint32_t clen = (uint32_t)(chi - clo);
    const uint32_t p0 = (uint32_t)(pos - 1);
    const int64_t g0 = clo + p0;
    int64_t ws = (g0 - 6) & ~(int64_t)15;
    if (ws < 0) ws = 0;
    const int o0 = (int)(g0 - ws);                            // byte of the variant's first base, 6..21 (less at genome start)
    const uint32_t wrow_b = sc.base + (uint32_t)(lane * kWinRowB);
    {
        const uint4* src = reinterpret_cast<const uint4*>(a.ref + ws);
        const uint4 x0 = src[0], x1 = src[1], x2 = src[2];
        uint32_t w[kWinDw] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w, x2.x, x2.y, x2.z, x2.w};
        if (ws < clo || ws + kWinBytes > chi) {               // contig-edge lanes: bytes outside the contig read as N
#pragma unroll
            for (int q = 0; q < kWinDw; ++q) {
                uint32_t m = 0;
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const int64_t gi = ws + 4 * q + bb;
                    m |= (gi >= clo && gi < chi) ? (0xFFu << (8 * bb)) : 0u;
                }
                w[q] &= m;
            }
        }
#pragma unroll
        for (int q = 0; q < kWinDw; ++q) lds_st32(wrow_b + 4u * q, (int32_t)w[q]);
    }
    // allele bytes: the tail of the longer allele
    const uint32_t lo_off = ins ? ao : ro;
    const int ln = ins ? al : rl;
    uint32_t ab[8];
    ab[0] = apool[lo_off + 1];
    ab[1] = apool[lo_off + (2 < ln ? 2 : ln - 1)];
#pragma unroll
    for (int q = 2; q < 8; ++q) ab[q] = apool[lo_off + (q + 1 < ln ? q + 1 : ln - 1)];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    CLK(pc, 0);
    auto wb = [&](int o) -> int { return (int)lds_u8(wrow_b + (uint32_t)o); };
    auto ref_at = [&](int d) -> int {                         // reference base at contig offset p0 + d (0 outside the contig)
        const int o = o0 + d;
        if (o >= 0 && o < kWinBytes) return wb(o);
        const int64_t gi = g0 + d;
        return (gi >= clo && gi < chi) ? (int)a.ref[gi] : 0;
    };
    // ---- is_hmer_indel: the run starts at the first base after the variant's alleles
    const int d_so = ins ? 1 : rl;
    const int so = o0 + d_so;
    int hmer_len = 0, hmer_nuc = 0, run = 0;
    {
        const int bb = (int)ab[0];
        bool mono = true;
#pragma unroll
        for (int q = 1; q < 8; ++q) mono &= ab[q] == (uint32_t)bb;   // clamped reads repeat the last byte
        if (ln > 9)
            for (int q = 9; q < ln; ++q) mono &= apool[lo_off + q] == bb;
        const uint32_t pstart = p0 + (uint32_t)d_so;
        if (mono && pstart < clen) {
            if (so + 12 <= kWinBytes) {
                int nrun = 0;
                bool go = true;
#pragma unroll
                for (int q = 0; q < 12; ++q) {
                    go = go && wb(so + q) == bb;
                    nrun += go ? 1 : 0;
                }
                run = nrun;
                if (nrun == 12)
                    while (ref_at(d_so + run) == bb && pstart + (uint32_t)run < clen) ++run;
            } else {
                while (pstart + (uint32_t)run < clen && ref_at(d_so + run) == bb) ++run;
            }
            const uint32_t room = clen - pstart;               // an N run may not run past the contig end
            if ((uint32_t)run > room) run = (int)room;
            if (run > 0) {
                hmer_len = run + (ins ? 0 : rl - 1);
                hmer_nuc = bb;
            }
        }
    }
    const bool is_h = hmer_len > 0;
    const int group = is_h ? 1 : 2;
    const bool pg_ok = group == 1 ? v.pg[1].ok != 0 : v.pg[2].ok != 0;
    if (live && !pg_ok) {                                      // no model for this variant type: score 0, PASS
        a.score[i] = 0.f;
        a.filter[i] = UGVC_FILTER_PASS;
    }
    // ---- record slots: one returning atomic per wave and group (issued here: its round trip runs under the joins)
    const bool mine = live && pg_ok;
    const unsigned long long m1 = __ballot(mine && group == 1), m2 = __ballot(mine && group == 2);
    const int shard = (int)(tile & (kShards - 1));
    unsigned got = 0;
    if (lane == 1 && m1 != 0) got = atomicAdd(&v.counters[(1 * kShards + shard) * kCounterStride], (unsigned)__popcll(m1));
    if (lane == 2 && m2 != 0) got = atomicAdd(&v.counters[(2 * kShards + shard) * kCounterStride], (unsigned)__popcll(m2));
    const unsigned long long below = (1ull << lane) - 1;
    const unsigned grank = (unsigned)__popcll((group == 1 ? m1 : m2) & below);

    // ---- get_motif_around (5), gc_content (10)
    int W[11];
#pragma unroll
    for (int q = 0; q < 11; ++q) W[q] = wb(o0 - 5 + q);
    if (o0 < 5) {                                              // genome start: the window begins at base 0
#pragma unroll
        for (int q = 0; q < 11; ++q) W[q] = ref_at(q - 5);
    }
    const int d_r = is_h ? d_so + run : rl;
    int lm = 0, rm = 0;
#pragma unroll
    for (int q = 0; q < kMotif; ++q) {
        const int rb = (o0 + d_r + kMotif <= kWinBytes) ? wb(o0 + d_r + q) : ref_at(d_r + q);
        lm = lm * 5 + W[q + 1];
        rm = rm * 5 + rb;
    }
    int gc_cnt = 0, gc_len = 0;
#pragma unroll
    for (int q = 0; q < kGcWindow; ++q) {
        const uint32_t pw = p0 + 1 - kGcWindow / 2 + q;       // wraps below 0 -> fails the bound test
        const bool inb = pw < clen;
        const int bb = W[q + 1];
        gc_len += inb;
        gc_cnt += inb && bb != 1 && bb != 4;
    }
    const uint32_t gc_idx = (uint32_t)(gc_len * 11 + gc_cnt);

    CLK(pc, 1);
    // ---- joins: the staged slices, two tables at a time in the scratch the window rows have left
    const uint64_t key = ((uint64_t)(uint32_t)c << 32) | (uint32_t)pos;
    JoinOut jo{false, false, false, 0u};
    if (!joins_on) {
    } else if (uni) {
        const uint32_t s0 = sc.base, s1 = sc.base + kIndelSlotB;
        auto fits = [&](int t) { return hi_[t] - lo_[t] + 4 <= kIndelRows - 1; };
        auto wide = [&](int t) { return !fits(t) && hi_[t] - lo_[t] + 4 <= kWideRows - 1; };
        // the first table that needs the wide slice: its rows are requested now and arrive under the narrow rounds
        int tw = -1;
        int wv[kWideChunks], we[kWideChunks];
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (tw < 0 && !(t == 0 && !a.has_runs) && wide(t)) {
                tw = t;
                wide_load(table_view(a, t), lo_[t] - 2, max(v.na[t] - 1, 0), lane, wv, we);
            }
        // round A: runs | blacklist
        const bool runs_on = a.has_runs != 0, bl_on = a.n_bl > 0;
        const bool bl_fit = hi_[kJoin5 - 1] - lo_[kJoin5 - 1] + 1 <= kIndelRows - 1;
        __builtin_amdgcn_wave_barrier();
        if (runs_on && fits(0)) stage_rows(s0, lo_[0] - 2, pl[0], ph[0], pre.sv[0], pre.ev[0], lane);
        if (bl_on && bl_fit) {
            lds_st64(s1 + 8u * lane, pre.bl[0]);
            lds_st64(s1 + 512u + 8u * lane, pre.bl[1]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (runs_on) {
            if (fits(0)) staged_verdict(a, s0, 0, lo_[0] - 2, pl[0], ph[0], pos, jo);
        }
        if (bl_on) {
            if (bl_fit) {
                uint32_t pb = s1 - 8u;
#pragma unroll
                for (int sb = 512; sb >= 8; sb >>= 1) {
                    const uint32_t cand = pb + (uint32_t)sb;
                    pb = lds_u64(cand) < key ? cand : pb;
                }
                if (lds_u64(pb + 8u) == key) jo.cohort = true;
            } else join_one_global(&a, kJoin5 - 1, lo_[kJoin5 - 1], hi_[kJoin5 - 1], 0, 0, pos, key, &jo);
        }
        // tracks, in pairs
#pragma unroll
        for (int t = 1; t < NT; t += 2) {
            __builtin_amdgcn_wave_barrier();
            if (fits(t)) stage_rows(s0, lo_[t] - 2, pl[t], ph[t], pre.sv[t], pre.ev[t], lane);
            if (t + 1 < NT && fits(t + 1)) stage_rows(s1, lo_[t + 1] - 2, pl[t + 1], ph[t + 1], pre.sv[t + 1 < NT ? t + 1 : t], pre.ev[t + 1 < NT ? t + 1 : t], lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (fits(t)) staged_verdict(a, s0, t, lo_[t] - 2, pl[t], ph[t], pos, jo);
            if (t + 1 < NT && fits(t + 1)) staged_verdict(a, s1, t + 1, lo_[t + 1] - 2, pl[t + 1 < NT ? t + 1 : t], ph[t + 1 < NT ? t + 1 : t], pos, jo);
        }
        // the tables the narrow slices could not hold
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if ((t == 0 && !a.has_runs) || fits(t)) continue;
            if (wide(t)) {
                if (t != tw) wide_load(table_view(a, t), lo_[t] - 2, max(v.na[t] - 1, 0), lane, wv, we);
                wide_verdict(a, s0, t, lo_[t] - 2, pl[t], ph[t], pos, lane, wv, we, jo);
            } else join_one_global(&a, t, lo_[t], hi_[t], pl[t], ph[t], pos, key, &jo);
        }
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t == 0 && !a.has_runs) continue;
            const TrackView& tv = table_view(a, t);
            const int pl_ = tv.ptr[c], ph_ = tv.ptr[c + 1];
            join_one_global(&a, t, pl_, ph_, pl_, ph_, pos, key, &jo);
        }
        if (a.n_bl > 0) join_one_global(&a, kJoin5 - 1, 0, (int)a.n_bl, 0, 0, pos, key, &jo);
    }
    uint8_t flags = (uint8_t)(jo.trk << UGVC_FLAG_TRACK0_SHIFT);
    if (jo.cohort) flags |= UGVC_FLAG_COHORT_FP;
    if (a.mark_hpol && (jo.inside_run || jo.close_run)) flags |= UGVC_FLAG_HPOL_RUN;
    if (live) a.flags[i] = flags;
    CLK(pc, 2);
    // ---- codes of the lane's own group
    const float vaf = dp > 0 ? __fdiv_rn((float)ada, (float)dp) : 0.0f;
    uint32_t cd[4] = {0, 0, 0, 0};
    if (__ballot(mine) != 0) {
        const float fx[3] = {qual, sor, vaf};
        uint32_t off3[3], len3[3], c3[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const uint2 d = lds_u32x2(sc.gtab_b + 64u + (uint32_t)(group - 1) * 24u + 8u * s);
            off3[s] = d.x;
            len3[s] = d.y;
        }
        rank3_sorted(fx, sc.thr_b, off3, len3, v.thr_bits, c3);
        cd[0] = c3[0]; cd[1] = c3[1]; cd[2] = c3[2];
        cd[3] = lds_u16(sc.gcr_b + 2u * ((uint32_t)group * kGcRank + gc_idx));
    }
    CLK(pc, 6);
    const uint32_t gcap_b = sc.gtab_b + (uint32_t)(group - 1) * 32u;
    auto rc = [&](int f, int x) -> uint32_t { return raw_code(x, (int)lds_u16(gcap_b + 2u * f)); };
    uint32_t r[kRec5Dwords];
    r[0] = cd[0] | (cd[1] << 16);                                        // qual, sor
    r[1] = rc(2, dp) | (rc(3, adr) << 16);
    r[2] = rc(4, ada) | (cd[2] << 16);                                   // ad_alt, vaf
    r[3] = rc(6, gq) | (rc(7, classify) << 16);
    r[4] = rc(8, indel_length) | (rc(9, hmer_len) << 16);
    r[5] = rc(10, hmer_nuc) | (rc(11, lm) << 16);
    r[6] = rc(12, rm) | (cd[3] << 16);                                   // right motif, gc
    r[7] = rc(14, 3) | ((jo.inside_run ? 2u : 1u) << 16);                // cycle skip: NA for indels
    r[8] = (jo.close_run ? 2u : 1u) | (((jo.trk >> 0) & 1u ? 2u : 1u) << 16);
    r[9] = ((jo.trk >> 1) & 1u ? 2u : 1u) | (((jo.trk >> 2) & 1u ? 2u : 1u) << 16);
    r[10] = ((jo.trk >> 3) & 1u ? 2u : 1u) | (((jo.trk >> 4) & 1u ? 2u : 1u) << 16);
    r[11] = i;
    const unsigned b1 = __shfl(got, 1), b2 = __shfl(got, 2);
    if (mine) {
        uint4* dst = v.rec5[group] + ((size_t)shard * v.shard_cap5 + (group == 1 ? b1 : b2) + grank) * 3;
        dst[0] = make_uint4(r[0], r[1], r[2], r[3]);
        dst[1] = make_uint4(r[4], r[5], r[6], r[7]);
        dst[2] = make_uint4(r[8], r[9], r[10], r[11]);
    }
    CLK(pc, 3);
}

// ---- single-sum walk of one forest over the wave's code planes -> (tree_score, FILTER) -------------------
template <int NTM>
__device__ __forceinline__ void walk_forest(const PackedGroupView& pg, uint32_t hi_b, uint32_t last_b, uint32_t p1_b,
                                            uint32_t planes_lane_b, float& score, uint8_t& filt) {
    const int T = pg.T, D = pg.D, H = (1 << D) >> 1;
    double a1 = 0.0;
    int t = 0;
    if (NTM > 8) {
        // more trees in flight per lane: in the fused kernel only part of a CU's waves walk at any time, so a walking
        // wave has to keep more LDS requests outstanding to fill the pipeline
        for (; t + NTM <= T; t += NTM) {
            uint32_t pi[NTM];
            walk4<NTM>(hi_b, last_b, planes_lane_b, t, D, H, pi);
            double pv[NTM];
#pragma unroll
            for (int q = 0; q < NTM; ++q) pv[q] = lds_f64(p1_b + 8u * pi[q]);
#pragma unroll
            for (int q = 0; q < NTM; ++q) a1 += pv[q];
        }
    }
    for (; t + 8 <= T; t += 8) {
        uint32_t pi[8];
        walk4<8>(hi_b, last_b, planes_lane_b, t, D, H, pi);
        double pv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) pv[q] = lds_f64(p1_b + 8u * pi[q]);
#pragma unroll
        for (int q = 0; q < 8; ++q) a1 += pv[q];
    }
    for (; t < T; ++t) {
        uint32_t pi[1];
        walk4<1>(hi_b, last_b, planes_lane_b, t, D, H, pi);
        a1 += lds_f64(p1_b + 8u * pi[0]);
    }
    const double half = 0.5 * (double)T, band = pg.band;
    score = (float)(a1 / (double)T);
    filt = a1 > half ? UGVC_FILTER_PASS : UGVC_FILTER_LOW_SCORE;
    // inside the band around T/2 (rounding of the two class sums, model_pack.hip) the class-0 sum decides as
    // scikit-learn's argmax does: redo the walk with both payload sums, in tree order (exact ties in practice)
    if (__builtin_amdgcn_ballot_w64(fabs(a1 - half) <= band) != 0) {
        double b0 = 0.0, b1 = 0.0;
        for (int tt = 0; tt < T; ++tt) {
            uint32_t pi[1];
            walk4<1>(hi_b, last_b, planes_lane_b, tt, D, H, pi);
            const double2 pv = pg.pairs[pi[0]];
            b0 += pv.x; b1 += pv.y;
        }
        const double q0 = b0 / (double)T, q1 = b1 / (double)T;
        if (fabs(a1 - half) <= band) filt = q1 > q0 ? UGVC_FILTER_PASS : UGVC_FILTER_LOW_SCORE;
    }
}

// LDS of a workgroup: group forest (hi | last | p1, 16-byte padded) | group 0's level-order threshold trees | the
// indel groups' sorted thresholds (skewed) | gc rank codes | css | wave scratch
struct Lds5 {
    uint32_t hi_b, last_b, p1_b, eyt_b, thr_b, gcr_b, css_b, gtab_b, scratch_b;
};

__device__ __forceinline__ Lds5 lds5_fill(unsigned char* smem, const V5Args& v, bool with_forest, int tid, int nthreads) {
    const PackedGroupView& pg = v.pg[0];
    Lds5 L;
    size_t off = 0;
    const size_t n_hi = with_forest ? ((size_t)pg.T << pg.D) / 2 : 0;
    const size_t b_hi = (n_hi * 4 + 15) & ~(size_t)15, b_last = (n_hi * 8 + 15) & ~(size_t)15;
    const size_t b_p1 = with_forest ? (((size_t)pg.n_pairs * 8 + 15) & ~(size_t)15) : 0;
    L.hi_b = lds_addr(smem);
    L.last_b = lds_addr(smem + b_hi);
    L.p1_b = lds_addr(smem + b_hi + b_last);
    off = b_hi + b_last + b_p1;
    if (with_forest) {
        const uint4* s0 = reinterpret_cast<const uint4*>(pg.hi4);
        const uint4* s1 = reinterpret_cast<const uint4*>(pg.last4);
        const uint4* s2 = reinterpret_cast<const uint4*>(pg.p1);
        uint4* d0 = reinterpret_cast<uint4*>(smem);
        uint4* d1 = reinterpret_cast<uint4*>(smem + b_hi);
        uint4* d2 = reinterpret_cast<uint4*>(smem + b_hi + b_last);
        const size_t n0 = b_hi / 16, n1 = b_last / 16, n2 = b_p1 / 16;
        for (size_t q = tid; q < n0 + n1 + n2; q += nthreads) {
            if (q < n0) d0[q] = s0[q];
            else if (q < n0 + n1) d1[q - n0] = s1[q - n0];
            else d2[q - n0 - n1] = s2[q - n0 - n1];
        }
    }
    float* eyt_l = reinterpret_cast<float*>(smem + off);
    for (int q = tid; q < v.eyt_len / 4; q += nthreads) reinterpret_cast<float4*>(eyt_l)[q] = reinterpret_cast<const float4*>(v.eyt)[q];
    L.eyt_b = lds_addr(eyt_l);
    off += (size_t)v.eyt_len * 4;
    float* thr_l = reinterpret_cast<float*>(smem + off);
    const int n_thr = v.thr_lds_len - v.thr0_len;                                          // groups 1 and 2
    for (int q = tid; q < n_thr; q += nthreads) thr_l[q + (q >> 5)] = v.thr[v.thr0_len + q];   // skewed: element j at j + (j >> 5)
    L.thr_b = lds_addr(thr_l);
    off += ((size_t)(n_thr + (n_thr >> 5) + 1) * 4 + 15) & ~(size_t)15;
    uint16_t* gcr = reinterpret_cast<uint16_t*>(smem + off);
    for (int q = tid; q < UGVC_N_GROUPS * kGcRank; q += nthreads) {
        const int g = q / kGcRank, r = q % kGcRank, len = r / 11, cnt = r % 11;
        const float f = (len > 0 && cnt <= len) ? (float)((double)cnt / (double)len) : 0.0f;
        const uint2 d = v.desc3[g * kMaxFeatures + 13];
        const float* t = v.thr + (d.x & 0xFFFFFu);
        const int n = (int)(d.y & 0xFFFFu);
        int rank = 0;
        for (int e = 0; e < n; ++e) rank += t[e] < f ? 1 : 0;
        gcr[q] = (uint16_t)rank;
    }
    L.gcr_b = lds_addr(gcr);
    off += kGcRankBytes;
    uint8_t* css = smem + off;
    for (int q = tid; q < 256; q += nthreads) css[q] = v.css_lut[q];
    L.css_b = lds_addr(css);
    off += 256;
    // per indel group: the clamps of the integer features (u16 [2][16]) and the (offset, length) of the three float
    // features' threshold slices in the staged table - read per lane by its group instead of scalar loads + selects
    unsigned char* gtab = smem + off;
    if (tid < 32) reinterpret_cast<uint16_t*>(gtab)[tid] = (uint16_t)v.cap5[1 + (tid >> 4)][tid & 15];
    if (tid >= 64 && tid < 70) {
        const int q = tid - 64, g = 1 + q / 3, sfeat = q % 3;
        const int fj = sfeat == 0 ? 0 : (sfeat == 1 ? 1 : 5);
        const uint2 d = v.desc3[g * kMaxFeatures + fj];
        reinterpret_cast<uint32_t*>(gtab + 64)[2 * q] = (d.x & 0xFFFFFu) - (uint32_t)v.thr0_len;      // the staged table starts at group 1
        reinterpret_cast<uint32_t*>(gtab + 64)[2 * q + 1] = d.y & 0xFFFFu;
    }
    L.gtab_b = lds_addr(gtab);
    off += kGtabBytes;
    L.scratch_b = lds_addr(smem + off);
    return L;
}

static size_t lds5_bytes(const V5Args& v, int n_waves) {
    const PackedGroupView& pg = v.pg[0];
    const bool with_forest = pg.ok != 0;
    const size_t n_hi = with_forest ? ((size_t)pg.T << pg.D) / 2 : 0;
    size_t b = ((n_hi * 4 + 15) & ~(size_t)15) + ((n_hi * 8 + 15) & ~(size_t)15);
    if (with_forest) b += ((size_t)pg.n_pairs * 8 + 15) & ~(size_t)15;
    b += (size_t)v.eyt_len * 4;
    const int n_thr = v.thr_lds_len - v.thr0_len;
    b += ((size_t)(n_thr + (n_thr >> 5) + 1) * 4 + 15) & ~(size_t)15;
    b += kGcRankBytes + 256 + kGtabBytes;
    const int n_iw = std::min(v.n_indel_waves, n_waves - 1);
    return b + (size_t)(n_waves - n_iw) * v.scratch_bytes + (size_t)n_iw * v.scratch_indel;
}

// Tile of a wave's k-th unit of work: the 64 shard counters of a class are scanned once per wave (lane s holds
// the inclusive count of shards 0..s); virtual index -> (shard, slot in the shard).
__device__ __forceinline__ int64_t tile_of(int64_t vidx, unsigned incl, int shard_tiles) {
    const int shard = (int)__popcll(__ballot(incl <= (unsigned)vidx));
    const unsigned excl = shard > 0 ? (unsigned)__builtin_amdgcn_readlane((int)incl, shard - 1) : 0u;
    return (int64_t)shard * shard_tiles + ((unsigned)vidx - excl);
}

// ---- Kf: persistent, one workgroup per CU; every wave works through tiles on its own --------------------
template <int NTRK, int NTW>
__global__ __launch_bounds__(kK2Threads) void fused5_kernel(const V5Args v) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = rfl(tid >> 6);
    const int n_waves = blockDim.x >> 6;
    const PackedGroupView& pg0 = v.pg[0];
    const bool has0 = pg0.ok != 0;
    // the thresholds of every group: SNP tiles rank against group 0's slices (the head of the table), indel
    // tiles against their own group's
    const Lds5 L = lds5_fill(smem, v, has0, tid, blockDim.x);
    __syncthreads();
    Scratch sc;
    sc.eyt_b = L.eyt_b; sc.thr_b = L.thr_b; sc.gcr_b = L.gcr_b; sc.css_b = L.css_b; sc.gtab_b = L.gtab_b;
    const int hslot = ((lane & 31) << 1) | (lane >> 5);
    // inclusive scans of the shard counters of both classes
    unsigned incl_s = v.tile_cnt[lane * kTileCntStride5], incl_i = v.tile_cnt[(kTileShards5 + lane) * kTileCntStride5];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned ys = __shfl_up(incl_s, d), yi = __shfl_up(incl_i, d);
        if (lane >= d) { incl_s += ys; incl_i += yi; }
    }
    const int64_t ns = (unsigned)__builtin_amdgcn_readlane((int)incl_s, 63);
    const int64_t ni = (v.f.ablate & 262144) ? 0 : (unsigned)__builtin_amdgcn_readlane((int)incl_i, 63);
    // Wave roles: the last `n_iw` waves of a workgroup featurise the indel tiles (memory-latency bound, no walk),
    // the others run the SNP pipeline: a tile's row indices and columns are fetched ONE TILE AHEAD (indices at the
    // top of the previous tile, columns just before its walk), so a tile starts with its window / allele / slice
    // gathers instead of two dependent round trips.  Work slots are wave-major over the workgroups: a short last
    // round leaves a few waves busy on every CU.
    // the LDS layout gives the last `n_indel_waves` waves the larger (window-row) scratch; how many of them actually work
    // on indel tiles follows the callset's class mix (an SNV-only callset has none: every wave runs the SNP pipeline)
    const int n_big = v.n_indel_waves, n_small = n_waves - n_big;
    sc.base = L.scratch_b + (uint32_t)(wave < n_small ? wave * v.scratch_bytes : n_small * v.scratch_bytes + (wave - n_small) * v.scratch_indel);
    const int want_iw = ni > 0 ? (int)((n_waves * ni * 13 + (ns + ni) * 10 - 1) / ((ns + ni) * 10)) : 0;      // ceil(1.3 x share of tiles)
    const int n_iw = want_iw < n_big ? want_iw : n_big, n_sw = n_waves - n_iw;
    const uint32_t planes_lane_b = sc.base + 2u * (uint32_t)hslot;
    // A featurize phase is a short instruction stream between long memory waits; the walk is a long stream that waits
    // on LDS.  Featurize runs at raised priority (kernel variant bit 29 turns that off): it wins the SIMD's issue
    // arbitration against the walking waves, gets back to walking sooner, and the walkers lose slots they would
    // mostly have spent waiting.
    const bool prio = !(v.f.ablate & (1 << 29));
    if (wave >= n_sw) {
        if (prio) __builtin_amdgcn_s_setprio(2);
        const int64_t stride = (int64_t)gridDim.x * n_iw;
        PhaseClk pc{};
#ifdef UGVC_PHASE_CLOCK
        pc.last = __builtin_readcyclecounter();
        const uint64_t t_begin = pc.last;
        int n_done = 0;
#endif
        for (int64_t ti = (int64_t)(wave - n_sw) * gridDim.x + blockIdx.x; ti < ni; ti += stride) {
            const int64_t tile = tile_of(ti, incl_i, v.shard_tiles);
            const uint32_t id = v.indel_idx[tile * 64 + lane];
            const bool live = id != ~0u;
            const uint32_t id0 = (uint32_t)rfl((int)id);       // (outside the select: a ternary would read the first PADDING lane)
            featurize_indel_tile<NTRK>(v, sc, tile, lane, live ? id : id0, live, pc);
            __builtin_amdgcn_wave_barrier();
#ifdef UGVC_PHASE_CLOCK
            ++n_done;
#endif
        }
#ifdef UGVC_PHASE_CLOCK
        if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 133) && wave == n_waves - 1)
            printf("iclk b%d w%d tiles %d total %llu | load+window %llu hmer+motif %llu joins %llu codes+record %llu (ranks %llu)\n", (int)blockIdx.x, wave, n_done,
                   (unsigned long long)(pc.last - t_begin), (unsigned long long)pc.acc[0], (unsigned long long)pc.acc[1], (unsigned long long)pc.acc[2],
                   (unsigned long long)pc.acc[3], (unsigned long long)pc.acc[6]);
#endif
        return;
    }
    const int64_t stride = (int64_t)gridDim.x * n_sw;
    int64_t ts = (int64_t)wave * gridDim.x + blockIdx.x;
    if (ts >= ns) return;
    auto fetch_ids = [&](int64_t vidx, uint32_t& id, uint32_t& i, bool& live) {
        id = v.snp_idx[tile_of(vidx, incl_s, v.shard_tiles) * 64 + lane];
        live = id != ~0u;
        const uint32_t id0 = (uint32_t)rfl((int)id);
        i = live ? id : id0;
    };
    uint32_t id, i;
    bool live;
    fetch_ids(ts, id, i, live);
    SnpCols cols = load_snp_cols(v.f, i);
    constexpr int NT = 1 + NTRK;
    const bool joins_on = !(v.f.ablate & 524288);
    uint32_t rec = (uint32_t)v.br_snp[tile_of(ts, incl_s, v.shard_tiles) * kRecS5 + (lane & (kRecS5 - 1))];
    SlicePre<NT> pre;
    issue_slices<NT>(v, joins_on && (int)__builtin_amdgcn_readlane((int)rec, 7) >= 0 ? rec : 0u, lane, pre);
    PhaseClk pc{};
#ifdef UGVC_PHASE_CLOCK
    pc.last = __builtin_readcyclecounter();
    const uint64_t t_begin = pc.last;
    int n_done = 0;
#endif
    for (; ts < ns; ts += stride) {
        const int64_t tile = tile_of(ts, incl_s, v.shard_tiles);
        const bool more = ts + stride < ns;
        uint32_t id_n = 0, i_n = 0;
        bool live_n = false;
        if (prio) __builtin_amdgcn_s_setprio(2);
        uint32_t rec_n = 0;
        if (more) {
            const int64_t tile_n = tile_of(ts + stride, incl_s, v.shard_tiles);
            id_n = v.snp_idx[tile_n * 64 + lane];                                                // consumed after the joins
            rec_n = (uint32_t)v.br_snp[tile_n * kRecS5 + (lane & (kRecS5 - 1))];
        }
        featurize_snp_tile<NTRK>(v, sc, tile, lane, i, live, has0, cols, rec, pre, pc);
        SnpCols cols_n = cols;
        if (more) {
            live_n = id_n != ~0u;
            const uint32_t id0 = (uint32_t)rfl((int)id_n);
            i_n = live_n ? id_n : id0;
            cols_n = load_snp_cols(v.f, i_n);                   // in flight during the walk
            if (joins_on && (int)__builtin_amdgcn_readlane((int)rec_n, 7) >= 0) issue_slices<NT>(v, rec_n, lane, pre);   // likewise
        }
        CLK(pc, 4);
        if (prio) __builtin_amdgcn_s_setprio(0);
        if (has0) {
            float score = 0.f;
            uint8_t filt = UGVC_FILTER_PASS;
            if (!(v.f.ablate & 131072)) walk_forest<NTW>(pg0, L.hi_b, L.last_b, L.p1_b, planes_lane_b, score, filt);
            if (live) {
                v.f.score[i] = score;
                v.f.filter[i] = filt;
            }
        } else if (live) {                                     // no model for substitutions: score 0, PASS
            v.f.score[i] = 0.f;
            v.f.filter[i] = UGVC_FILTER_PASS;
        }
        cols = cols_n; i = i_n; live = live_n; rec = rec_n;
        __builtin_amdgcn_wave_barrier();
        CLK(pc, 5);
#ifdef UGVC_PHASE_CLOCK
        ++n_done;
#endif
    }
#ifdef UGVC_PHASE_CLOCK
    if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 133) && (wave == 0 || wave == 7))
        printf("clk b%d w%d tiles %d total %llu | stage %llu window %llu joins %llu codes %llu cols_n %llu walk %llu\n", (int)blockIdx.x, wave, n_done,
               (unsigned long long)(pc.last - t_begin), (unsigned long long)pc.acc[0], (unsigned long long)pc.acc[1], (unsigned long long)pc.acc[2],
               (unsigned long long)pc.acc[3], (unsigned long long)pc.acc[4], (unsigned long long)pc.acc[5]);
    if (lane == 0 && blockIdx.x == 0 && wave == 0) printf("  issue %llu eyt %llu\n", (unsigned long long)pc.acc[6], (unsigned long long)pc.acc[7]);
#endif
}

// ---- K2: the indel groups' forests over the raw-code records ---------------------------------------------
__global__ __launch_bounds__(kK2Threads) void forest5_kernel(const V5Args v) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned shard_off[kShards + 1];
    __shared__ unsigned totals[UGVC_N_GROUPS];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int n_waves = blockDim.x >> 6;
    if (tid < UGVC_N_GROUPS) totals[tid] = 0;
    __syncthreads();
    for (int q = tid; q < UGVC_N_GROUPS * kShards; q += blockDim.x) {
        const unsigned cshard = v.counters[q * kCounterStride];
        if (cshard && q >= kShards) atomicAdd(&totals[q / kShards], cshard);
    }
    __syncthreads();
    // workgroups are split over the two indel groups in proportion to count x trees x depth
    const int B = gridDim.x;
    unsigned cnt[UGVC_N_GROUPS];
    double work[UGVC_N_GROUPS], tot = 0.0;
    cnt[0] = 0; work[0] = 0.0;
    for (int g = 1; g < UGVC_N_GROUPS; ++g) {
        cnt[g] = v.pg[g].ok ? totals[g] : 0u;
        work[g] = (double)cnt[g] * v.pg[g].T * v.pg[g].D;
        tot += work[g];
    }
    if (tot == 0.0) return;
    int nb[UGVC_N_GROUPS] = {0, 0, 0}, used = 0, big = 1;
    for (int g = 1; g < UGVC_N_GROUPS; ++g) {
        nb[g] = cnt[g] ? (int)(B * (work[g] / tot) + 0.5) : 0;
        if (cnt[g] && nb[g] < 1) nb[g] = 1;
        used += nb[g];
        if (work[g] > work[big]) big = g;
    }
    nb[big] += B - used;
    if (nb[big] < 1) return;
    int g = 1, lb = blockIdx.x;
    while (g < UGVC_N_GROUPS - 1 && lb >= nb[g]) { lb -= nb[g]; ++g; }
    g = rfl(g);
    lb = rfl(lb);
    const int nbg = rfl(nb[g]);
    const PackedGroupView pg = v.pg[g];
    const unsigned n = (unsigned)rfl((int)cnt[g]);
    if (n == 0 || nbg == 0) return;
    if (wave == 0) {                                             // exclusive scan of the group's shard counts
        unsigned x[4], s = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { x[q] = v.counters[(g * kShards + lane * 4 + q) * kCounterStride]; s += x[q]; }
        unsigned incl = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned y = __shfl_up(incl, d);
            if (lane >= d) incl += y;
        }
        unsigned runv = incl - s;
#pragma unroll
        for (int q = 0; q < 4; ++q) { shard_off[lane * 4 + q] = runv; runv += x[q]; }
        if (lane == 63) shard_off[kShards] = runv;
    }
    const int D = pg.D, H = (1 << D) >> 1;
    const size_t n_hi = (size_t)pg.T * H;
    const size_t b_hi = (n_hi * 4 + 15) & ~(size_t)15, b_last = (n_hi * 8 + 15) & ~(size_t)15;
    const size_t b_p1 = ((size_t)pg.n_pairs * 8 + 15) & ~(size_t)15;
    {
        const uint4* s0 = reinterpret_cast<const uint4*>(pg.hi4);
        const uint4* s1 = reinterpret_cast<const uint4*>(pg.last4);
        const uint4* s2 = reinterpret_cast<const uint4*>(pg.p1);
        uint4* d0 = reinterpret_cast<uint4*>(smem);
        uint4* d1 = reinterpret_cast<uint4*>(smem + b_hi);
        uint4* d2 = reinterpret_cast<uint4*>(smem + b_hi + b_last);
        const size_t n0 = b_hi / 16, n1 = b_last / 16, n2 = b_p1 / 16;
        for (size_t q = tid; q < n0 + n1 + n2; q += blockDim.x) {
            if (q < n0) d0[q] = s0[q];
            else if (q < n0 + n1) d1[q - n0] = s1[q - n0];
            else d2[q - n0 - n1] = s2[q - n0 - n1];
        }
    }
    __syncthreads();
    const uint32_t hi_b = lds_addr(smem), last_b = lds_addr(smem + b_hi), p1_b = lds_addr(smem + b_hi + b_last);
    const int hslot = ((lane & 31) << 1) | (lane >> 5);
    const uint32_t planes_b = lds_addr(smem + b_hi + b_last + b_p1) + (uint32_t)(wave * kMaxFeatures * 128);
    const uint32_t planes_lane_b = planes_b + 2u * (uint32_t)hslot;
    const unsigned waves = (unsigned)nbg * n_waves;
    const uint4* __restrict__ rec = v.rec5[g];
    auto fetch = [&](unsigned chunk, bool& live, uint4& r0, uint4& r1, uint4& r2) {
        const unsigned r = chunk * 64 + lane;
        live = r < n;
        const unsigned rr = live ? r : n - 1;
        int lo = 0, len = kShards;
        while (len > 1) {                                        // shard of record rr
            const int half = len >> 1;
            const bool ge = shard_off[lo + half] <= rr;
            lo = ge ? lo + half : lo;
            len = ge ? len - half : half;
        }
        const uint4* src = rec + ((size_t)lo * v.shard_cap5 + (rr - shard_off[lo])) * 3;
        r0 = src[0]; r1 = src[1]; r2 = src[2];
    };
    unsigned chunk = (unsigned)rfl(wave) * (unsigned)nbg + (unsigned)lb;
    bool live_next = false;
    uint4 n0 = make_uint4(0, 0, 0, 0), n1 = n0, n2 = n0;
    if ((uint64_t)chunk * 64 < n) fetch(chunk, live_next, n0, n1, n2);
    for (; (uint64_t)chunk * 64 < n; chunk += waves) {
        const uint4 q0 = n0, q1 = n1, q2 = n2;
        const bool live = live_next;
        if ((uint64_t)(chunk + waves) * 64 < n) fetch(chunk + waves, live_next, n0, n1, n2);
        // record dword d holds the codes of features 2 d and 2 d + 1; one 16-bit store per plane
        const uint32_t w[11] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z};
#pragma unroll
        for (int d = 0; d < 11; ++d) {
            lds_st16(planes_lane_b + 128u * (2 * d), w[d] & 0xFFFFu);
            lds_st16(planes_lane_b + 128u * (2 * d + 1), w[d] >> 16);
        }
        float score;
        uint8_t filt;
        walk_forest<8>(pg, hi_b, last_b, p1_b, planes_lane_b, score, filt);
        if (live) {
            v.f.score[q2.w] = score;
            v.f.filter[q2.w] = filt;
        }
    }
}

static size_t k5_forest_lds(const PackedGroupView& pg, int n_waves) {
    const size_t n_hi = ((size_t)pg.T << pg.D) / 2;
    return ((n_hi * 4 + 15) & ~(size_t)15) + ((n_hi * 8 + 15) & ~(size_t)15) + (((size_t)pg.n_pairs * 8 + 15) & ~(size_t)15) +
           (size_t)n_waves * kMaxFeatures * 128;
}

// LDS budget of the fused kernel for this configuration: 16, 12 or 8 waves of scratch beside the SNP forest
int v5_fused_waves(const V5Args& v) {
    for (int w : {16, 12, 8}) {
        if (lds5_bytes(v, w) <= 158 * 1024) return w;
    }
    return 0;
}

using K5 = void (*)(const V5Args);
// 16 trees in flight per lane (8 measured 4 % slower: 506 vs 485 us; kernel variant bit 28 selects 8 for the 3-track kernel)
static K5 fused5_for(int n_tracks, bool narrow = false) {
    if (narrow && n_tracks == 3) return fused5_kernel<3, 8>;
    switch (n_tracks) {
        case 0: return fused5_kernel<0, 16>;
        case 1: return fused5_kernel<1, 16>;
        case 2: return fused5_kernel<2, 16>;
        case 3: return fused5_kernel<3, 16>;
        case 4: return fused5_kernel<4, 16>;
        default: return fused5_kernel<5, 16>;
    }
}

int launch_filter_v5(ugvc_ctx* ctx, const FilterArgs& a) {
    if (a.n == 0) return 0;
    V5Args v;
    if (v5_fill_args(ctx, v, a)) return -1;
    static bool attr_set = false;
    if (!attr_set) {
        for (K5 f : {fused5_for(0), fused5_for(1), fused5_for(2), fused5_for(3), fused5_for(4), fused5_for(5), fused5_for(3, true), (K5)forest5_kernel})
            UGVC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
        attr_set = true;
    }
    // UGVC_DEBUG_SYNC=1: name every launch on stderr and wait for it (a GPU memory fault aborts the process; the
    // last name printed is the kernel that faulted)
    static const bool dbg = getenv("UGVC_DEBUG_SYNC") != nullptr;
    auto step = [&](const char* name) -> int {
        if (!dbg) return 0;
        fprintf(stderr, "[v5] %s done? ", name);
        fflush(stderr);
        UGVC_HIP(hipStreamSynchronize(ctx->stream));
        fprintf(stderr, "ok\n");
        return 0;
    };
    UGVC_HIP(hipMemsetAsync(v.tile_cnt, 0, 2 * kTileShards5 * kTileCntStride5 * 4, ctx->stream));
    hipLaunchKernelGGL(compact5_kernel, dim3((unsigned)v.n_cblocks), dim3(256), 0, ctx->stream, v);
    if (step("compact5")) return -1;
    // real tiles of both classes together: at most one per 64 rows plus one per class and compaction block
    const int n_act = (a.has_runs ? 1 : 0) + a.n_tracks + (a.n_bl > 0 ? 1 : 0);
    const int64_t nbr = ((a.n + 63) / 64 + 2 * (int64_t)v.n_cblocks) * 2 * std::max(n_act, 1);
    hipLaunchKernelGGL(bracket5_kernel, dim3((unsigned)((nbr + 255) / 256)), dim3(256), 0, ctx->stream, v);
    if (step("bracket5")) return -1;
    const size_t lds_f = lds5_bytes(v, v.n_waves);
    if (dbg) fprintf(stderr, "[ugvc v5] fused5: %d waves (%d indel), %zu B of LDS\n", v.n_waves, v.n_indel_waves, lds_f);
    hipLaunchKernelGGL(fused5_for(a.n_tracks, (a.ablate & (1 << 28)) != 0), dim3((unsigned)ctx->n_cus), dim3(v.n_waves * 64), lds_f, ctx->stream, v);
    if (step("fused5")) return -1;
    if (!(a.ablate & 262144) && (v.pg[1].ok || v.pg[2].ok)) {
        int n_waves = 0;
        size_t lds = 0;
        for (int w : {16, 12, 8, 4}) {
            size_t need = 0;
            for (int g = 1; g < UGVC_N_GROUPS; ++g)
                if (v.pg[g].ok) need = std::max(need, k5_forest_lds(v.pg[g], w));
            if (need + 1088 <= 158 * 1024) { n_waves = w; lds = need; break; }
        }
        if (n_waves == 0) return fail("internal: packed forest does not fit LDS");
        hipLaunchKernelGGL(forest5_kernel, dim3((unsigned)ctx->n_cus), dim3(n_waves * 64), lds, ctx->stream, v);
        if (step("forest5")) return -1;
    }
    UGVC_HIP(hipGetLastError());
    return 0;
}

}  // namespace ugvc
