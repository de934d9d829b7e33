// Device helpers shared by the scoring-pass kernels (kernels_v3.hip, kernels_v5.hip): raw LDS accessors,
// lower bounds over global tables, the flow-space cycle-skip walk, the LDS forest walks.
#pragma once
#include "ugvc_v2.hpp"

namespace ugvc {

constexpr int kWinDw = 12;            // 48-byte reference window per variant
constexpr int kWinStride = 13;        // dwords per lane row (odd: conflict-free column access)
constexpr int kWinBytes = kWinDw * 4;

__device__ __forceinline__ int rfl(int x) { return __builtin_amdgcn_readfirstlane(x); }

// Raw LDS byte addresses (what ds_read takes).  A __shared__ object's generic pointer cast to the
// local address space IS its LDS offset; the descents below carry such offsets in VGPRs so a
// step is one v_add + one ds_read, with no re-derivation from an index.
#define UGVC_LDS __attribute__((address_space(3)))
template <class T> __device__ __forceinline__ uint32_t lds_addr(T* p) {
    return (uint32_t)(uintptr_t)(UGVC_LDS T*)p;
}
__device__ __forceinline__ int lds_i32(uint32_t a) { return *(UGVC_LDS const int32_t*)(uintptr_t)a; }
__device__ __forceinline__ uint32_t lds_u32(uint32_t a) { return *(UGVC_LDS const uint32_t*)(uintptr_t)a; }
__device__ __forceinline__ uint64_t lds_u64(uint32_t a) { return *(UGVC_LDS const uint64_t*)(uintptr_t)a; }
__device__ __forceinline__ double lds_f64(uint32_t a) { return *(UGVC_LDS const double*)(uintptr_t)a; }
__device__ __forceinline__ float lds_f32(uint32_t a) { return *(UGVC_LDS const float*)(uintptr_t)a; }
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint2 lds_u32x2(uint32_t a) {
    const u32x2_t x = *(UGVC_LDS const u32x2_t*)(uintptr_t)a;
    return make_uint2(x.x, x.y);
}
__device__ __forceinline__ uint32_t lds_u16(uint32_t a) { return *(UGVC_LDS const uint16_t*)(uintptr_t)a; }

__device__ __forceinline__ int lb_i32_g(const int32_t* __restrict__ a, int lo, int hi, int key) {
    int base = lo, len = hi - lo;
    while (len > 0) {
        const int half = len >> 1;
        const bool lt = a[base + half] < key;
        base = lt ? base + half + 1 : base;
        len = lt ? len - half - 1 : half;
    }
    return base;
}

__device__ __forceinline__ int lb_u64_g(const uint64_t* __restrict__ a, int lo, int hi, uint64_t key) {
    int base = lo, len = hi - lo;
    while (len > 0) {
        const int half = len >> 1;
        const bool lt = a[base + half] < key;
        base = lt ? base + half + 1 : base;
        len = lt ? len - half - 1 : half;
    }
    return base;
}

__device__ __forceinline__ const TrackView& table_view(const FilterArgs& f, int t) {   // t: 0 runs, 1.. tracks
    return t == 0 ? f.runs : f.tracks[t - 1];
}
__device__ __forceinline__ bool table_present(const FilterArgs& f, int t) {
    return t == 0 ? f.has_runs != 0 : (t - 1) < f.n_tracks;
}

// Two-level lower bound: the 1/64 sample `coarse` (coarse[k] = a[64 k], L2-resident) narrows [lo, hi) to one
// 64-element block, the block itself is finished by binary search: ~16 L2-hit steps + 6 steps on two cache lines
// instead of 22 steps of HBM latency.  (8-ary steps - seven probes issued together - measured in the per-tile bracket kernel the v5 pass had in round 2:
// 48 us instead of 33; the searches are bound by the number of scattered gathers, not by the length of the chain.)
template <class T>
__device__ __forceinline__ int lb_two_level_g(const T* __restrict__ a, const T* __restrict__ coarse, int lo, int hi, T key) {
    if (coarse && hi - lo > 128) {
        // blocks whose first element lies inside (lo, hi): j in [jl, jh); count those below the key
        const int jl = (lo >> 6) + 1, jh = (hi + 63) >> 6;
        int base = jl, len = jh - jl;
        while (len > 0) {
            const int half = len >> 1;
            const bool lt = coarse[base + half] < key;
            base = lt ? base + half + 1 : base;
            len = lt ? len - half - 1 : half;
        }
        // base = first block in [jl, jh] whose first element is >= key: the answer lies in block base - 1 (or at its end)
        const int blo = (base - 1) << 6, bhi = base << 6;
        lo = blo > lo ? blo : lo;
        hi = bhi < hi ? bhi : hi;
    }
    int b = lo, len = hi - lo;
    while (len > 0) {
        const int half = len >> 1;
        const bool lt = a[b + half] < key;
        b = lt ? b + half + 1 : b;
        len = lt ? len - half - 1 : half;
    }
    return b;
}

template <class SeqR, class SeqA>
__device__ __forceinline__ int cycle_skip_walk(int L, const uint8_t flow[4], SeqR seq_r, SeqA seq_a) {
    int pr = 0, pa = 0, lr = 0, la = 0;
    bool poss = false;
    for (int s = 0; pr < L || pa < L; ++s) {
        const int b = flow[s & 3];
        const bool ar = pr < L, aa = pa < L;
        int hr = 0, ha = 0;
        if (ar) { while (pr + hr < L && seq_r(pr + hr) == b) ++hr; pr += hr; ++lr; }
        if (aa) { while (pa + ha < L && seq_a(pa + ha) == b) ++ha; pa += ha; ++la; }
        if (ar && aa && hr != ha && (hr == 0 || ha == 0)) poss = true;
    }
    if (lr != la) return 2;
    return poss ? 1 : 0;
}

// idx' = 2 * idx + (lane's bit of mask): one v_addc_co_u32 (the compare result feeds the carry-in)
__device__ __forceinline__ uint32_t twice_plus_carry(uint32_t idx, unsigned long long mask) {
    uint32_t out;
    unsigned long long cout;
    asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(out), "=s"(cout) : "v"(idx), "s"(mask));
    return out;
}

// A node visit is 4 VALU + 2 LDS reads: v_lshl_add (node address), ds_read_b32, v_add_sdwa (code
// address from the node's high half), ds_read_u16, v_cmp_sdwa (code vs rank) and v_addc.  The
// scoring pass is VALU-issue bound on gfx950 (one wave-instruction per 4 cycles per SIMD; measured
// 73 % VALU busy at 7.3 VALU per visit), so the instruction count per visit is what matters; NT
// independent trees per lane cover the two dependent LDS latencies of a level.
template <int NT>
__device__ __forceinline__ void walk3(uint32_t nodes_b, uint32_t planes_lane_b, int t, int D, int NL, int (&leaf)[NT]) {
    uint32_t idx[NT], tb[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        tb[k] = nodes_b + 4u * (uint32_t)((t + k) * NL);
        idx[k] = 1;
    }
    for (int d = 0; d < D; ++d) {
        uint32_t w[NT], code[NT];
#pragma unroll
        for (int k = 0; k < NT; ++k) w[k] = lds_u32(tb[k] + 4u * idx[k]);
#pragma unroll
        for (int k = 0; k < NT; ++k) code[k] = lds_u16(planes_lane_b + (w[k] >> 16));
#pragma unroll
        for (int k = 0; k < NT; ++k)
            idx[k] = twice_plus_carry(idx[k], __builtin_amdgcn_ballot_w64(code[k] > (w[k] & 0xFFFFu)));
    }
#pragma unroll
    for (int k = 0; k < NT; ++k) leaf[k] = (t + k) * NL + (int)idx[k] - NL;
}

// Single-sum RF walk (PackedGroupView::fast4).  Levels 0..D-2 as in walk3 over a heap of H = 2^(D-1)
// dwords per tree; the last level is ONE 8-byte read (64 banks) of {node word, payload indices of
// both children}, issued with its code read, and the child is picked by a compare + SDWA select -
// no leaf-index gather, no 16-byte payload gather: the class-1 probability is one 8-byte read.
__device__ __forceinline__ uint32_t pick_half(uint32_t code, uint32_t node, uint32_t both) {
    uint32_t out;
    asm("v_cmp_gt_u16_sdwa vcc, %1, %2 src0_sel:WORD_0 src1_sel:WORD_0\n\t"
        "v_cndmask_b32_sdwa %0, %3, %3, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1"
        : "=v"(out) : "v"(code), "v"(node), "v"(both) : "vcc");
    return out;
}

template <int NT>
__device__ __forceinline__ void walk4(uint32_t hi_b, uint32_t last_b, uint32_t planes_lane_b, int t, int D, int H,
                                      uint32_t (&pidx)[NT]) {
    uint32_t idx[NT], tb[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        tb[k] = hi_b + 4u * (uint32_t)((t + k) * H);
        idx[k] = 1;
    }
    if (D > 1) {
        // the root is the same word for every lane: no index to carry in, the child is 2 + (code > thr)
        uint32_t w[NT], code[NT];
#pragma unroll
        for (int k = 0; k < NT; ++k) w[k] = lds_u32(tb[k] + 4u);
#pragma unroll
        for (int k = 0; k < NT; ++k) code[k] = lds_u16(planes_lane_b + (w[k] >> 16));
#pragma unroll
        for (int k = 0; k < NT; ++k) idx[k] = code[k] > (w[k] & 0xFFFFu) ? 3u : 2u;
    }
    for (int d = 1; d < D - 1; ++d) {
        uint32_t w[NT], code[NT];
#pragma unroll
        for (int k = 0; k < NT; ++k) w[k] = lds_u32(tb[k] + 4u * idx[k]);
#pragma unroll
        for (int k = 0; k < NT; ++k) code[k] = lds_u16(planes_lane_b + (w[k] >> 16));
#pragma unroll
        for (int k = 0; k < NT; ++k)
            idx[k] = twice_plus_carry(idx[k], __builtin_amdgcn_ballot_w64(code[k] > (w[k] & 0xFFFFu)));
    }
    uint2 wl[NT];
    uint32_t code[NT];
    // entry of heap index i sits at i - H; the per-tree base stays one SGPR (readfirstlane keeps the
    // compiler from folding it into the per-lane index: one v_lshl_add per read)
#pragma unroll
    for (int k = 0; k < NT; ++k)
        wl[k] = lds_u32x2((uint32_t)rfl((int)(last_b + 8u * (uint32_t)((t + k) * H - H))) + 8u * idx[k]);
#pragma unroll
    for (int k = 0; k < NT; ++k) code[k] = lds_u16(planes_lane_b + (wl[k].x >> 16));
#pragma unroll
    for (int k = 0; k < NT; ++k) pidx[k] = pick_half(code[k], wl[k].x, wl[k].y);
}

// ---- round 4: the walk over ONE heap for all trees of a group (model_pack.hip: pack_group5) -------------------------
// Node (tree t, level d, position j) = dword I = (T + t) 2^d + j of the `hi` table, child = 2 I + c for every tree: the
// state of a walk is I alone and one scalar base serves all trees in flight.  The ROOT words of the batch come from a scalar
// load (wave-uniform: no LDS read, no per-lane address), I after the root = 2 (T + t + k) + c is one v_addc with the tree's
// offset as an inline constant; the last level's payload IS the byte offset of the leaf's class-1 probability in p1.
typedef uint32_t u32x16_t __attribute__((ext_vector_type(16)));
typedef uint32_t u32x8_t __attribute__((ext_vector_type(8)));
#define UGVC_CONST_AS __attribute__((address_space(4)))

template <int K2>
__device__ __forceinline__ uint32_t base_plus_const_plus_carry(uint32_t base, unsigned long long mask) {
    uint32_t out;
    unsigned long long cout;
    asm("v_addc_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(out), "=s"(cout) : "v"(base), "n"(K2), "s"(mask));
    return out;
}

template <int NT>
__device__ __forceinline__ void walk6(uint32_t hi_b, uint32_t last_base, uint32_t planes_lane_b, const uint32_t* roots, int t, int T, int D,
                                      uint32_t (&pay)[NT]) {
    static_assert(NT <= 16, "sixteen trees in flight at most (inline constants of the root step)");
    uint32_t I[NT];
    {                                                            // (D >= 2: a depth-1 forest is walked by stump_payload)
        uint32_t rw[NT];
        if constexpr (NT == 16) {
            const u32x16_t r = *(const UGVC_CONST_AS u32x16_t*)(uintptr_t)(roots + t);      // (t is a multiple of 16: 64-byte aligned)
#pragma unroll
            for (int k = 0; k < NT; ++k) rw[k] = r[k];
        } else if constexpr (NT == 8) {
            const u32x8_t r = *(const UGVC_CONST_AS u32x8_t*)(uintptr_t)(roots + t);        // (t is a multiple of 8)
#pragma unroll
            for (int k = 0; k < NT; ++k) rw[k] = r[k];
        } else {
#pragma unroll
            for (int k = 0; k < NT; ++k) rw[k] = *(const UGVC_CONST_AS uint32_t*)(uintptr_t)(roots + t + k);
        }
        uint32_t code[NT];
#pragma unroll
        for (int k = 0; k < NT; ++k) code[k] = lds_u16(planes_lane_b + (rw[k] >> 16));
        const uint32_t vb2 = 2u * (uint32_t)(T + t);
        unsigned long long m[NT];
#pragma unroll
        for (int k = 0; k < NT; ++k) m[k] = __builtin_amdgcn_ballot_w64(code[k] > (rw[k] & 0xFFFFu));
#define UGVC_ROOT_STEP(K) if constexpr (NT > K) I[K] = base_plus_const_plus_carry<2 * K>(vb2, m[K]);
        UGVC_ROOT_STEP(0) UGVC_ROOT_STEP(1) UGVC_ROOT_STEP(2) UGVC_ROOT_STEP(3) UGVC_ROOT_STEP(4) UGVC_ROOT_STEP(5) UGVC_ROOT_STEP(6) UGVC_ROOT_STEP(7)
        UGVC_ROOT_STEP(8) UGVC_ROOT_STEP(9) UGVC_ROOT_STEP(10) UGVC_ROOT_STEP(11) UGVC_ROOT_STEP(12) UGVC_ROOT_STEP(13) UGVC_ROOT_STEP(14) UGVC_ROOT_STEP(15)
#undef UGVC_ROOT_STEP
    }
    for (int d = 1; d < D - 1; ++d) {
        uint32_t w[NT], code[NT];
#pragma unroll
        for (int k = 0; k < NT; ++k) w[k] = lds_u32(hi_b + 4u * I[k]);
#pragma unroll
        for (int k = 0; k < NT; ++k) code[k] = lds_u16(planes_lane_b + (w[k] >> 16));
#pragma unroll
        for (int k = 0; k < NT; ++k)
            I[k] = twice_plus_carry(I[k], __builtin_amdgcn_ballot_w64(code[k] > (w[k] & 0xFFFFu)));
    }
    uint2 wl[NT];
    uint32_t code[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) wl[k] = lds_u32x2(last_base + 8u * I[k]);
#pragma unroll
    for (int k = 0; k < NT; ++k) code[k] = lds_u16(planes_lane_b + (wl[k].x >> 16));
#pragma unroll
    for (int k = 0; k < NT; ++k) pay[k] = pick_half(code[k], wl[k].x, wl[k].y);
}

// a forest of depth 1: the root is the last level (entry t of the `last` table)
__device__ __forceinline__ uint32_t stump_payload(uint32_t last_b, uint32_t planes_lane_b, int t) {
    const uint2 wl = lds_u32x2(last_b + 8u * (uint32_t)t);
    const uint32_t code = lds_u16(planes_lane_b + (wl.x >> 16));
    return pick_half(code, wl.x, wl.y);
}

}  // namespace ugvc
