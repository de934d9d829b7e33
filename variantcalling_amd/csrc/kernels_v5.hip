// v5 scoring pass (gfx950): featurize -> lookup -> score -> FILTER as two launches.
//
// What v3 measured (profiles/r01_*): its featurize kernel (K1, 378 us per 5 M variants) waits on a chain of
// dependent memory round trips per 256-variant tile at 4 waves/SIMD, with the vector ALU mostly idle; its forest
// kernel (K2, 321 us) is bound by the LDS pipeline and runs AFTER K1; the two exchange a 16-byte record per
// variant through HBM (160 MB per pass) and K2 stores its results scattered.  v5 puts both on the same waves:
//
//   fused5_kernel     one 16-wave workgroup per CU owns a contiguous range of the callset's rows and holds the
//                     SNP forest in LDS (rank-coded complete trees, single-sum layout) with the float thresholds.
//                     Prologue: the rows are split by variant class (ref_len == alt_len: the SNP forest; else an
//                     indel) into two dense, ordered lists of row indices; a tile is 64 consecutive entries - one
//                     wave's work, pure in class, so every lane of a wave walks the SAME forest.  After that a wave
//                     is autonomous (no workgroup barrier): it takes CONSECUTIVE tiles, so where its next tile
//                     starts in every side table is where its last variant ended (carried ranks; a fresh search -
//                     64 probes per step by the whole wave - only at its first tile and at contig changes).  Per
//                     tile it loads the columns, an 11-base reference window per lane (one 16-byte load, realigned
//                     with v_alignbyte), stages the side-table slices the tile can touch in wave-private LDS
//                     (sentinel padded: the lock-step descents carry no bounds test and no branch), derives the
//                     features, writes 16-bit codes straight into its code planes and walks the forest; score /
//                     FILTER / flags leave in variant order.  Row indices, columns and slices of the NEXT tile are
//                     in flight during the current one.  A quarter of the waves (by the class mix) work on the
//                     indel tiles instead (48-byte window, homopolymer logic, slices of two or six rows per lane)
//                     and leave 48-byte raw-code records in their variant-type group's list.
//   forest5_kernel    walks the indel groups' forests over those records (the v3 forest kernel on raw records).
//
// (Until round 2 the class split and the per-tile lower bounds were two kernels of their own in front - atomically
// handed-out tile slots, one search thread per tile and table: 17 + 37 us per 5 M variants and ~50 us of latency at
// any size.  Ordered lists make both unnecessary.)
//
// Codes: floats (qual, sor, vaf, gc) are ranked against the group's sorted thresholds (exact, as v3); every
// other feature is a non-negative integer and is used as it stands, clamped to one past the largest threshold
// the forest tests (ugvc_v2.hpp) - no code tables, no gathers.
// The number of annotation tracks is a template parameter: the per-table code is straight-line.
// Semantics are those of the oracle (oracle/oracle.py); parity tests run v5, v3 and v1 against it.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <string>
#include <vector>

#include "ugvc_walk.hpp"
#include <atomic>

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

// Element `idx` of a resident array through `SGPR base + 32-bit byte offset` addressing (one 32-bit shift instead of
// 64-bit address arithmetic per lane): every array of this pass is shorter than 4 GiB (v5_available checks the row counts).
// A reference window is read ONCE per pass, at a place no other lane shares (variants lie ~600 bases apart): a NON-TEMPORAL load,
// so that the window lines do not push the side-table slices - which every tile re-reads - out of the caches (round 5, two builds
// alternating on one box: pass 0.4144 against 0.4165 ms, HBM traffic 1.829 against 1.867 GB, profiles/r05_nt_window_ab.txt).
typedef uint32_t u32x4_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 load_window16(const uint8_t* p) {
    const u32x4_nt x = __builtin_nontemporal_load(reinterpret_cast<const u32x4_nt*>(p));
    return make_uint4(x.x, x.y, x.z, x.w);
}
template <class T> __device__ __forceinline__ T ldg32(const T* base, uint32_t idx) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + (uint32_t)(idx * (uint32_t)sizeof(T)));
}
template <class T> __device__ __forceinline__ void stg32(T* base, uint32_t idx, T x) {
    *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + (uint32_t)(idx * (uint32_t)sizeof(T))) = x;
}

constexpr int kGcRank = 121;         // gc_content takes 121 values (count / len, len and count in 0..10): its rank code is a table
constexpr int kGcRankBytes = 768;    // 3 groups x 121 u16, padded
constexpr int kWinRowB = kWinStride * 4;

// ---- where a wave stands in the side tables -------------------------------------------------------------
// A wave works through CONSECUTIVE tiles of its class in callset order, so the lower bounds its next tile starts
// from are the ranks its last variant ended at: they are carried, not searched.  A fresh search happens at a wave's
// first tile and when the contig changes, and then the whole wave does it together.
template <int NT>
struct Brk {
    int c;                          // contig the state belongs to (-1: none - the next tile searches afresh)
    int64_t clo, chi;               // its span of the reference
    int plo[NT], phi[NT];           // its row range in every interval table
    int L[NT];                      // # starts < pos of the last variant seen (global row index)
    int Lb;                         // # blacklist keys < its key
};

// Lower bounds of (pos | key) in every table at once, by the whole wave: 64 evenly spaced probes per table and
// step, one gather each, the ballot's population count picks the sub-range - a 3 M-row table closes in four round
// trips (binary: 22).
template <int NT>
__device__ __forceinline__ void coop_search(const FilterArgs& a, Brk<NT>& bk, int pos, uint64_t key, int lane) {
    int b[NT], len[NT];
    int bb = 0, blen = a.n_bl > 0 ? (int)a.n_bl : 0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        b[t] = bk.plo[t];
        len[t] = (t > 0 || a.has_runs) ? bk.phi[t] - bk.plo[t] : 0;
    }
    for (;;) {
        bool any = blen > 0;
#pragma unroll
        for (int t = 0; t < NT; ++t) any |= len[t] > 0;
        if (!any) break;
        int x[NT], step[NT];
        uint64_t xk = ~0ull;
        int bstep = 0;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            step[t] = (len[t] + 63) >> 6;
            x[t] = INT32_MAX;
            if (len[t] > 0 && lane * step[t] < len[t]) x[t] = table_view(a, t).starts[b[t] + lane * step[t]];
        }
        if (blen > 0) {
            bstep = (blen + 63) >> 6;
            if (lane * bstep < blen) xk = a.bl[bb + lane * bstep];
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (len[t] > 0) {
                const int k = (int)__popcll(__ballot(x[t] < pos));          // probes ascend: the first k are below
                const int end = b[t] + len[t];
                const int nb = k > 0 ? b[t] + (k - 1) * step[t] + 1 : b[t];
                const int ne = min(b[t] + k * step[t], end);
                b[t] = nb;
                len[t] = ne > nb ? ne - nb : 0;
            }
        }
        if (blen > 0) {
            const int k = (int)__popcll(__ballot(xk < key));
            const int end = bb + blen;
            const int nb = k > 0 ? bb + (k - 1) * bstep + 1 : bb;
            const int ne = min(bb + k * bstep, end);
            bb = nb;
            blen = ne > nb ? ne - nb : 0;
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) bk.L[t] = rfl(b[t]);
    bk.Lb = rfl(bb);
}

template <int NT>
__device__ __forceinline__ void brk_refresh(const FilterArgs& a, Brk<NT>& bk, int c0, int pos0, int lane) {
    bk.c = c0;
    bk.clo = cload(a.contig_off + c0);
    bk.chi = cload(a.contig_off + c0 + 1);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        bk.plo[t] = bk.phi[t] = 0;
        if (t == 0 && !a.has_runs) continue;
        const TrackView& tv = table_view(a, t);
        bk.plo[t] = cload(tv.ptr + c0);
        bk.phi[t] = cload(tv.ptr + c0 + 1);
    }
    coop_search<NT>(a, bk, pos0, ((uint64_t)(uint32_t)c0 << 32) | (uint32_t)pos0, lane);
}

// ---- the state for the SECOND contig of a workgroup, searched once and left in LDS (round 4) ----------------------
// A workgroup whose rows cross a contig boundary holds two waves (one per class) that have to search afresh in the middle of
// their work - four dependent round trips, ~19 k ticks: those workgroups were the last of every launch.  The workgroup's last
// SNP wave (the short share: it has room, or no tiles at all in a shard-sized launch) searches for the first row of the second
// contig before its own tiles and publishes the state at `rec_b` (gtab + 512: 24 dwords); a wave that meets that contig adopts
// it if it is there by then, and searches itself if not.  The published ranks belong to the second contig's FIRST row of
// either class: lower bounds for every later row of it (rows ascend by position: validated at upload).
template <int NT>
__device__ __forceinline__ void brk_publish(uint32_t rec_b, const Brk<NT>& bk, int lane) {
    if (lane == 0) {
        lds_st32(rec_b + 4u, bk.Lb);
        lds_st64(rec_b + 8u, (uint64_t)bk.clo);
        lds_st64(rec_b + 16u, (uint64_t)bk.chi);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            lds_st32(rec_b + 24u + 4u * t, bk.plo[t]);
            lds_st32(rec_b + 24u + 4u * (kJoin5 + t), bk.phi[t]);
            lds_st32(rec_b + 24u + 4u * (2 * kJoin5 + t), bk.L[t]);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) lds_st32(rec_b, bk.c + 1);
}
template <int NT>
__device__ __forceinline__ bool brk_adopt(uint32_t rec_b, Brk<NT>& bk, int c0) {
    if (rfl(lds_i32(rec_b)) != c0 + 1) return false;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    bk.c = c0;
    bk.Lb = rfl(lds_i32(rec_b + 4u));
    const uint64_t lo = lds_u64(rec_b + 8u), hi = lds_u64(rec_b + 16u);
    bk.clo = (int64_t)(((uint64_t)(uint32_t)rfl((int)(lo >> 32)) << 32) | (uint32_t)rfl((int)lo));
    bk.chi = (int64_t)(((uint64_t)(uint32_t)rfl((int)(hi >> 32)) << 32) | (uint32_t)rfl((int)hi));
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        bk.plo[t] = rfl(lds_i32(rec_b + 24u + 4u * t));
        bk.phi[t] = rfl(lds_i32(rec_b + 24u + 4u * (kJoin5 + t)));
        bk.L[t] = rfl(lds_i32(rec_b + 24u + 4u * (2 * kJoin5 + t)));
    }
    return true;
}

#ifdef UGVC_PHASE_CLOCK
struct PhaseClk { uint64_t last; uint64_t acc[16]; };
#define CLK(pc, k) do { const uint64_t now_ = __builtin_readcyclecounter(); (pc).acc[k] += now_ - (pc).last; (pc).last = now_; } while (0)
#else
struct PhaseClk {};
#define CLK(pc, k) do { } while (0)
#endif

// ---- joins ---------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_max_i32(int x);
struct __attribute__((packed, aligned(1))) U2u { uint32_t x, y; };     // an 8-byte load from any byte address
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
__device__ __forceinline__ int join_one_global(const FilterArgs* ap, int t, int lo, int hi, int plo, int phi, int pos, uint64_t key,
                                            JoinOut* op) {
    const FilterArgs& a = *ap;
    JoinOut o = *op;
    int rank = lo;                                           // the lane's lower bound (what the next tile starts from)
    if (t == kJoin5 - 1) {
        const int r = lb_u64_g(a.bl, lo, hi, key);
        o.cohort = r < (int)a.n_bl && a.bl[r] == key;
        rank = r;
    } else if (phi > plo) {
        const TrackView& tv = table_view(a, t);
        const int sg = lb_i32_g(tv.starts, lo, hi, pos);
        const int top = phi - 1;
        auto S = [&](int i) { return tv.starts[i < plo ? plo : (i > top ? top : i)]; };     // clamped into the contig's rows;
        auto E = [&](int i) { return tv.ends[i < plo ? plo : (i > top ? top : i)]; };       // interval_verdict's guards discard them
        if (t == 0) { o.inside_run = o.close_run = false; }
        else o.trk &= ~(1u << (t - 1));
        interval_verdict(t, sg, plo, phi, pos, a.hpol_dist, S, E, o);
        rank = sg;
    }
    *op = o;
    return rank;
}

// Indel tiles cover ~5x the span of an SNP tile, so their slices are staged per table, from two rows before the
// carried rank: two rows per lane (kIndelRows), fetched into registers one tile ahead and written to the wave's
// scratch two tables at a time once the window rows are dead; six rows per lane for a table the host marks dense
// (V5Args::iwide), fetched at the joins.  A slice whose last row does not reach the tile's last variant is searched
// in HBM instead (join_one_global).
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

__device__ __forceinline__ int staged_verdict_at(const FilterArgs& a, uint32_t slot_b, int t, int L0, int plo, int phi, int pos, uint32_t rank,
                                                 JoinOut& o) {
    if (phi <= plo) return plo;
    const int sg = L0 + (int)rank;
    auto S = [&](int gi) { return lds_i32(slot_b + 4u * (uint32_t)(gi - L0)); };
    auto E = [&](int gi) { return lds_i32(slot_b + 512u + 4u * (uint32_t)(gi - L0)); };
    interval_verdict(t, sg, plo, phi, pos, a.hpol_dist, S, E, o);
    return sg;
}

__device__ __forceinline__ int staged_verdict(const FilterArgs& a, uint32_t slot_b, int t, int L0, int plo, int phi, int pos, JoinOut& o) {
    if (phi <= plo) return plo;
    return staged_verdict_at(a, slot_b, t, L0, plo, phi, pos, staged_rank(slot_b, pos), o);
}

// A table too dense for the two-rows-per-lane slice (a 3 M-interval track under a 220 kb indel tile: ~210 rows):
// six rows per lane, the whole scratch, a round of its own; the descent clamps its probes to the last staged row
// (rows past the searched range compare like the padding would).
constexpr int kWideChunks = 6, kWideRows = 64 * kWideChunks;          // 384 rows: starts | ends = 3072 B

__device__ __forceinline__ void wide_load(const TrackView& tv, int L0, int top, int lane, int (&wv)[kWideChunks], int (&we)[kWideChunks]) {
#pragma unroll
    for (int h = 0; h < kWideChunks; ++h) {
        const uint32_t gs = (uint32_t)max(min(L0 + 64 * h + lane, top), 0);
        wv[h] = ldg32(tv.starts, gs); we[h] = ldg32(tv.ends, gs);
    }
}

__device__ __forceinline__ void wide_store(uint32_t A, int L0, int plo, int phi, int lane, const int (&wv)[kWideChunks], const int (&we)[kWideChunks]) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int h = 0; h < kWideChunks; ++h) {
        const int gi = L0 + 64 * h + lane;
        lds_st32(A + 256u * h + 4u * lane, gi < plo ? INT32_MIN : (gi >= phi ? INT32_MAX : wv[h]));
        lds_st32(A + 4u * kWideRows + 256u * h + 4u * lane, we[h]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int wide_verdict(const FilterArgs& a, uint32_t A, int t, int L0, int plo, int phi, int pos, JoinOut& o) {
    if (phi <= plo) return plo;
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
    return sg;
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
    // (round 4: the level step is the walk's - v_lshl_add for the address off a SCALAR base, the compare into an SGPR pair,
    // v_addc for i = 2 i + carry; written with a select and a shift-or the compiler spent 5.7 instructions per level and feature)
    uint32_t i[3] = {1u, 1u, 1u};
    uint32_t b[3];
#pragma unroll
    for (int e = 0; e < 3; ++e) b[e] = (uint32_t)rfl((int)base[e]);
    const int bmin = min(bits[0], min(bits[1], bits[2])), bmax = max(bits[0], max(bits[1], bits[2]));
    for (int s = 0; s < bmin; ++s) {
        float t[3];
#pragma unroll
        for (int e = 0; e < 3; ++e) t[e] = lds_f32(b[e] + 4u * i[e]);
#pragma unroll
        for (int e = 0; e < 3; ++e) i[e] = twice_plus_carry(i[e], __builtin_amdgcn_ballot_w64(t[e] < fx[e]));
    }
    for (int s = bmin; s < bmax; ++s) {
#pragma unroll
        for (int e = 0; e < 3; ++e)
            if (s < bits[e]) {
                const float t = lds_f32(b[e] + 4u * i[e]);
                i[e] = twice_plus_carry(i[e], __builtin_amdgcn_ballot_w64(t < fx[e]));
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
    k.c = ldg32(a.contig, i); k.pos = ldg32(a.pos, i); k.rl = ldg32(a.ref_len, i);
    k.ro = ldg32(a.ref_off, i); k.ao = ldg32(a.alt_off, i);
    k.qual = ldg32(a.qual, i); k.sor = ldg32(a.sor, i);
    k.dp = ldg32(a.dp, i); k.adr = ldg32(a.ad_ref, i); k.ada = ldg32(a.ad_alt, i); k.gq = ldg32(a.gq, i);
    return k;
}

// The side-table slices of one SNP tile, in registers: fetched from the carried ranks alone, one tile ahead (they are
// in flight during the previous tile's walk and are written to the wave's LDS scratch when that walk has finished
// with its code planes).  They start at the ranks the previous tile's last variant ended at (Brk).
template <int NT>
struct SlicePre {
    int sv[NT][2], ev[NT][2];
    uint64_t bl;
};

template <int NT>
__device__ __forceinline__ void issue_slices(const V5Args& v, const Brk<NT>& bk, int lane, SlicePre<NT>& s) {
    const FilterArgs& a = v.f;
    s.bl = ~0ull;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        s.sv[t][0] = s.sv[t][1] = s.ev[t][0] = s.ev[t][1] = 0;
        if (t == 0 && !a.has_runs) continue;
        const TrackView& tv = table_view(a, t);
        const int Lt = bk.L[t] - 2;
        const int top = max(v.na[t] - 1, 0);
        {
            const uint32_t gs = (uint32_t)max(min(Lt + lane, top), 0);
            s.sv[t][0] = ldg32(tv.starts, gs); s.ev[t][0] = ldg32(tv.ends, gs);
        }
        if (v.jcap[t] > 64) {
            const uint32_t gs = (uint32_t)max(min(Lt + 64 + lane, top), 0);
            s.sv[t][1] = ldg32(tv.starts, gs); s.ev[t][1] = ldg32(tv.ends, gs);
        }
    }
    if (a.n_bl > 0) {
        const int64_t gi = (int64_t)bk.Lb + lane;
        if (gi < a.n_bl) s.bl = ldg32(a.bl, (uint32_t)gi);
    }
}


// ---- config C5: the N x F f32 feature matrix from the same featurize waves (round 3) -------------------
// `train_models_pipeline` fits on this matrix (docs/train_models_pipeline.md:5-10).  Round 2 built it with the universal
// one-thread-per-variant kernel (per-lane binary searches: 0.10 of the HBM roofline); with WX the fused kernel's
// tiles write the lane's RAW feature values (schema.BASE_FEATURES order) instead of ranking them and walking - no model
// is needed, nothing but X and `group` is written.  A row is 4 F bytes: 16-byte stores when F is a multiple of 4.
__device__ __forceinline__ void store_feature_row(const FilterArgs& a, uint32_t i, bool live, const float (&x)[kMaxFeatures], int group) {
    if (!live || (a.ablate & 4194304)) return;                   // (profiling bit: the matrix is not written)
    const int F = UGVC_N_BASE_FEATURES + a.n_tracks;
    float* row = a.X + (size_t)i * (size_t)F;
    if ((F & 3) == 0) {
#pragma unroll
        for (int q = 0; q < kMaxFeatures / 4 + 1; ++q)
            if (4 * q < F) reinterpret_cast<float4*>(row)[q] = make_float4(x[4 * q], x[(4 * q + 1) % kMaxFeatures], x[(4 * q + 2) % kMaxFeatures], x[(4 * q + 3) % kMaxFeatures]);
    } else {
#pragma unroll
        for (int f = 0; f < kMaxFeatures; ++f)
            if (f < F) row[f] = x[f];
    }
    if (a.group) a.group[i] = (uint8_t)group;
}

// The same rows, COALESCED: a lane-per-row store puts 64 separate 16-byte pieces, 80 bytes apart, into every store
// instruction (~40 cache lines touched per instruction, five instructions per tile - measured: 57 of the 187 us of a 2 M-row
// build, `--variant 4194304` against 0).  Here the tile's rows go through the wave's LDS scratch row-major - which IS the
// order of the matrix in memory - and lane j of store k writes 16-byte piece 64 k + j of that image (row = piece / (F/4)):
// consecutive lanes write consecutive addresses wherever the tile's rows are neighbours in the callset (they mostly are:
// a tile is 64 consecutive rows of one class).  LDS: 64 F floats + 64 row indices behind them at `base`.
template <int F>
__device__ __forceinline__ void store_feature_rows_tile(const FilterArgs& a, uint32_t base, int lane, uint32_t i, bool live,
                                                        const float (&x)[kMaxFeatures], int group) {
    static_assert(F % 4 == 0 && F <= kMaxFeatures, "16-byte pieces");
    if (a.ablate & 4194304) return;                              // (profiling bit: the matrix is not written)
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int Q = F / 4;
    constexpr uint32_t kIds = 64u * F * 4u;
    __builtin_amdgcn_wave_barrier();                             // (the tile's staged slices are dead; LDS runs a wave's accesses in order)
    const uint32_t rb = base + (uint32_t)lane * (uint32_t)(F * 4);
#pragma unroll
    for (int q = 0; q < Q; ++q)
        *(UGVC_LDS f32x4*)(uintptr_t)(rb + 16u * q) = f32x4{x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]};
    lds_st32(base + kIds + 4u * (uint32_t)lane, live ? (int32_t)i : -1);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        const uint32_t p = (uint32_t)(k * 64 + lane);
        const uint32_t r = p / (uint32_t)Q, q = p - r * (uint32_t)Q;
        const uint32_t id = lds_u32(base + kIds + 4u * r);
        const f32x4 val = *(UGVC_LDS const f32x4*)(uintptr_t)(base + 16u * p);
        if (id != ~0u) *reinterpret_cast<f32x4*>(a.X + (size_t)id * (size_t)F + 4u * q) = val;
    }
    if (live && a.group) a.group[i] = (uint8_t)group;
    __builtin_amdgcn_wave_barrier();                             // (before the next tile stages its slices here)
}

// ---- the joins of an SNP tile (lanes 0 .. last are its rows, all of contig c_seg: tile_cut; the lanes behind repeat lane 0) ------
// The slices fetched from the carried ranks (`pre`) go to the wave's LDS scratch, sentinel padded; six lock-step descent
// steps (seven for a table staged with 128 rows) rank every table at once; verdicts from a few reads around the rank; a table
// whose staged slice does not reach the tile's last variant is searched in HBM from the carried rank.  Leaves the ranks at
// lane `last` in `bk` (where the wave's next tile starts).  (The name is from the version that joined a tile contig by contig.)
template <int NT>
__device__ __forceinline__ void snp_join_segment(const V5Args& v, const Scratch& sc, Brk<NT>& bk, const SlicePre<NT>& pre, int lane, int pos,
                                                 uint64_t key, int c_seg, int last, JoinOut& jo) {
    const FilterArgs& a = v.f;
    const int pos_max = __builtin_amdgcn_readlane(pos, last);
    int L[NT], plo[NT], phi[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        L[t] = bk.L[t] - 2;
        plo[t] = bk.plo[t];
        phi[t] = bk.phi[t];
    }
    const int Lb0 = bk.Lb;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t == 0 && !a.has_runs) continue;
        const int cap = v.jcap[t];
        const uint32_t dS = sc.base + 4u * (uint32_t)v.joff[t], dE = dS + 4u * (uint32_t)cap;
        // (the staged rows lie inside the contig's rows in all but a contig's first and last tiles: a wave-uniform test
        // instead of two compares and two selects per lane and half)
        const bool inner = L[t] >= plo[t] && L[t] + cap <= phi[t];
        if (inner) {
            lds_st32(dS + 4u * lane, pre.sv[t][0]);
            lds_st32(dE + 4u * lane, pre.ev[t][0]);
            if (cap > 64) {
                lds_st32(dS + 256u + 4u * lane, pre.sv[t][1]);
                lds_st32(dE + 256u + 4u * lane, pre.ev[t][1]);
            }
        } else {
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
    }
    if (a.n_bl > 0) lds_st64(sc.base + 4u * (uint32_t)v.joff[kJoin5 - 1] + 8u * lane, pre.bl);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // rank among the staged starts of every table: seven descent steps in lock-step, no bounds test (sentinels)
    // and no branch (a table staged with 64 entries makes a step of zero at 64)
    uint32_t p[NT], A[NT], maskB[NT];
    const uint32_t Ab = sc.base + 4u * (uint32_t)v.joff[kJoin5 - 1];
    uint32_t pb = Ab - 8u;
    const uint64_t key_max = ((uint64_t)(uint32_t)c_seg << 32) | (uint32_t)pos_max;
    bool miss = false;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        A[t] = sc.base + 4u * (uint32_t)v.joff[t];
        p[t] = A[t] - 4u;
        maskB[t] = 4u * (uint32_t)(v.jcap[t] - 1);
        // a staged slice that does not reach the segment's last variant: that table is searched in HBM (dense stretches)
        if (t > 0 || a.has_runs) miss |= lds_i32(A[t] + maskB[t]) < pos_max;
    }
    if (a.n_bl > 0) miss |= lds_u64(Ab + 8u * (kBlCap5 - 1)) < key_max;
    // (the 64-row step exists only for a table staged with 128 rows - the densest one or two: a wave-uniform branch per table
    // instead of a step of zero for the others)
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (v.jcap[t] > 64) {
            const uint32_t cand = p[t] + 256u;
            p[t] = lds_i32(cand) < pos ? cand : p[t];
        }
    if (kBlCap5 > 64) {
        const uint32_t cb = pb + 512u;
        pb = lds_u64(cb) < key ? cb : pb;
    }
#pragma unroll
    for (int sb = 128; sb >= 4; sb >>= 1) {
        uint32_t cand[NT];
        int x[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            cand[t] = p[t] + (uint32_t)sb;
            x[t] = lds_i32(cand[t]);
        }
        const uint32_t cb = pb + (uint32_t)((2 * sb) & (8 * (kBlCap5 - 1)));
        const uint64_t xk = lds_u64(cb);
#pragma unroll
        for (int t = 0; t < NT; ++t) p[t] = x[t] < pos ? cand[t] : p[t];
        pb = xk < key ? cb : pb;
    }
    int sg_l[NT];
    int rb_l = Lb0 + (int)((pb + 8u - Ab) >> 3);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        sg_l[t] = bk.L[t];
        if (t == 0 && !a.has_runs) continue;
        const uint32_t dE = 4u * (uint32_t)v.jcap[t];
        const int sg = L[t] + (int)((p[t] + 4u - A[t]) >> 2);   // staged starts below pos
        sg_l[t] = sg;
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
            if (__ballot(lds_i32(A[t] + maskB[t]) < pos_max) != 0)
                sg_l[t] = join_one_global(&a, t, max(L[t] + 2, plo[t]), phi[t], plo[t], phi[t], pos, key, &jo);
        }
        if (a.n_bl > 0 && __ballot(lds_u64(Ab + 8u * (kBlCap5 - 1)) < key_max) != 0)
            rb_l = join_one_global(&a, kJoin5 - 1, Lb0, (int)a.n_bl, 0, 0, pos, key, &jo);
    }
    // the next tile of this wave (or the next contig segment of this one) starts where this segment's last variant ended
#pragma unroll
    for (int t = 0; t < NT; ++t) bk.L[t] = __builtin_amdgcn_readlane(sg_l[t], last);
    bk.Lb = __builtin_amdgcn_readlane(rb_l, last);
}

// ---- SNP / MNP tile: features of 64 substitutions (ref_len == alt_len) ---------------------------------
// Writes the flags column, leaves the 16-bit codes of the group-0 forest in the wave's code planes.
template <int NTRK, bool WX>
__device__ __forceinline__ void featurize_snp_tile(const V5Args& v, const Scratch& sc, int lane, uint32_t i, bool live, bool has_model,
                                                   const SnpCols& k, Brk<1 + NTRK>& bk, SlicePre<1 + NTRK>& pre, PhaseClk& pc) {
    constexpr int NT = 1 + NTRK;                               // interval tables: runs + tracks
    const FilterArgs& a = v.f;
    const int c = k.c, pos = k.pos, rl = k.rl;
    const uint32_t ro = k.ro, ao = k.ao;
    const int c0 = rfl(c);
    // Every lane of a tile belongs to ONE contig: the tile loop ends a tile where the contig changes (fused5_kernel: tile_cut)
    // and the padding lanes repeat lane 0's row.  (Round 3 sent a tile that spanned two contigs through per-lane 20-step binary
    // searches in HBM, ~100 dependent round trips: the 23 workgroups that hold a contig boundary were the slowest of every
    // launch, which waited for them - profiles/r04_wave_clk_static.txt.)
    const int n_live = (int)__popcll(__ballot(live));
    const bool joins_on = !(a.ablate & 524288);
    if (c0 != bk.c) {                                           // the wave's first tile, or a new contig: search afresh
        if (!(bk.c >= 0 && brk_adopt<NT>(sc.gtab_b + 512u, bk, c0))) brk_refresh<NT>(a, bk, c0, rfl(pos), lane);
        if (joins_on) issue_slices<NT>(v, bk, lane, pre);
    }
    const int64_t clo = bk.clo, chi = bk.chi;
    const uint32_t clen = (uint32_t)(chi - clo);
    const uint32_t p0 = (uint32_t)(pos - 1);
    const int64_t g0 = clo + p0;
    // 11 bases pos-5 .. pos+5: one dword-aligned 16-byte load, realigned per lane (the buffer is padded by 64
    // bytes at both ends)
    const int64_t wa = (g0 - 5) & ~(int64_t)3;
    const uint32_t sh = (uint32_t)(g0 - 5) & 3u;
    const uint4 xw = load_window16(a.ref + wa);
    const uint32_t rbase = a.alleles[ro], abase = a.alleles[ao];

    const uint64_t key = ((uint64_t)(uint32_t)c << 32) | (uint32_t)pos;
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
    CLK(pc, 0);

    // ---- joins
    JoinOut jo{false, false, false, 0u};
    if (joins_on) snp_join_segment<NT>(v, sc, bk, pre, lane, pos, key, c0, n_live - 1, jo);
    uint8_t flags = (uint8_t)(jo.trk << UGVC_FLAG_TRACK0_SHIFT);
    if (jo.cohort) flags |= UGVC_FLAG_COHORT_FP;
    if (a.mark_hpol && (jo.inside_run || jo.close_run)) flags |= UGVC_FLAG_HPOL_RUN;
    if (live && !WX) stg32(a.flags, i, flags);
    CLK(pc, 2);
    if (!has_model && !WX) return;
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
    const uint32_t gc_code = WX ? 0u : lds_u16(sc.gcr_b + 2u * (gc_len * 11u + (gc_len - n_at)));       // group 0
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

    if (WX) {
        float x[kMaxFeatures];
#pragma unroll
        for (int f = 0; f < kMaxFeatures; ++f) x[f] = 0.f;
        x[0] = qual; x[1] = sor; x[2] = (float)dp; x[3] = (float)adr; x[4] = (float)ada; x[5] = vaf; x[6] = (float)gq;
        x[11] = (float)lm; x[12] = (float)rm;
        x[13] = gc_len > 0 ? (float)((double)(gc_len - n_at) / (double)gc_len) : 0.0f;
        x[14] = (float)css;
        x[15] = jo.inside_run ? 1.f : 0.f;
        x[16] = jo.close_run ? 1.f : 0.f;
#pragma unroll
        for (int t = 0; t < UGVC_MAX_TRACKS; ++t) x[UGVC_N_BASE_FEATURES + t] = (jo.trk >> t) & 1u ? 1.f : 0.f;
        if constexpr (((UGVC_N_BASE_FEATURES + NTRK) & 3) == 0) store_feature_rows_tile<UGVC_N_BASE_FEATURES + NTRK>(a, sc.base, lane, i, live, x, 0);
        else store_feature_row(a, i, live, x, 0);
        return;
    }
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
__device__ __forceinline__ void issue_indel_slices(const V5Args& v, const Brk<1 + NTRK>& bk, int lane, IndelPre<1 + NTRK>& pre) {
    constexpr int NT = 1 + NTRK;
    const FilterArgs& a = v.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        pre.sv[t][0] = pre.sv[t][1] = pre.ev[t][0] = pre.ev[t][1] = 0;
        if ((t == 0 && !a.has_runs) || ((v.iwide >> t) & 1)) continue;      // (dense tables: six rows per lane, fetched at the joins)
        const TrackView& tv = table_view(a, t);
        const int top = max(v.na[t] - 1, 0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t gs = (uint32_t)max(min(bk.L[t] - 2 + 64 * h + lane, top), 0);
            pre.sv[t][h] = ldg32(tv.starts, gs); pre.ev[t][h] = ldg32(tv.ends, gs);
        }
    }
    pre.bl[0] = pre.bl[1] = ~0ull;
    if (a.n_bl > 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t gi = (int64_t)bk.Lb + 64 * h + lane;
            if (gi < a.n_bl) pre.bl[h] = ldg32(a.bl, (uint32_t)gi);
        }
    }
}

struct IndelCols {                  // the columns of one indel (fetched one tile ahead of their use)
    int c, pos, rl, al;
    uint32_t ro, ao;
    float qual, sor;
    int dp, adr, ada, gq;
};

__device__ __forceinline__ IndelCols load_indel_cols(const FilterArgs& a, uint32_t i) {
    IndelCols k;
    k.c = ldg32(a.contig, i); k.pos = ldg32(a.pos, i); k.rl = ldg32(a.ref_len, i); k.al = ldg32(a.alt_len, i);
    k.ro = ldg32(a.ref_off, i); k.ao = ldg32(a.alt_off, i);
    k.qual = ldg32(a.qual, i); k.sor = ldg32(a.sor, i);
    k.dp = ldg32(a.dp, i); k.adr = ldg32(a.ad_ref, i); k.ada = ldg32(a.ad_alt, i); k.gq = ldg32(a.gq, i);
    return k;
}

// `touch_next`: the caller's loads for the NEXT tile (columns, row indices - requested at the top of this tile) are named in an empty
// asm just before this tile's last stores.  The loop copies them into its registers at its latch, behind those stores and behind
// the slice loads requested after this function: the wait-count pass cannot order them across the loop's branches, waits for
// EVERYTHING there, and the slice loads' whole round trip was exposed at the end of every tile (round 4; ~4 k of a tile's 38 k
// cycles).  Waited for here, they have been in flight for a whole tile and nothing younger is.
template <int NTRK, bool WX, class TouchNext>
__device__ __forceinline__ void featurize_indel_tile(const V5Args& v, const Scratch& sc, uint32_t rshard, int split, int lane, uint32_t i, bool live,
                                                     const IndelCols& k, Brk<1 + NTRK>& bk, IndelPre<1 + NTRK>& pre, PhaseClk& pc, TouchNext touch_next) {
    constexpr int NT = 1 + NTRK;
    const FilterArgs& a = v.f;
    const uint8_t* __restrict__ apool = a.alleles;
    const bool joins_on = !(a.ablate & 524288);
    const int c = k.c, pos = k.pos, rl = k.rl, al = k.al;
    const uint32_t ro = k.ro, ao = k.ao;
    const float qual = k.qual, sor = k.sor;
    const int dp = k.dp, adr = k.adr, ada = k.ada, gq = k.gq;
    const int c0 = rfl(c);
    if (c0 != bk.c) {                                           // the wave's first tile, or a new contig: search afresh
        if (!(bk.c >= 0 && brk_adopt<NT>(sc.gtab_b + 512u, bk, c0))) brk_refresh<NT>(a, bk, c0, rfl(pos), lane);
        if (joins_on) issue_indel_slices<NTRK>(v, bk, lane, pre);
    }
    const int n_live = (int)__popcll(__ballot(live));
    const bool ins = rl < al;
    const int classify = ins ? 1 : 2;
    const int indel_length = ins ? al - rl : rl - al;
    const int64_t clo = bk.clo, chi = bk.chi;                  // (one contig per tile: fused5_kernel, tile_cut)
    const uint32_t clen = (uint32_t)(chi - clo);
    const uint32_t p0 = (uint32_t)(pos - 1);
    const int64_t g0 = clo + p0;
    int64_t ws = (g0 - 6) & ~(int64_t)15;
    if (ws < 0) ws = 0;
    const int o0 = (int)(g0 - ws);                            // byte of the variant's first base, 6..21 (less at genome start)
    const uint32_t wrow_b = sc.base + (uint32_t)(lane * kWinRowB);
    // allele bytes: the tail of the longer allele.  Requested BEFORE the window: the window's rows go to LDS behind a wait for
    // everything in flight, and loads written after that wait were a second dependent round trip of every tile (round 4).
    const uint32_t lo_off = ins ? ao : ro;
    const int ln = ins ? al : rl;
    // (bases 1..8 of the allele as ONE unaligned 8-byte load - the pool is padded by 16 bytes, api.hip: ugvc_upload_variants)
    const U2u abw = *reinterpret_cast<const U2u*>(apool + lo_off + 1);
    {
        const uint4 x0 = load_window16(a.ref + ws), x1 = load_window16(a.ref + ws + 16), x2 = load_window16(a.ref + ws + 32);
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
        const int bb = (int)(abw.x & 0xFFu);
        const uint32_t pat = (uint32_t)bb * 0x01010101u;
        // bytes 0 .. n - 1 of a loaded 8-byte word against the first base, n = bases of the allele in the word
        auto same8 = [&](uint32_t lo, uint32_t hi, int n) {
            const uint32_t mlo = n >= 4 ? ~0u : (n <= 0 ? 0u : (1u << (8 * n)) - 1u);
            const uint32_t mhi = n >= 8 ? ~0u : (n <= 4 ? 0u : (1u << (8 * (n - 4))) - 1u);
            return (((lo ^ pat) & mlo) | ((hi ^ pat) & mhi)) == 0u;
        };
        bool mono = same8(abw.x, abw.y, ln - 1);
        // (alleles longer than nine bases: eight bases per round trip - one dependent round trip per BYTE of the
        // tile's longest allele until round 4)
        if (__ballot(ln > 9) != 0) {
            const int ln_max = wave_max_i32(ln);
            for (int q0 = 9; q0 < ln_max; q0 += 8) {
                const U2u x = *reinterpret_cast<const U2u*>(apool + lo_off + (uint32_t)min(q0, ln - 1));
                mono &= same8(x.x, x.y, ln - q0);
            }
        }
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
    const bool pg_ok = !WX && (group == 1 ? v.pg[1].ok != 0 : v.pg[2].ok != 0);
    if (live && !pg_ok && !WX) {                               // no model for this variant type: score 0, PASS
        a.score[i] = 0.f;
        a.filter[i] = UGVC_FILTER_PASS;
    }
    // ---- record slots: one returning atomic per wave and group (issued here: its round trip runs under the joins)
    const bool mine = live && pg_ok;
    // A tile is a run of <= 64 list entries from any offset (tile_cut): its first `split` lanes belong to the 64-entry block
    // `rshard`, the others to the next one - each lane's record goes to ITS block's shard, so a (workgroup, block) pair never
    // holds more than 64 records (what the shards are sized for); an aligned tile (split = 64: all but the few behind a contig
    // boundary) issues the two atomics of round 3.
    const unsigned long long lowm = split >= 64 ? ~0ull : ((1ull << split) - 1ull);
    const unsigned long long m1 = __ballot(mine && group == 1), m2 = __ballot(mine && group == 2);
    const bool hi_blk = lane >= split;
    const int shard0 = (int)(rshard & (kShards - 1)), shard1 = (int)((rshard + 1u) & (kShards - 1));
    const int shard = hi_blk ? shard1 : shard0;
    // (ONE atomic instruction, executed by lanes 1..4 for the (group, block) pairs that have records: four `if (lane == k) got =
    // atomicAdd(..)` statements compiled to four dependent round trips - each waits for the one before, `got` being one register)
    unsigned got = 0;
    {
        const unsigned c0 = (unsigned)__popcll(m1 & lowm), c1 = (unsigned)__popcll(m2 & lowm), c2 = (unsigned)__popcll(m1 & ~lowm),
                       c3 = (unsigned)__popcll(m2 & ~lowm);
        const unsigned i0 = (unsigned)((1 * kShards + shard0) * kCounterStride), i1 = (unsigned)((2 * kShards + shard0) * kCounterStride),
                       i2 = (unsigned)((1 * kShards + shard1) * kCounterStride), i3 = (unsigned)((2 * kShards + shard1) * kCounterStride);
        const unsigned cnt_k = lane == 1 ? c0 : (lane == 2 ? c1 : (lane == 3 ? c2 : (lane == 4 ? c3 : 0u)));
        const unsigned idx_k = lane == 1 ? i0 : (lane == 2 ? i1 : (lane == 3 ? i2 : i3));
        if (cnt_k != 0) got = atomicAdd(&v.counters[idx_k], cnt_k);
    }
    const unsigned long long below = (1ull << lane) - 1;
    const unsigned long long mine_blk = (group == 1 ? m1 : m2) & (hi_blk ? ~lowm : lowm);
    const unsigned grank = (unsigned)__popcll(mine_blk & below);

    CLK(pc, 8);                                                 // (hmer run, record slots requested)
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
    // the joins of the tile (lanes 0 .. last are live, contig c_seg)
    auto join_seg = [&](int c_seg, int last, JoinOut& jo) {
        const int pos_max = __builtin_amdgcn_readlane(pos, last);
        const uint32_t s0 = sc.base, s1 = sc.base + kIndelSlotB;
        auto on = [&](int t) { return t > 0 || a.has_runs; };
        auto is_wide = [&](int t) { return ((v.iwide >> t) & 1) != 0; };
        // a staged slice reaches the tile's last variant if its last row does (or is padding); else that table is
        // searched in HBM from the carried rank
        auto covers = [&](uint32_t last_row_b) { return rfl(lds_i32(last_row_b)) >= pos_max; };
        int sg_l[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) sg_l[t] = bk.L[t];
        int rb_l = bk.Lb;
        // the first dense table: its rows are requested when the narrow slices have been staged and arrive under the narrow
        // rounds.  (Requested BEFORE the staging, as until round 4, they were waited for there: the stores of the staged rows wait for
        // everything in flight - the loop's back edge hides the order of the requests from the wait-count pass.)
        int tw = -1;
        int wv[kWideChunks], we[kWideChunks];
        auto request_wide = [&]() {
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (tw < 0 && on(t) && is_wide(t)) {
                    tw = t;
                    wide_load(table_view(a, t), bk.L[t] - 2, max(v.na[t] - 1, 0), lane, wv, we);
                }
        };
        auto narrow = [&](uint32_t slot, int t) {
            if (covers(slot + 4u * (kIndelRows - 1))) sg_l[t] = staged_verdict(a, slot, t, bk.L[t] - 2, bk.plo[t], bk.phi[t], pos, jo);
            else sg_l[t] = join_one_global(&a, t, max(bk.L[t], bk.plo[t]), bk.phi[t], bk.plo[t], bk.phi[t], pos, key, &jo);
        };
        const bool bl_on = a.n_bl > 0;
        if (v.indel_one_round) {
            // every narrow slice and the blacklist keys side by side (1 KB each), ranked in lock-step
            uint32_t slot[NT], p[NT];
            uint32_t nslot = 0;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                slot[t] = s0;
                if (!on(t) || is_wide(t)) continue;
                slot[t] = s0 + kIndelSlotB * nslot++;
                stage_rows(slot[t], bk.L[t] - 2, bk.plo[t], bk.phi[t], pre.sv[t], pre.ev[t], lane);
            }
            const uint32_t sb_b = s0 + kIndelSlotB * nslot;
            if (bl_on) {
                lds_st64(sb_b + 8u * lane, pre.bl[0]);
                lds_st64(sb_b + 512u + 8u * lane, pre.bl[1]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            request_wide();
            CLK(pc, 9);                                         // (narrow slices staged)
            uint32_t pb = sb_b - 8u;
#pragma unroll
            for (int t = 0; t < NT; ++t) p[t] = slot[t] - 4u;
#pragma unroll
            for (int sb = 256; sb >= 4; sb >>= 1) {
                uint32_t cand[NT];
                int x[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    cand[t] = p[t] + (uint32_t)sb;
                    x[t] = (on(t) && !is_wide(t)) ? lds_i32(cand[t]) : 0;
                }
                const uint32_t cb = pb + 2u * (uint32_t)sb;
                const uint64_t xk = bl_on ? lds_u64(cb) : 0ull;
#pragma unroll
                for (int t = 0; t < NT; ++t) p[t] = x[t] < pos ? cand[t] : p[t];
                pb = xk < key ? cb : pb;
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (!on(t) || is_wide(t)) continue;
                if (covers(slot[t] + 4u * (kIndelRows - 1)))
                    sg_l[t] = staged_verdict_at(a, slot[t], t, bk.L[t] - 2, bk.plo[t], bk.phi[t], pos, (p[t] + 4u - slot[t]) >> 2, jo);
                else sg_l[t] = join_one_global(&a, t, max(bk.L[t], bk.plo[t]), bk.phi[t], bk.plo[t], bk.phi[t], pos, key, &jo);
            }
            if (bl_on) {
                const uint64_t key_max = ((uint64_t)(uint32_t)c_seg << 32) | (uint32_t)pos_max;
                if (__ballot(lds_u64(sb_b + 8u * (kIndelRows - 1)) < key_max) == 0) {
                    if (lds_u64(pb + 8u) == key) jo.cohort = true;
                    rb_l = bk.Lb + (int)((pb + 8u - sb_b) >> 3);
                } else rb_l = join_one_global(&a, kJoin5 - 1, bk.Lb, (int)a.n_bl, 0, 0, pos, key, &jo);
            }
        } else {
        // round A: runs | blacklist
        __builtin_amdgcn_wave_barrier();
        if (on(0) && !is_wide(0)) stage_rows(s0, bk.L[0] - 2, bk.plo[0], bk.phi[0], pre.sv[0], pre.ev[0], lane);
        if (bl_on) {
            lds_st64(s1 + 8u * lane, pre.bl[0]);
            lds_st64(s1 + 512u + 8u * lane, pre.bl[1]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        request_wide();
        if (on(0) && !is_wide(0)) narrow(s0, 0);
        if (bl_on) {
            const uint64_t key_max = ((uint64_t)(uint32_t)c_seg << 32) | (uint32_t)pos_max;
            const uint64_t last_key = lds_u64(s1 + 8u * (kIndelRows - 1));
            if (__ballot(last_key < key_max) == 0) {
                uint32_t pb = s1 - 8u;
#pragma unroll
                for (int sb = 512; sb >= 8; sb >>= 1) {
                    const uint32_t cand = pb + (uint32_t)sb;
                    pb = lds_u64(cand) < key ? cand : pb;
                }
                if (lds_u64(pb + 8u) == key) jo.cohort = true;
                rb_l = bk.Lb + (int)((pb + 8u - s1) >> 3);
            } else rb_l = join_one_global(&a, kJoin5 - 1, bk.Lb, (int)a.n_bl, 0, 0, pos, key, &jo);
        }
        // tracks, in pairs
#pragma unroll
        for (int t = 1; t < NT; t += 2) {
            const int t2 = t + 1 < NT ? t + 1 : t;
            __builtin_amdgcn_wave_barrier();
            if (!is_wide(t)) stage_rows(s0, bk.L[t] - 2, bk.plo[t], bk.phi[t], pre.sv[t], pre.ev[t], lane);
            if (t + 1 < NT && !is_wide(t2)) stage_rows(s1, bk.L[t2] - 2, bk.plo[t2], bk.phi[t2], pre.sv[t2], pre.ev[t2], lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (!is_wide(t)) narrow(s0, t);
            if (t + 1 < NT && !is_wide(t2)) narrow(s1, t2);
        }
        }
        CLK(pc, 10);                                            // (narrow tables ranked, verdicts)
        // the dense tables: six rows per lane, the whole scratch, one at a time
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (!on(t) || !is_wide(t)) continue;
            if (t != tw) wide_load(table_view(a, t), bk.L[t] - 2, max(v.na[t] - 1, 0), lane, wv, we);
            wide_store(s0, bk.L[t] - 2, bk.plo[t], bk.phi[t], lane, wv, we);
            if (covers(s0 + 4u * (kWideRows - 1))) sg_l[t] = wide_verdict(a, s0, t, bk.L[t] - 2, bk.plo[t], bk.phi[t], pos, jo);
            else sg_l[t] = join_one_global(&a, t, max(bk.L[t], bk.plo[t]), bk.phi[t], bk.plo[t], bk.phi[t], pos, key, &jo);
        }
        // the next tile of this wave starts where this tile's last variant ended
#pragma unroll
        for (int t = 0; t < NT; ++t) bk.L[t] = __builtin_amdgcn_readlane(sg_l[t], last);
        bk.Lb = __builtin_amdgcn_readlane(rb_l, last);
    };
    if (joins_on) join_seg(c0, n_live - 1, jo);
    uint8_t flags = (uint8_t)(jo.trk << UGVC_FLAG_TRACK0_SHIFT);
    if (jo.cohort) flags |= UGVC_FLAG_COHORT_FP;
    if (a.mark_hpol && (jo.inside_run || jo.close_run)) flags |= UGVC_FLAG_HPOL_RUN;
    if (live && !WX) stg32(a.flags, i, flags);
    CLK(pc, 2);
    if (WX) {
        float x[kMaxFeatures];
#pragma unroll
        for (int f = 0; f < kMaxFeatures; ++f) x[f] = 0.f;
        x[0] = qual; x[1] = sor; x[2] = (float)dp; x[3] = (float)adr; x[4] = (float)ada;
        x[5] = dp > 0 ? __fdiv_rn((float)ada, (float)dp) : 0.0f;
        x[6] = (float)gq; x[7] = (float)classify; x[8] = (float)indel_length; x[9] = (float)hmer_len; x[10] = (float)hmer_nuc;
        x[11] = (float)lm; x[12] = (float)rm;
        x[13] = gc_len > 0 ? (float)((double)gc_cnt / (double)gc_len) : 0.0f;
        x[14] = 3.f;                                               // cycle skip: NA for indels
        x[15] = jo.inside_run ? 1.f : 0.f;
        x[16] = jo.close_run ? 1.f : 0.f;
#pragma unroll
        for (int t = 0; t < UGVC_MAX_TRACKS; ++t) x[UGVC_N_BASE_FEATURES + t] = (jo.trk >> t) & 1u ? 1.f : 0.f;
        touch_next();
        if constexpr (((UGVC_N_BASE_FEATURES + NTRK) & 3) == 0) store_feature_rows_tile<UGVC_N_BASE_FEATURES + NTRK>(a, sc.base, lane, i, live, x, group);
        else store_feature_row(a, i, live, x, group);
        return;
    }
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
    const unsigned b1 = __shfl(got, 1), b2 = __shfl(got, 2), b3 = __shfl(got, 3), b4 = __shfl(got, 4);
    touch_next();
    if (mine) {
        const unsigned slot0 = group == 1 ? (hi_blk ? b3 : b1) : (hi_blk ? b4 : b2);
        // (a select between the two pointers: `v.rec5[group]` with the lane's group is a VECTOR load from the launch arguments and a wait)
        uint4* const rec_g = group == 1 ? v.rec5[1] : v.rec5[2];
        uint4* dst = rec_g + ((size_t)shard * v.shard_cap5 + slot0 + grank) * 3;
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
    // The p1 table sits at LDS address 0 (first table of the image, no static __shared__ in either kernel - both trap at
    // entry otherwise): a payload IS the address of the leaf's class-1 probability.  (Written as `p1_b + payload` the add
    // survives as `v_add 0`: the LDS address is assigned after the optimiser has run.)
    (void)p1_b;
    // (the last level's entry of heap node I is I - T H: the subtraction is folded into the table's base)
    const uint32_t last_base = (uint32_t)rfl((int)(last_b - 8u * (uint32_t)(T * H)));
    const uint32_t* __restrict__ roots = pg.roots;
    double a1 = 0.0;
    int t = 0;
    auto one_tree = [&](int tt) -> uint32_t {
        if (D == 1) return stump_payload(last_b, planes_lane_b, tt);
        uint32_t pi[1];
        walk6<1>(hi_b, last_base, planes_lane_b, roots, tt, T, D, pi);
        return pi[0];
    };
    if (D > 1) {
    if (NTM > 8) {
        // more trees in flight per lane: in the fused kernel only part of a CU's waves walk at any time, so a walking
        // wave has to keep more LDS requests outstanding to fill the pipeline
        for (; t + NTM <= T; t += NTM) {
            uint32_t pi[NTM];
            walk6<NTM>(hi_b, last_base, planes_lane_b, roots, t, T, D, pi);
            double pv[NTM];
#pragma unroll
            for (int q = 0; q < NTM; ++q) pv[q] = lds_f64(pi[q]);
#pragma unroll
            for (int q = 0; q < NTM; ++q) a1 += pv[q];
        }
    }
    for (; t + 8 <= T; t += 8) {
        uint32_t pi[8];
        walk6<8>(hi_b, last_base, planes_lane_b, roots, t, T, D, pi);
        double pv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) pv[q] = lds_f64(pi[q]);
#pragma unroll
        for (int q = 0; q < 8; ++q) a1 += pv[q];
    }
    }
    for (; t < T; ++t) a1 += lds_f64(one_tree(t));
    const double half = 0.5 * (double)T, band = pg.band;
    // score = (float)(a1 / T), the reference's mean of the trees' class-1 probabilities.  The f64 division is ~20 vector
    // instructions; a1 * (1 / T) is within two double ulps of the quotient, so the two round to the same float unless the product
    // sits that close to the midpoint of two floats (bits 28..0 of the double's mantissa around 0x10000000): then - 2^-26 of the
    // lanes - the whole wave divides.
    {
        const double xq = a1 * pg.inv_T;
        const uint32_t low = (uint32_t)__double2loint(xq) & 0x1FFFFFFFu;
        const bool near_mid = low - (0x10000000u - 8u) <= 16u;
        score = __builtin_amdgcn_ballot_w64(near_mid) != 0 ? (float)(a1 / (double)T) : (float)xq;
    }
    filt = a1 > half ? UGVC_FILTER_PASS : UGVC_FILTER_LOW_SCORE;
    // inside the band around T/2 (rounding of the two class sums, model_pack.hip) the class-0 sum decides as
    // scikit-learn's argmax does: redo the walk with both payload sums, in tree order (exact ties in practice)
    if (__builtin_amdgcn_ballot_w64(fabs(a1 - half) <= band) != 0) {
        double b0 = 0.0, b1 = 0.0;
        for (int tt = 0; tt < T; ++tt) {
            const double2 pv = pg.pairs[one_tree(tt) >> 3];
            b0 += pv.x; b1 += pv.y;
        }
        const double q0 = b0 / (double)T, q1 = b1 / (double)T;
        if (fabs(a1 - half) <= band) filt = q1 > q0 ? UGVC_FILTER_PASS : UGVC_FILTER_LOW_SCORE;
    }
}

// A forest's three tables (hi | last | p1, each padded to 16 bytes) -> one contiguous LDS image, 16 bytes per thread and
// trip.  The first KB trips' loads are ALL issued before the first store: written as a plain loop with a three-way source
// select the copy ran one dependent round trip per trip (six for the 91 KB SNP forest: ~9 us of every launch at any size).
// The first KB trips' loads are ALL issued before the first store, UNPREDICATED (indices clamped into the tables, sources
// chosen by selects): a load under `if (q < total)` compiles to a branch with `s_waitcnt vmcnt(0)` behind it - one dependent
// round trip per trip, six for the 91 KB SNP forest, ~9 us of every launch at any size (round 3, and the first version of
// this function).  ForestLoads keeps the pieces between issue() and store() so that other loads can be issued in between.
template <int KB>
struct ForestLoads {
    uint4 r[KB];
    __device__ __forceinline__ void issue(const uint4* s0, size_t n0, const uint4* s1, size_t n1, const uint4* s2, size_t n2, int tid, int nthreads) {
        const size_t total = n0 + n1 + n2;
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            size_t q = (size_t)tid + (size_t)k * nthreads;
            q = q < total ? q : (total ? total - 1 : 0);
            const uint4* p = q < n0 ? s0 + q : (q < n0 + n1 ? s1 + (q - n0) : s2 + (q - n0 - n1));
            r[k] = total ? *p : make_uint4(0, 0, 0, 0);
        }
    }
    __device__ __forceinline__ void store(unsigned char* dst, const uint4* s0, size_t n0, const uint4* s1, size_t n1, const uint4* s2, size_t n2, int tid,
                                          int nthreads) const {
        const size_t total = n0 + n1 + n2;
        uint4* d = reinterpret_cast<uint4*>(dst);
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            const size_t q = (size_t)tid + (size_t)k * nthreads;
            if (q < total) d[q] = r[k];
        }
        for (size_t q = (size_t)tid + (size_t)KB * nthreads; q < total; q += nthreads)
            d[q] = q < n0 ? s0[q] : (q < n0 + n1 ? s1[q - n0] : s2[q - n0 - n1]);
    }
};

template <int KB>
__device__ __forceinline__ void fill_forest_lds(unsigned char* dst, const uint4* s0, size_t n0, const uint4* s1, size_t n1, const uint4* s2,
                                                size_t n2, int tid, int nthreads) {
    ForestLoads<KB> f;
    f.issue(s0, n0, s1, n1, s2, n2, tid, nthreads);
    f.store(dst, s0, n0, s1, n1, s2, n2, tid, nthreads);
}

// LDS of a workgroup: group forest (hi | last | p1, 16-byte padded) | group 0's level-order threshold trees | the
// indel groups' sorted thresholds (skewed) | gc rank codes | css | wave scratch
struct Lds5 {
    uint32_t hi_b, last_b, p1_b, eyt_b, thr_b, gcr_b, css_b, gtab_b, scratch_b;
};

__device__ __forceinline__ Lds5 lds5_fill(unsigned char* smem, const V5Args& v, bool with_forest, int tid, int nthreads, bool model_tables = true) {
    const PackedGroupView& pg = v.pg[0];
    Lds5 L;
    size_t off = 0;
    const size_t n_hi = with_forest ? ((size_t)pg.T << pg.D) / 2 : 0;
    const size_t b_hi = (n_hi * 4 + 15) & ~(size_t)15, b_last = (n_hi * 8 + 15) & ~(size_t)15;
    const size_t b_p1 = with_forest ? (((size_t)pg.n_pairs * 8 + 15) & ~(size_t)15) : 0;
    // (p1 FIRST: its base is a compile-time LDS address - the walk's final read takes the payload as its address and the
    // base in the instruction's offset field)
    L.p1_b = lds_addr(smem);
    L.hi_b = lds_addr(smem + b_p1);
    L.last_b = lds_addr(smem + b_p1 + b_hi);
    // ---- every global load of the fill is issued here, unpredicated, before the first LDS store (round 4: each table used to
    // be its own load -> wait -> store loop, ~8 dependent round trips in front of the first tile)
    ForestLoads<6> fl;
    fl.issue(reinterpret_cast<const uint4*>(pg.p1), b_p1 / 16, reinterpret_cast<const uint4*>(pg.hi4), b_hi / 16,
             reinterpret_cast<const uint4*>(pg.last4), b_last / 16, tid, nthreads);
    const int n_eyt4 = v.eyt_len / 4;
    const float4 r_eyt = n_eyt4 > 0 ? reinterpret_cast<const float4*>(v.eyt)[min(tid, n_eyt4 - 1)] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int n_thr = v.thr_lds_len - v.thr0_len;                                          // groups 1 and 2
    constexpr int KT = 4;                                                                  // (kThr3 = 3584 floats at most: four per thread)
    float r_thr[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) r_thr[k] = n_thr > 0 ? v.thr[v.thr0_len + min(tid + k * nthreads, n_thr - 1)] : 0.f;
    const uint32_t r_gcr = model_tables ? reinterpret_cast<const uint32_t*>(v.gcr)[min(tid, kGcRankBytes / 4 - 1)] : 0u;
    const uint8_t r_css = v.css_lut[min(tid, 255)];
    uint2 r_desc = make_uint2(0u, 0u);
    {
        const int q = min(max(tid - 64, 0), 5), g = 1 + q / 3, sfeat = q % 3;
        const int fj = sfeat == 0 ? 0 : (sfeat == 1 ? 1 : 5);
        if (model_tables) r_desc = v.desc3[g * kMaxFeatures + fj];      // (wave-uniform condition: no per-lane branch, no wait behind it)
    }
    // ---- the stores
    off = b_hi + b_last + b_p1;
    if (with_forest)
        fl.store(smem, reinterpret_cast<const uint4*>(pg.p1), b_p1 / 16, reinterpret_cast<const uint4*>(pg.hi4), b_hi / 16,
                 reinterpret_cast<const uint4*>(pg.last4), b_last / 16, tid, nthreads);
    float* eyt_l = reinterpret_cast<float*>(smem + off);
    if (tid < n_eyt4) reinterpret_cast<float4*>(eyt_l)[tid] = r_eyt;
    for (int q = tid + nthreads; q < n_eyt4; q += nthreads) reinterpret_cast<float4*>(eyt_l)[q] = reinterpret_cast<const float4*>(v.eyt)[q];
    L.eyt_b = lds_addr(eyt_l);
    off += (size_t)v.eyt_len * 4;
    float* thr_l = reinterpret_cast<float*>(smem + off);
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        const int q = tid + k * nthreads;
        if (q < n_thr) thr_l[q + (q >> 5)] = r_thr[k];                                     // skewed: element j at j + (j >> 5)
    }
    for (int q = tid + KT * nthreads; q < n_thr; q += nthreads) thr_l[q + (q >> 5)] = v.thr[v.thr0_len + q];
    L.thr_b = lds_addr(thr_l);
    off += ((size_t)(n_thr + (n_thr >> 5) + 1) * 4 + 15) & ~(size_t)15;
    uint16_t* gcr = reinterpret_cast<uint16_t*>(smem + off);
    if (model_tables && tid < kGcRankBytes / 4) reinterpret_cast<uint32_t*>(gcr)[tid] = r_gcr;   // (built on the host: model_pack.hip)
    L.gcr_b = lds_addr(gcr);
    off += kGcRankBytes;
    uint8_t* css = smem + off;
    if (tid < 256) css[tid] = r_css;
    L.css_b = lds_addr(css);
    off += 256;
    // per indel group: the clamps of the integer features (u16 [2][16]) and the (offset, length) of the three float
    // features' threshold slices in the staged table - read per lane by its group instead of scalar loads + selects
    unsigned char* gtab = smem + off;
    if (tid < 32) reinterpret_cast<uint16_t*>(gtab)[tid] = (uint16_t)v.cap5[1 + (tid >> 4)][tid & 15];
    if (model_tables && tid >= 64 && tid < 70) {
        const int q = tid - 64;
        reinterpret_cast<uint32_t*>(gtab + 64)[2 * q] = (r_desc.x & 0xFFFFFu) - (uint32_t)v.thr0_len;      // the staged table starts at group 1
        reinterpret_cast<uint32_t*>(gtab + 64)[2 * q + 1] = r_desc.y & 0xFFFFu;
    }
    L.gtab_b = lds_addr(gtab);
    off += kGtabBytes;
    L.scratch_b = lds_addr(smem + off);
    return L;
}

// (a workgroup may hold all 160 KiB of a CU's LDS; 158 until round 4, whose boundary words - kGtabBytes - needed 256 bytes more)
constexpr int kLds5Limit = 159 * 1024;
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

// ---- a tile ends where the contig changes (round 4) ----------------------------------------------------------
// A wave's tiles are runs of <= 64 consecutive entries of its workgroup's list, taken from a list OFFSET: when the live lanes of
// a tile hold two contigs (one tile per contig boundary and class) the tile is cut after the first contig's rows - the others
// wait for the next tile, which starts at the boundary - so that featurize_*_tile only ever sees one contig.  The lanes cut off
// repeat lane 0's row, like the padding lanes of a list's last tile.  Returns the number of entries the tile consumes.
__device__ __forceinline__ int wave_max_i32(int x) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x = max(x, __shfl_xor(x, d));
    return rfl(x);
}
__device__ __forceinline__ float rfl_f(float x) { return __int_as_float(rfl(__float_as_int(x))); }
__device__ __forceinline__ int tile_cut(bool& live, uint32_t& i, SnpCols& k) {
    const int c0 = rfl(k.c);
    const unsigned long long lv = __ballot(live), same = __ballot(live && k.c == c0);
    if (same == lv) return 64;
    const int n0 = (int)__popcll(same);
    live = live && k.c == c0;                                       // (rows are sorted by contig: a prefix of the live lanes)
    if (!live) {
        i = (uint32_t)rfl((int)i);
        k.c = c0; k.pos = rfl(k.pos); k.rl = rfl(k.rl); k.ro = (uint32_t)rfl((int)k.ro); k.ao = (uint32_t)rfl((int)k.ao);
        k.qual = rfl_f(k.qual); k.sor = rfl_f(k.sor); k.dp = rfl(k.dp); k.adr = rfl(k.adr); k.ada = rfl(k.ada); k.gq = rfl(k.gq);
    }
    return n0;
}
__device__ __forceinline__ int tile_cut(bool& live, uint32_t& i, IndelCols& k) {
    const int c0 = rfl(k.c);
    const unsigned long long lv = __ballot(live), same = __ballot(live && k.c == c0);
    if (same == lv) return 64;
    const int n0 = (int)__popcll(same);
    live = live && k.c == c0;
    if (!live) {
        i = (uint32_t)rfl((int)i);
        k.c = c0; k.pos = rfl(k.pos); k.rl = rfl(k.rl); k.al = rfl(k.al); k.ro = (uint32_t)rfl((int)k.ro); k.ao = (uint32_t)rfl((int)k.ao);
        k.qual = rfl_f(k.qual); k.sor = rfl_f(k.sor); k.dp = rfl(k.dp); k.adr = rfl(k.adr); k.ada = rfl(k.ada); k.gq = rfl(k.gq);
    }
    return n0;
}

// The pass clock's end words (ugvc_pass_clock): left by workgroup 0's first wave on EVERY way out - also when it has no tiles, or
// works on indel tiles (an indel-only first workgroup): round 5 wrote them on the SNP path alone and the probe then failed with
// "no clock words came back" (ADVICE r5).
__device__ __forceinline__ void pass_clock_end(const V5Args& v) {
    if (!v.wave_clk) return;
    v.wave_clk[1] = __builtin_amdgcn_s_memrealtime();
    v.wave_clk[3] = __builtin_readcyclecounter();
}

// ---- Kf: persistent, one workgroup per CU; every wave works through tiles on its own --------------------
// A workgroup owns `rows_wg` consecutive rows of the callset.  Prologue (the only workgroup barriers): the forest
// and the threshold tables go to LDS while every wave counts the variant classes of its sixteenth of the rows; the
// wave counts are scanned and the rows are written as two dense lists of row indices - substitutions (ref_len ==
// alt_len: the SNP forest) and indels - in callset order.  A tile is 64 consecutive entries of a list: pure in
// class (every lane of a wave walks the SAME forest), ascending in position.  Waves then take CONSECUTIVE tiles:
// the first `n_sw` waves share the SNP tiles, the others the indel tiles, in proportion to the tile counts.
// (Round 6 compiled the WX instantiation for 6 waves per SIMD - 80 registers, three 8-wave workgroups per CU - to hide more of the
// matrix build's round trips: 168 spilled registers, 0.205 against 0.152 ms per 2 M rows, profiles/r06_fm_dense_ab.txt.  Not kept.)
template <int NTRK, int NTW, bool WX = false>
__global__ __launch_bounds__(kK2Threads) void fused5_kernel(const V5Args v) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NT = 1 + NTRK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = rfl(tid >> 6);
    const int n_waves = blockDim.x >> 6;
    const FilterArgs& a = v.f;
#ifdef UGVC_PHASE_CLOCK
    const uint64_t k_begin = __builtin_readcyclecounter();
#endif
    if (lds_addr(smem) != 0u) __builtin_trap();                  // (walk_forest: payloads are absolute LDS addresses)
    const uint64_t wclk_entry = v.wave_clk ? __builtin_readcyclecounter() : 0;
    // (the sustained shader clock of the pass: workgroup 0's first wave leaves the constant 100 MHz counter and the shader-clock
    // counter at its entry and at its end behind the per-wave records - ugvc_pass_clock; stored at once, nothing stays live)
    if (v.wave_clk && blockIdx.x == 0 && tid == 0) {
        v.wave_clk[0] = __builtin_amdgcn_s_memrealtime();          // (the four clock words sit in FRONT of the per-wave records)
        v.wave_clk[2] = __builtin_readcyclecounter();
    }
    const int64_t r0 = (int64_t)blockIdx.x * v.rows_wg;
    if (r0 >= a.n) return;                                       // (uniform: before the first barrier)
    const int64_t r1 = min(r0 + (int64_t)v.rows_wg, a.n);
    const PackedGroupView& pg0 = v.pg[0];
    const bool has0 = !WX && pg0.ok != 0;
    // the thresholds of every group: SNP tiles rank against group 0's slices (the head of the table), indel
    // tiles against their own group's
    // ---- classes of this wave's rows (eight groups of 64 in flight).  The first eight groups are requested BEFORE the
    // LDS fill (their round trip runs under it) and are kept for the list pass below: a wave of a shard-sized callset has
    // no more than that, so its rows are read once.
    const int64_t m = (r1 - r0 + n_waves - 1) / n_waves;
    const int64_t w0 = r0 + (int64_t)wave * m, w1 = min(w0 + m, r1);
    constexpr int U = 8;
    // The lane's class in each of the wave's first 32 groups of 64 rows, two bits per group (bit 0 substitution, bit 1
    // indel): the list pass below reads them back instead of the columns (round 3 re-read every group but the first
    // eight: two more dependent round trips of a 5 M-variant launch's prologue).
    uint32_t cls[2] = {0u, 0u};
    int rl[U], al[U];
    auto load_classes = [&](int64_t g, int (&x)[U], int (&y)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = g + 64 * u + lane;
            x[u] = y[u] = -1;
            if (r < w1) { x[u] = a.ref_len[r]; y[u] = a.alt_len[r]; }
        }
    };
    load_classes(w0, rl, al);                                    // (requested BEFORE the LDS fill: their round trip runs under it)
    // ---- a contig boundary inside the workgroup's rows (round 4).  The tile loops cut a tile where the contig changes
    // (tile_cut); with tiles laid on a 64-entry grid from the head of the list that costs the wave that holds the boundary one
    // EXTRA tile - a third of a wave's work in a shard-sized launch, whose slowest workgroups were exactly those.  So the grid
    // restarts at the boundary: entries before the first row of the second contig are tiled from the list's head, the others
    // from that row's place in the list; the one extra tile of the workgroup falls on its last wave, which has room.  The row is
    // found here by the whole workgroup: one probe per thread under the fill, one more round after the first barrier.  (Further
    // boundaries in the same workgroup - small contigs - are left to tile_cut.)
    const int rows_here = (int)(r1 - r0);
    const int probe_step = (int)(((unsigned)rows_here + blockDim.x - 1u) / blockDim.x);
    const int64_t probe_row = r0 + (int64_t)tid * probe_step;
    const uint32_t c_first = a.contig[r0], c_last = a.contig[r1 - 1];
    const uint32_t c_probe = a.contig[min(probe_row, r1 - 1)];
    const Lds5 L = lds5_fill(smem, v, has0, tid, blockDim.x, !WX);
    unsigned cs = 0, ci = 0;
    {
        int gi = 0;
        for (int64_t g = w0;; g += 64 * U, gi += U) {
            int rln[U], aln[U];
            const bool more = g + 64 * U < w1;
            if (more) load_classes(g + 64 * U, rln, aln);        // the next batch is in flight while this one is counted
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool snp = rl[u] >= 0 && rl[u] == al[u], ind = rl[u] >= 0 && rl[u] != al[u];
                cs += (unsigned)__popcll(__ballot(snp));
                ci += (unsigned)__popcll(__ballot(ind));
                const int k = gi + u;
                const uint32_t two = (snp ? 1u : 0u) | (ind ? 2u : 0u);
                if (k < 16) cls[0] |= two << (2 * k);
                else if (k < 32) cls[1] |= two << (2 * (k - 16));
            }
            if (!more) break;
#pragma unroll
            for (int u = 0; u < U; ++u) { rl[u] = rln[u]; al[u] = aln[u]; }
        }
    }
    // (no static __shared__ in this kernel: the dynamic LDS then starts at address 0, the forest's p1 table with it, and the
    // walk's final read needs no base added to its payload - ugvc_walk.hpp: walk6; the per-wave counts live behind gtab)
    unsigned (*wcnt)[kK2Threads / 64] = reinterpret_cast<unsigned (*)[kK2Threads / 64]>(smem + (L.gtab_b - lds_addr(smem)) + 128);
    unsigned (*bnd)[kK2Threads / 64] = reinterpret_cast<unsigned (*)[kK2Threads / 64]>(smem + (L.gtab_b - lds_addr(smem)) + 256);   // [3][16]
    const bool has_b = rfl((int)c_first) != rfl((int)c_last) && probe_step <= 64 && m <= 2048;   // (m: the classes of a wave's rows are kept for 32 groups)
    if (has_b) {
        const unsigned long long fl = __ballot(probe_row >= r1 || c_probe != c_first);
        if (lane == 0) bnd[0][wave] = fl ? (unsigned)(wave * 64 + __builtin_ctzll(fl)) : ~0u;
    }
    if (lane == 0) { wcnt[0][wave] = cs; wcnt[1][wave] = ci; }
    if (tid == 0) lds_st32(L.gtab_b + 512u, 0);                  // (no second-contig state published yet: brk_publish)
    __syncthreads();
    // the first row of the second contig: the probe found it to one stride; one more load per lane closes it
    int64_t b_row = r1;
    uint32_t c_fine = 0;
    if (has_b) {
        unsigned F = ~0u;
        for (int w = 0; w < n_waves; ++w) F = min(F, bnd[0][w]);
        F = (unsigned)rfl((int)F);                                // (>= 1: thread 0 probes row r0 itself)
        const int64_t lo_row = r0 + (int64_t)(F - 1u) * probe_step + 1;
        c_fine = a.contig[min(lo_row + lane, r1 - 1)];           // (consumed after the list pass: in flight under it)
        b_row = lo_row;                                          // (+ the first lane whose contig differs: below)
    }
    unsigned ps = 0, pi = 0, ns_l = 0, ni_l = 0;
    for (int w = 0; w < n_waves; ++w) {
        const unsigned x = wcnt[0][w], y = wcnt[1][w];
        ps += w < wave ? x : 0u;
        pi += w < wave ? y : 0u;
        ns_l += x;
        ni_l += y;
    }
    uint32_t* __restrict__ ls = v.snp_idx + (size_t)blockIdx.x * v.list_stride;
    uint32_t* __restrict__ li = v.indel_idx + (size_t)blockIdx.x * v.list_stride;
    {
        const unsigned long long below = (1ull << lane) - 1;
        int gi = 0;
        for (int64_t g = w0; g < w1; g += 64, ++gi) {
            bool snp, ind;
            if (gi < 32) {
                const uint32_t two = (cls[gi >> 4] >> (2 * (gi & 15))) & 3u;
                snp = (two & 1u) != 0;
                ind = (two & 2u) != 0;
            } else {                                             // (a wave with more than 2048 rows: a launch of > 8 M variants)
                const int64_t r = g + lane;
                int x = -1, y = -1;
                if (r < w1) { x = a.ref_len[r]; y = a.alt_len[r]; }
                snp = x >= 0 && x == y;
                ind = x >= 0 && x != y;
            }
            const unsigned long long ms = __ballot(snp), mi = __ballot(ind);
            const uint32_t r = (uint32_t)(g + lane);
            if (snp) ls[ps + (unsigned)__popcll(ms & below)] = r;
            if (ind) li[pi + (unsigned)__popcll(mi & below)] = r;
            ps += (unsigned)__popcll(ms);
            pi += (unsigned)__popcll(mi);
        }
        if (wave == n_waves - 1) {                               // padding of the last tile of either list
            ls[ns_l + lane] = ~0u;
            li[ni_l + lane] = ~0u;
        }
    }
    if (has_b) {
        // this wave's substitutions / indels in rows before the boundary, from the kept class bits (the fine probe's round trip
        // ran under the list pass)
        const unsigned long long fl = __ballot(c_fine != c_first);    // (set at some lane < probe_step: the stride's last row differs)
        b_row += fl ? (int64_t)__builtin_ctzll(fl) : 0;
        unsigned sb_w = 0, ib_w = 0;
        int gi = 0;
        for (int64_t g = w0; g < w1 && g < b_row; g += 64, ++gi) {
            const uint32_t two = (cls[gi >> 4] >> (2 * (gi & 15))) & 3u;
            const int64_t nb = b_row - g;                        // rows of this group before the boundary (> 0)
            const unsigned long long mb = nb >= 64 ? ~0ull : (1ull << nb) - 1ull;
            sb_w += (unsigned)__popcll(__ballot((two & 1u) != 0) & mb);
            ib_w += (unsigned)__popcll(__ballot((two & 2u) != 0) & mb);
        }
        if (lane == 0) { bnd[1][wave] = sb_w; bnd[2][wave] = ib_w; }
    }
    __threadfence_block();
    __syncthreads();
    // places of the boundary in the two lists (entries before it), or the list's length without one
    unsigned b_s = ns_l, b_i = ni_l;
    if (has_b) {
        b_s = b_i = 0;
        for (int w = 0; w < n_waves; ++w) { b_s += bnd[1][w]; b_i += bnd[2][w]; }
        b_s = (unsigned)rfl((int)b_s);                           // (read from LDS: uniform, but in vector registers until named so)
        b_i = (unsigned)rfl((int)b_i);
    }
    Scratch sc;
    sc.eyt_b = L.eyt_b; sc.thr_b = L.thr_b; sc.gcr_b = L.gcr_b; sc.css_b = L.css_b; sc.gtab_b = L.gtab_b;
    const int hslot = ((lane & 31) << 1) | (lane >> 5);
    // tile T of a list: entries from 64 T before the boundary's tiles, from b + 64 (T - Tb) behind them
    // (32-bit: a workgroup's lists hold fewer than 2^31 entries)
    const int tb_s = (int)((b_s + 63u) >> 6), tb_i = (int)((b_i + 63u) >> 6);
    const int nst = tb_s + (int)((ns_l - b_s + 63u) >> 6);
    const int nit = (a.ablate & 262144) ? 0 : tb_i + (int)((ni_l - b_i + 63u) >> 6);
    auto tile_off = [](int T, int tb, unsigned b) { return T < tb ? 64 * T : (int)b + 64 * (T - tb); };
    // Wave roles.  The LDS layout gives the last `n_indel_waves` waves the larger (window-row) scratch; how many of
    // them actually work on indel tiles follows the class mix of the workgroup's rows (an SNV-only callset has none:
    // every wave runs the SNP pipeline).  A wave fetches its next tile's row indices, columns and table slices while
    // it works on the current one.
    const int n_big = v.n_indel_waves, n_small = n_waves - n_big;
    sc.base = L.scratch_b + (uint32_t)(wave < n_small ? wave * v.scratch_bytes : n_small * v.scratch_bytes + (wave - n_small) * v.scratch_indel);
    // (an indel tile's cost relative to an SNP tile's, in 1/256: V5Args::indel_w - 0.85 for the scoring pass, ~3 when nothing is walked)
    const int64_t wi = (int64_t)nit * v.indel_w, ws = (int64_t)nst * 256;
    int n_iw = nit > 0 ? (int)((n_waves * wi + (ws + wi) - 1) / (ws + wi)) : 0;
    n_iw = n_iw < n_big ? n_iw : n_big;
    if (nst == 0) n_iw = n_big;
    const int n_sw = n_waves - n_iw;
    const uint32_t planes_lane_b = sc.base + 2u * (uint32_t)hslot;
    const bool joins_on = !(a.ablate & 524288);
    if (wave >= n_sw) {
        const int q = (nit + n_iw - 1) / n_iw;
        const int t0 = (wave - n_sw) * q, t1 = min(t0 + q, nit);
        if (t0 >= t1) { if (blockIdx.x == 0 && wave == 0 && lane == 0) pass_clock_end(v); return; }
        const uint64_t wclk_first = v.wave_clk ? __builtin_readcyclecounter() : 0;
        PhaseClk pc{};
#ifdef UGVC_PHASE_CLOCK
        pc.last = __builtin_readcyclecounter();
        const uint64_t t_begin = pc.last;
#endif
        Brk<NT> bk{};
        bk.c = -1;
        IndelPre<NT> pre{};
        // this wave's entries of the workgroup's indel list; a tile = up to 64 of them from an offset (tile_cut)
        const int e0 = rfl(tile_off(t0, tb_i, b_i)), e1 = rfl(min(tile_off(t1, tb_i, b_i), (int)ni_l));
        auto ids_at = [&](int off, uint32_t& i, bool& live) {
            const uint32_t id = li[off + lane];                // (the list is padded by 64 entries: off < e1 <= ni_l)
            live = id != ~0u && off + lane < e1;
            const uint32_t id0 = (uint32_t)rfl((int)id);       // (outside the select: a ternary would read the first PADDING lane)
            i = live ? id : id0;
        };
        uint32_t i, i_n = 0;
        bool live, live_n = false;
        int off = e0;
        ids_at(off, i, live);
        IndelCols cols = load_indel_cols(a, i);
        if (off + 64 < e1) ids_at(off + 64, i_n, live_n);
        unsigned n_done = 0;
        const UGVC_CONST V5Args* const vk = (const UGVC_CONST V5Args*)__builtin_amdgcn_kernarg_segment_ptr();
        for (;;) {
            const UGVC_CONST V5Args* vq = vk;                   // (launch arguments re-read per tile, as in the SNP loop below)
            asm volatile("" : "+s"(vq));
            const V5Args& vt = *(const V5Args*)vq;
            const int take = tile_cut(live, i, cols);
            const int off_n = off + take;
            if (take != 64 && off_n < e1) ids_at(off_n, i_n, live_n);   // (the ids fetched ahead were those of off + 64)
            const bool more = off_n < e1;
            // two tiles ahead: row indices; one tile ahead: columns (both in flight during this tile)
            uint32_t i_n2 = 0;
            bool live_n2 = false;
            IndelCols cols_n = cols;
            if (more) cols_n = load_indel_cols(vt.f, i_n);
            const int off_n2 = off_n + 64;
            uint32_t id_n2 = ~0u;
            if (more && off_n2 < e1) id_n2 = li[off_n2 + lane];
            CLK(pc, 11);                                        // (tile bookkeeping: cut, next columns and row indices requested)
            auto touch_next = [&]() {
                asm volatile("" ::"v"(cols_n.c), "v"(cols_n.pos), "v"(cols_n.rl), "v"(cols_n.al), "v"(cols_n.ro), "v"(cols_n.ao), "v"(cols_n.qual),
                             "v"(cols_n.sor), "v"(cols_n.dp), "v"(cols_n.adr), "v"(cols_n.ada), "v"(cols_n.gq), "v"(id_n2));
            };
            featurize_indel_tile<NTRK, WX>(vt, sc, (uint32_t)(blockIdx.x * 7 + (off >> 6)), 64 - (int)(off & 63), lane, i, live, cols, bk, pre, pc, touch_next);
            ++n_done;
            if (!more) break;
            if (joins_on && bk.c >= 0) issue_indel_slices<NTRK>(vt, bk, lane, pre);
            {
                live_n2 = id_n2 != ~0u && off_n2 + lane < e1;
                const uint32_t id0 = (uint32_t)rfl((int)id_n2);
                i_n2 = live_n2 ? id_n2 : id0;
            }
            off = off_n;
            cols = cols_n; i = i_n; live = live_n; i_n = i_n2; live_n = live_n2;
            __builtin_amdgcn_wave_barrier();
        }
        if (v.wave_clk && lane == 0) {
            unsigned long long* w = v.wave_clk + 4 + ((size_t)blockIdx.x * 16 + wave) * 4;
            w[0] = wclk_entry; w[1] = wclk_first; w[2] = __builtin_readcyclecounter(); w[3] = (unsigned long long)n_done | ((unsigned long long)n_done << 32);
            if (blockIdx.x == 0 && wave == 0) pass_clock_end(v);   // (an indel-only first workgroup: wave 0 works on indel tiles)
        }
#ifdef UGVC_PHASE_CLOCK
        if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 133))
            printf("iclk b%d w%d tiles %d total %llu | bookkeeping %llu load+window %llu hmer %llu motif+gc %llu staging %llu narrow %llu wide %llu ranks %llu record %llu\n",
                   (int)blockIdx.x, wave, (int)n_done, (unsigned long long)(pc.last - t_begin), (unsigned long long)pc.acc[11], (unsigned long long)pc.acc[0],
                   (unsigned long long)pc.acc[8], (unsigned long long)pc.acc[1], (unsigned long long)pc.acc[9], (unsigned long long)pc.acc[10],
                   (unsigned long long)pc.acc[2], (unsigned long long)pc.acc[6], (unsigned long long)pc.acc[3]);
#endif
        return;
    }
    // (equal shares, the last wave takes what is left.  Shares that differ by at most one tile - the larger ones to the oldest
    // waves - were measured: no faster, profiles/r04_even_shares_ab.txt; the waves of a SIMD share its issue slots, so a wave
    // that finishes early leaves them to the others: what counts is the workgroup's total work, not the spread of its waves' ends)
    const int q = (nst + n_sw - 1) / n_sw;
    int t0 = wave * q, t1 = min(t0 + q, nst);
    if (v.snp_cum[16] != 0) {                                   // (profiling: weighted shares)
        const int tot = v.snp_cum[n_sw];
        t0 = (int)((int64_t)nst * v.snp_cum[wave] / tot);
        t1 = (int)((int64_t)nst * v.snp_cum[wave + 1] / tot);
    }
    if (has_b && wave == n_sw - 1 && b_row < r1) {               // (brk_publish: the second contig's state, once for the workgroup)
        Brk<NT> b2{};
        brk_refresh<NT>(a, b2, rfl((int)a.contig[b_row]), rfl(a.pos[b_row]), lane);
        brk_publish<NT>(L.gtab_b + 512u, b2, lane);
    }
    if (t0 >= t1) { if (blockIdx.x == 0 && wave == 0 && lane == 0) pass_clock_end(v); return; }
    const uint64_t wclk_first = v.wave_clk ? __builtin_readcyclecounter() : 0;
    // this wave's entries of the workgroup's SNP list; a tile = up to 64 of them from an offset (tile_cut)
    const int e0 = rfl(tile_off(t0, tb_s, b_s)), e1 = rfl(min(tile_off(t1, tb_s, b_s), (int)ns_l));
    auto ids_at = [&](int off, uint32_t& i, bool& live) {
        const uint32_t id = ls[off + lane];                    // (the list is padded by 64 entries: off < e1 <= ns_l)
        live = id != ~0u && off + lane < e1;
        const uint32_t id0 = (uint32_t)rfl((int)id);
        i = live ? id : id0;
    };
    uint32_t i;
    bool live;
    int off = e0;
    ids_at(off, i, live);
    SnpCols cols = load_snp_cols(a, i);
    Brk<NT> bk{};
    bk.c = -1;
    SlicePre<NT> pre{};
    PhaseClk pc{};
    unsigned n_done = 0;
#ifdef UGVC_PHASE_CLOCK
    pc.last = __builtin_readcyclecounter();
    const uint64_t t_begin = pc.last;
#endif
    // A featurize phase is a short instruction stream between long memory waits; the walk is a long stream that waits
    // on LDS.  (Raised priority for the featurize phase - s_setprio - made no measurable difference: variant bit 29.)
    const bool prio = !(a.ablate & (1 << 29));
    const UGVC_CONST V5Args* const vk = (const UGVC_CONST V5Args*)__builtin_amdgcn_kernarg_segment_ptr();
    for (;;) {
        // The launch arguments are read afresh in every tile (the pointer passes through an empty asm): hoisted out of
        // the loop they would sit in ~100 SGPRs, spill to VGPR lanes and come back as v_readlane - vector-ALU work, which
        // is what this kernel is short of; scalar loads from the constant cache are not.
        const UGVC_CONST V5Args* vq = vk;
        asm volatile("" : "+s"(vq));
        const V5Args& vt = *(const V5Args*)vq;
        const FilterArgs& at = vt.f;
        const int off_n = off + tile_cut(live, i, cols);    // (the tile ends where the contig changes)
        CLK(pc, 12);                                            // (the wait for this tile's columns, requested before the previous walk)
        const bool more = off_n < e1;
        uint32_t id_n = 0, i_n = 0;
        bool live_n = false;
        if (prio) __builtin_amdgcn_s_setprio(2);
        if (more) id_n = ls[off_n + lane];                       // consumed after the joins
        featurize_snp_tile<NTRK, WX>(vt, sc, lane, i, live, has0, cols, bk, pre, pc);
        SnpCols cols_n = cols;
        if (more) {
            live_n = id_n != ~0u && off_n + lane < e1;
            const uint32_t id0 = (uint32_t)rfl((int)id_n);
            i_n = live_n ? id_n : id0;
            cols_n = load_snp_cols(at, i_n);                    // in flight during the walk
            if (joins_on && bk.c >= 0) issue_slices<NT>(vt, bk, lane, pre);   // likewise: from where this tile's last variant ended
        }
        CLK(pc, 4);
        if (prio) __builtin_amdgcn_s_setprio(0);
        if (has0) {
            float score = 0.f;
            uint8_t filt = UGVC_FILTER_PASS;
            if (!(at.ablate & 131072)) walk_forest<NTW>(vt.pg[0], L.hi_b, L.last_b, L.p1_b, planes_lane_b, score, filt);
            if (live) {
                stg32(at.score, i, score);
                stg32(at.filter, i, filt);
            }
        } else if (live && !WX) {                              // no model for substitutions: score 0, PASS
            stg32(at.score, i, 0.f);
            stg32(at.filter, i, (uint8_t)UGVC_FILTER_PASS);
        }
        ++n_done;
        if (!more) break;
        off = off_n;
        cols = cols_n; i = i_n; live = live_n;
        __builtin_amdgcn_wave_barrier();
        CLK(pc, 5);
    }
    if (v.wave_clk && lane == 0) {
        unsigned long long* w = v.wave_clk + 4 + ((size_t)blockIdx.x * 16 + wave) * 4;
        w[0] = wclk_entry; w[1] = wclk_first; w[2] = __builtin_readcyclecounter(); w[3] = (unsigned long long)n_done;
        if (blockIdx.x == 0 && wave == 0) pass_clock_end(v);
    }
#ifdef UGVC_PHASE_CLOCK
    if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 133))
        printf("clk b%d w%d tiles %d total %llu | stage %llu window %llu joins %llu codes %llu cols_n %llu walk %llu\n", (int)blockIdx.x, wave, (int)n_done,
               (unsigned long long)(pc.last - t_begin), (unsigned long long)pc.acc[0], (unsigned long long)pc.acc[1], (unsigned long long)pc.acc[2],
               (unsigned long long)pc.acc[3], (unsigned long long)pc.acc[4], (unsigned long long)pc.acc[5]);
    if (lane == 0 && blockIdx.x == 0 && wave == 0)
        printf("  cut %llu issue %llu eyt %llu | prologue (fill, class lists) %llu cycles before the first tile\n", (unsigned long long)pc.acc[12],
               (unsigned long long)pc.acc[6], (unsigned long long)pc.acc[7], (unsigned long long)(t_begin - k_begin));
#endif
}

// ---- K2: the indel groups' forests over the raw-code records ---------------------------------------------
__global__ __launch_bounds__(kK2Threads) void forest5_kernel(const V5Args v) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef UGVC_PHASE_CLOCK
    const uint64_t f_begin = __builtin_readcyclecounter();
    uint64_t f_ready = 0, f_walk = 0, f_last = 0;
    int f_chunks = 0;
#endif
    // (no static __shared__: see fused5_kernel; the launch adds 1088 bytes behind the forest and the code planes for these)
    if (lds_addr(smem) != 0u) __builtin_trap();                  // (walk_forest: payloads are absolute LDS addresses)
    unsigned* const shard_off = reinterpret_cast<unsigned*>(smem + v.forest_lds_tail);       // [kShards + 1]
    unsigned* const totals = shard_off + kShards + 4;                                         // [UGVC_N_GROUPS]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int n_waves = blockDim.x >> 6;
    if (tid < UGVC_N_GROUPS) totals[tid] = 0;
    __syncthreads();
    for (int q = tid; q < UGVC_N_GROUPS * kShards; q += blockDim.x) {
        const unsigned cshard = v.counters[q * kCounterStride];
        if (cshard && q >= kShards) atomicAdd(&totals[q / kShards], cshard);
        if (blockIdx.x == 0) v.counters_next[q * kCounterStride] = 0;      // the NEXT pass's record lists start empty (two sets, alternating)
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
    fill_forest_lds<6>(smem, reinterpret_cast<const uint4*>(pg.p1), b_p1 / 16, reinterpret_cast<const uint4*>(pg.hi4), b_hi / 16,
                       reinterpret_cast<const uint4*>(pg.last4), b_last / 16, tid, blockDim.x);
    __syncthreads();
    const uint32_t p1_b = lds_addr(smem), hi_b = lds_addr(smem + b_p1), last_b = lds_addr(smem + b_p1 + b_hi);
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
#ifdef UGVC_PHASE_CLOCK
    f_ready = __builtin_readcyclecounter();
#endif
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
#ifdef UGVC_PHASE_CLOCK
        f_last = __builtin_readcyclecounter();
#endif
        walk_forest<8>(pg, hi_b, last_b, p1_b, planes_lane_b, score, filt);
#ifdef UGVC_PHASE_CLOCK
        f_walk += __builtin_readcyclecounter() - f_last;
        ++f_chunks;
#endif
        if (live) {
            v.f.score[q2.w] = score;
            v.f.filter[q2.w] = filt;
        }
    }
#ifdef UGVC_PHASE_CLOCK
    if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 200) && (wave == 0 || wave == 15))
        printf("fclk b%d w%d g%d: ready after %llu cycles, %d chunks, walk %llu, total %llu\n", (int)blockIdx.x, wave, g,
               (unsigned long long)(f_ready - f_begin), f_chunks, (unsigned long long)f_walk, (unsigned long long)(__builtin_readcyclecounter() - f_begin));
#endif
}

static size_t k5_forest_lds(const PackedGroupView& pg, int n_waves) {
    const size_t n_hi = ((size_t)pg.T << pg.D) / 2;
    return ((n_hi * 4 + 15) & ~(size_t)15) + ((n_hi * 8 + 15) & ~(size_t)15) + (((size_t)pg.n_pairs * 8 + 15) & ~(size_t)15) +
           (size_t)n_waves * kMaxFeatures * 128;
}

// LDS budget of the fused kernel for this configuration: 16, 12 or 8 waves of scratch beside the SNP forest
int v5_fused_waves(const V5Args& v) {
    for (int w : {16, 12, 8}) {
        if (lds5_bytes(v, w) <= kLds5Limit) return w;
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

static K5 fused5_wx_for(int n_tracks) {
    switch (n_tracks) {
        case 0: return fused5_kernel<0, 16, true>;
        case 1: return fused5_kernel<1, 16, true>;
        case 2: return fused5_kernel<2, 16, true>;
        case 3: return fused5_kernel<3, 16, true>;
        case 4: return fused5_kernel<4, 16, true>;
        default: return fused5_kernel<5, 16, true>;
    }
}

// The N x F feature matrix (a.X, a.group) from the fused kernel's featurize waves; no model needed.
int launch_feature_matrix_v5(ugvc_ctx* ctx, const FilterArgs& a) {
    if (a.n == 0) return 0;
    V5Args v;
    if (v5_fill_args(ctx, v, a, false)) return -1;
    static bool attr_set[64] = {};
    if (!attr_set[ctx->device & 63]) {
        for (int t = 0; t <= UGVC_MAX_TRACKS; ++t)
            UGVC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fused5_wx_for(t)), hipFuncAttributeMaxDynamicSharedMemorySize, kLds5Limit));
        attr_set[ctx->device & 63] = true;
    }
    for (int g = 0; g < UGVC_N_GROUPS; ++g) v.pg[g].ok = 0;      // (nothing is ranked or walked: the LDS holds scratch only)
    v.run_forest = 0;
    v.n_waves = 16;
    v.n_indel_waves = 8;
    v.indel_w = 768;                                              // without the walk an indel tile costs about three SNP tiles
    // every wave's LDS scratch also holds a tile's rows on their way out (store_feature_rows_tile): 64 F floats + 64 indices
    const int rows_lds = (64 * kMaxFeatures * 4 + 256 + 63) & ~63;
    v.scratch_bytes = std::max(v.scratch_bytes, rows_lds);
    v.scratch_indel = std::max(v.scratch_indel, rows_lds);
    if (const char* e = getenv("UGVC_FM_INDEL_W")) v.indel_w = std::max(1, (int)(atof(e) * 256.0));      // (profiling)
    if (lds5_bytes(v, v.n_waves) > (size_t)kLds5Limit) return fail("internal: feature-matrix scratch does not fit LDS");
    const unsigned n_wg = (unsigned)((a.n + v.rows_wg - 1) / v.rows_wg);
    UGVC_LAUNCH(fused5_wx_for(a.n_tracks), dim3(n_wg), dim3(v.n_waves * 64), lds5_bytes(v, v.n_waves), ctx->stream, v);
    UGVC_HIP(hipGetLastError());
    return 0;
}

// The scoring kernels' function attributes (per device, once).  Also the first use of the library's code object on the device:
// ugvc_reserve calls it from a helper thread so that a tool's first pass does not pay the load (~10 ms).
int v5_warm(ugvc_ctx* ctx) {
    static std::atomic<bool> attr_set[64] = {};
    if (!attr_set[ctx->device & 63].load()) {
        for (K5 f : {fused5_for(0), fused5_for(1), fused5_for(2), fused5_for(3), fused5_for(4), fused5_for(5), fused5_for(3, true), (K5)forest5_kernel,
                     (K5)fused5_kernel<3, 16>})
            UGVC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, kLds5Limit));
        attr_set[ctx->device & 63].store(true);
    }
    return 0;
}

int launch_filter_v5(ugvc_ctx* ctx, const FilterArgs& a) {
    if (a.n == 0) return 0;
    V5Args v;
    if (v5_fill_args(ctx, v, a)) return -1;
    if (v5_warm(ctx)) return -1;
    static const bool dbg = getenv("UGVC_DEBUG_SYNC") != nullptr;     // (UGVC_LAUNCH names every launch and waits for it)
    // UGVC_WAVE_CLK=<file>: every wave of the fused kernel leaves its entry / first-tile / end clocks and its tile counts; the
    // buffer of the LAST pass is written to the file (tools/wave_clk.py reads it).  Waits for every pass: profiling only.
    static const char* wclk_path = getenv("UGVC_WAVE_CLK");
    DeviceBuf& wclk_buf = ctx->wclk_buf;                              // (per context, freed with it: ADVICE r5)
    const size_t wclk_bytes = (size_t)((a.n + v.rows_wg - 1) / v.rows_wg) * 16 * 4 * 8 + 32;     // (+ the pass clock words)
    const bool want_clk = wclk_path || ctx->clk_probe;
    if (want_clk) {
        if (ensure(wclk_buf, wclk_bytes)) return -1;
        UGVC_HIP(hipMemsetAsync(wclk_buf.p, 0, wclk_bytes, ctx->stream));
        v.wave_clk = wclk_buf.as<unsigned long long>();
    }
    const size_t lds_f = lds5_bytes(v, v.n_waves);
    if (dbg) fprintf(stderr, "[ugvc v5] fused5: %d waves (%d indel), %zu B of LDS\n", v.n_waves, v.n_indel_waves, lds_f);
    const unsigned n_wg = (unsigned)((a.n + v.rows_wg - 1) / v.rows_wg);
    UGVC_LAUNCH(fused5_for(a.n_tracks, (a.ablate & (1 << 28)) != 0), dim3(n_wg), dim3(v.n_waves * 64), lds_f, ctx->stream, v);
    if (want_clk) {
        std::vector<unsigned long long> h(wclk_bytes / 8);
        UGVC_HIP(copy_out(ctx, h.data(), wclk_buf.p, wclk_bytes));
        UGVC_HIP(hipStreamSynchronize(ctx->stream));
        const unsigned long long* tail = h.data();
        // {100 MHz ticks, shader-clock ticks} between the entry and the end of workgroup 0's first wave
        ctx->clk_rt = tail[1] > tail[0] ? tail[1] - tail[0] : 0;
        ctx->clk_sh = tail[3] > tail[2] ? tail[3] - tail[2] : 0;
        if (wclk_path) {
            static int wclk_pass = 0;                              // (the previous pass stays beside it as <file>.prev: are the slow workgroups the same ones?)
            if (wclk_pass++ > 0) (void)rename(wclk_path, (std::string(wclk_path) + ".prev").c_str());
            if (FILE* f = fopen(wclk_path, "wb")) { fwrite(h.data() + 4, 1, wclk_bytes - 32, f); fclose(f); }
        }
    }
    if (v.run_forest) {
        int n_waves = 0;
        size_t lds = 0;
        for (int w : {16, 12, 8, 4}) {
            size_t need = 0;
            for (int g = 1; g < UGVC_N_GROUPS; ++g)
                if (v.pg[g].ok) need = std::max(need, k5_forest_lds(v.pg[g], w));
            if (need + 1088 <= 158 * 1024) { n_waves = w; lds = need; break; }
        }
        if (n_waves == 0) return fail("internal: packed forest does not fit LDS");
        v.forest_lds_tail = (int)lds;
        UGVC_LAUNCH(forest5_kernel, dim3((unsigned)ctx->n_cus), dim3(n_waves * 64), lds + 1088, ctx->stream, v);
    }
    UGVC_HIP(hipGetLastError());
    return 0;
}

}  // namespace ugvc
