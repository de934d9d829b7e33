// v4 featurize kernel (gfx950): same outputs as featurize3_kernel (kernels_v3.hip), about half its
// vector instructions per variant.  K1 of the scoring pass is VALU-issue and latency bound (phase clocks:
// tools/phase3.py), so v4 removes instructions and per-lane state rather than bytes:
//   * tiles never span contigs (host tile table, <= 256 consecutive variants of ONE contig): contig
//     bounds, CSR ranges of the side tables and the reference base address are wave-uniform scalars,
//     columns are addressed as scalar base + lane offset, no per-lane CSR gathers, no 64-bit math;
//   * staged side-table slices carry sentinels (INT_MIN before the contig's range, INT_MAX behind it,
//     power-of-two padded), so a descent step is add / ds_read / compare / select (3 VALU, was 5) and the
//     interval tests need no index arithmetic or range checks; blacklist keys are staged as their
//     32-bit position halves (one contig per tile) and searched like the other tables;
//   * reference-window features work on packed bytes: the 12-byte homopolymer look-ahead is three
//     v_alignbyte + xor + find-first-set, motifs are v_dot4 dot products, GC content a popcount and a
//     table look-up instead of an f64 division, the allele tail one 8-byte load instead of eight
//     byte gathers.
// Semantics are those of the oracle (oracle/oracle.py); tests/test_gpu_parity.py runs v4, v3, v2, v1.
#include <limits.h>

#include "ugvc_v2.hpp"

namespace ugvc {

namespace {

constexpr int kWinDw = 12;            // 48-byte reference window per variant
constexpr int kWinStride = 13;        // dwords per lane row (odd: conflict-free column access)
constexpr int kWinBytes = kWinDw * 4;
constexpr int kPool4 = 2560;          // dwords of LDS for the staged side-table slices of one tile
constexpr int kSeg4 = 2 * (kJoin3 - 1) + 1;   // staged segments: starts + ends of 6 tables, blacklist positions

#define UGVC_LDS __attribute__((address_space(3)))
__device__ __forceinline__ int rfl(int x) { return __builtin_amdgcn_readfirstlane(x); }
template <class T> __device__ __forceinline__ uint32_t lds_addr(T* p) { return (uint32_t)(uintptr_t)(UGVC_LDS T*)p; }
__device__ __forceinline__ int lds_i32(uint32_t a) { return *(UGVC_LDS const int32_t*)(uintptr_t)a; }
__device__ __forceinline__ uint32_t lds_u32(uint32_t a) { return *(UGVC_LDS const uint32_t*)(uintptr_t)a; }
__device__ __forceinline__ float lds_f32(uint32_t a) { return *(UGVC_LDS const float*)(uintptr_t)a; }

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

// 16-ary lower bound (see kernels_v3.hip): 15 independent pivot loads per round
template <class T>
__device__ __forceinline__ int lb_wide_g(const T* __restrict__ a, int lo, int hi, T key) {
    while (hi - lo > 16) {
        const int step = (hi - lo) >> 4;
        T x[15];
#pragma unroll
        for (int k = 0; k < 15; ++k) x[k] = a[lo + (k + 1) * step];
        int c = 0;
#pragma unroll
        for (int k = 0; k < 15; ++k) c += x[k] < key ? 1 : 0;
        const int nlo = c == 0 ? lo : lo + c * step + 1;
        hi = c == 15 ? hi : lo + (c + 1) * step;
        lo = nlo;
    }
    const int n = hi - lo;
    int c = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const T x = k < n ? a[lo + k] : key;
        c += (k < n && x < key) ? 1 : 0;
    }
    return lo + c;
}

__device__ __forceinline__ const TrackView& table_view(const FilterArgs& f, int t) { return t == 0 ? f.runs : f.tracks[t - 1]; }
__device__ __forceinline__ bool table_present(const FilterArgs& f, int t) { return t == 0 ? f.has_runs != 0 : (t - 1) < f.n_tracks; }

// bytes of {hi, lo} starting at byte `sh` (0..3) of lo
__device__ __forceinline__ uint32_t bytes_at(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }
// bit 7 of every byte that is non-zero (exact for any byte value)
__device__ __forceinline__ uint32_t nonzero_bytes(uint32_t x) { return (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u; }
// number of leading (lowest-address) zero bytes of x, 0..4
__device__ __forceinline__ uint32_t lead_zero_bytes(uint32_t x) {
    const uint32_t y = nonzero_bytes(x);
    return y ? (uint32_t)__builtin_ctz(y) >> 3 : 4u;
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

struct Plan4 {                 // staging plan of one tile (LDS), one column per searched table
    int p0[8];                 // LDS byte address of the element BEFORE the tile's search range (starts slice)
    int dE[8];                 // byte distance from a starts element to the ends element of the same interval
    int bits[8];               // descent depth: 2^bits > search range
    int segSrc[16];            // per staged segment: first global index copied
    int segDst[16];            // pool offset (dwords)
    int segCnt[16];            // elements
    int segLo[16], segHi[16];  // the contig's range of the source table: below -> INT_MIN, at/above -> INT_MAX
    int lo[8], hi[8];          // search range (global indices) for the unstaged path
    int plo[8], phi[8];
    int maxbits;
    int staged;
};

}  // namespace

// ---- K0: brackets3[b][a] = first index of searched array a (a < 6: starts of table a inside the tile's
// contig; a == 6: blacklist keys) that is >= the first variant of tile b; row n_tiles holds the lengths.
__global__ void bracket4_kernel(const V2Args v) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nb = v.n_blocks;
    if (gid < UGVC_N_GROUPS * kShards) v.counters[gid * kCounterStride] = 0;
    const int b = (int)(gid >> 3), a = (int)(gid & 7);
    if (b > nb || a >= kJoin3) return;
    const FilterArgs& f = v.f;
    int out = 0;
    if (a == kJoin3 - 1) {
        if (f.n_bl > 0) {
            if (b == nb) out = (int)f.n_bl;
            else {
                const int64_t i = v.tiles4[b].x;
                out = lb_wide_g<uint64_t>(f.bl, 0, (int)f.n_bl, ((uint64_t)f.contig[i] << 32) | (uint32_t)f.pos[i]);
            }
        }
    } else if (table_present(f, a)) {
        const TrackView& tv = table_view(f, a);
        if (b == nb) out = tv.ptr[f.n_contigs];
        else {
            const int64_t i = v.tiles4[b].x;
            const int c = f.contig[i];
            out = lb_wide_g<int32_t>(tv.starts, tv.ptr[c], tv.ptr[c + 1], f.pos[i]);
        }
    }
    v.brackets3[gid] = out;
}

__global__ __launch_bounds__(kBlock, 4) void featurize4_kernel(const V2Args v) {
    __shared__ uint32_t win[kBlock * kWinStride];                  // 13 KB
    __shared__ int32_t pool[kPool4];                               // 10 KB
    __shared__ __attribute__((aligned(16))) float thr_lds[kThr3];   // 14 KB
    // per (group, feature): {table offset | kind << 30, table length, byte offset of the code's dword, bit offset}
    __shared__ uint4 desc_lds[UGVC_N_GROUPS * kMaxFeatures];
    __shared__ float gc_lut[11 * 11];                              // (f32)((f64)count / (f64)length)
    __shared__ uint8_t css_lds[256];
    __shared__ Plan4 plan;

    const FilterArgs& a = v.f;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int F = UGVC_N_BASE_FEATURES + a.n_tracks;
    const uint8_t* __restrict__ apool = a.alleles;
    const int okbits = (v.pg[0].ok ? 1 : 0) | (v.pg[1].ok ? 2 : 0) | (v.pg[2].ok ? 4 : 0);
    const int gbtbits = (v.pg[0].kind == UGVC_MODEL_GBT ? 1 : 0) | (v.pg[1].kind == UGVC_MODEL_GBT ? 2 : 0) |
                        (v.pg[2].kind == UGVC_MODEL_GBT ? 4 : 0);

    // ---- once per workgroup: model-side tables into LDS
    for (int k = tid; k < UGVC_N_GROUPS * kMaxFeatures; k += kBlock) {
        const uint2 d = v.desc3[k];
        desc_lds[k] = make_uint4(d.x, d.y & 0xFFFFu, ((d.y >> 16) & 3u) * 4u, (d.y >> 18) & 31u);
    }
    // (16-byte pieces: the threshold table is padded to a multiple of four floats on the host)
    for (int k = tid; k < (v.thr_lds_len + 3) / 4; k += kBlock)
        reinterpret_cast<float4*>(thr_lds)[k] = reinterpret_cast<const float4*>(v.thr)[k];
    if (tid < 121) {
        const int cnt = tid / 11, len = tid % 11;
        gc_lut[tid] = len > 0 ? (float)((double)cnt / (double)len) : 0.0f;
    }
    css_lds[tid] = v.css_lut[tid];
    __shared__ unsigned long long prof_lds[8];
    const bool prof_on = (a.ablate & 64) != 0 && v.prof != nullptr;
    unsigned long long prof_t = 0;
    if (prof_on && tid == 0) {
        for (int k = 0; k < 8; ++k) prof_lds[k] = 0;
        prof_t = clock64();
    }
    __syncthreads();
    const uint32_t pool_b = lds_addr(pool);

    for (int tile = blockIdx.x; tile < v.n_blocks; tile += gridDim.x) {
        // ---- tile descriptor (uniform): first variant, count, contig and its bounds
        const int2 td = v.tiles4[tile];
        const int t_first = rfl(td.x), t_cnt = rfl(td.y & 0xFFFF), c = rfl((td.y >> 16) & 0xFF);
        const int64_t clo = a.contig_off[c], chi = a.contig_off[c + 1];
        const uint32_t clen = (uint32_t)(chi - clo);
        const bool live = tid < t_cnt;
        const uint32_t li = live ? (uint32_t)tid : (uint32_t)(t_cnt - 1);   // idle lanes shadow the tile's last variant
        const uint32_t i = (uint32_t)t_first + li;

        // ---- staging plan: lanes 0..6 of wave 0, one searched table each
        if (tid < kJoin3) {
            const int t = tid;
            const int32_t* brow = v.brackets3 + (int64_t)tile * 8;
            const int lo = brow[t], hi_raw = brow[8 + t];
            int plo = 0, phi = 0;
            bool present = false;
            if (t == kJoin3 - 1) {
                present = a.n_bl > 0;
                if (present) { plo = v.bl_ptr[c]; phi = v.bl_ptr[c + 1]; }
            } else if (table_present(a, t)) {
                present = true;
                const TrackView& tv = table_view(a, t);
                plo = tv.ptr[c]; phi = tv.ptr[c + 1];
            }
            const int hi = hi_raw < phi ? hi_raw : phi;       // the next tile may lie in the next contig
            const int len = present && hi > lo ? hi - lo : 0;
            const int bits = len > 0 ? 32 - __builtin_clz((unsigned)len) : 0;
            const int halo = t == kJoin3 - 1 ? 0 : 2;
            const int L = lo - halo;                          // may be negative / below plo: sentinels
            const int cntS = present ? halo + (1 << bits) : 0;           // indices L .. lo + 2^bits - 1
            const int cntE = present && t < kJoin3 - 1 ? halo + len + 1 : 0;   // indices L .. hi
            // exclusive prefix over the 7 planning lanes: pool offsets
            int mine = cntS + cntE, incl = mine;
#pragma unroll
            for (int d = 1; d < 8; d <<= 1) {
                const int y = __shfl_up(incl, d);
                if (lane >= d) incl += y;
            }
            const int offS = incl - mine, offE = offS + cntS;
            plan.p0[t] = (int)pool_b + 4 * (offS + halo - 1);
            plan.dE[t] = 4 * (offE - offS);
            plan.bits[t] = bits;
            plan.lo[t] = lo; plan.hi[t] = hi; plan.plo[t] = plo; plan.phi[t] = phi;
            plan.segSrc[t] = L; plan.segDst[t] = offS; plan.segCnt[t] = cntS; plan.segLo[t] = plo; plan.segHi[t] = phi;
            if (t < kJoin3 - 1) {
                plan.segSrc[7 + t] = L; plan.segDst[7 + t] = offE; plan.segCnt[7 + t] = cntE;
                plan.segLo[7 + t] = plo; plan.segHi[7 + t] = phi;
            }
            int mb = bits;                                    // running maximum: the last planning lane holds the overall one
#pragma unroll
            for (int d = 1; d < 8; d <<= 1) {
                const int y = __shfl_up(mb, d);
                if (lane >= d) mb = y > mb ? y : mb;
            }
            if (t == kJoin3 - 1) {
                plan.maxbits = mb;
                plan.staged = incl <= kPool4;
            }
        }
        // ---- variant columns: scalar base of the tile + lane offset
        const int pos = (a.pos + t_first)[li];
        const int rl = (a.ref_len + t_first)[li], al = (a.alt_len + t_first)[li];
        const uint32_t ro = (a.ref_off + t_first)[li], ao = (a.alt_off + t_first)[li];
        const float qual = (a.qual + t_first)[li], sor = (a.sor + t_first)[li];
        const int dp = (a.dp + t_first)[li], adr = (a.ad_ref + t_first)[li], ada = (a.ad_alt + t_first)[li];
        const int gq = (a.gq + t_first)[li];

        // ---- classify_indel (lengths only), then the second batch of loads: reference window, allele bytes
        const bool indel = rl != al;
        const bool ins = rl < al;
        const int classify = !indel ? 0 : (ins ? 1 : 2);
        const int indel_length = ins ? al - rl : rl - al;
        const uint32_t p0 = (uint32_t)(pos - 1);              // 0-based offset inside the contig
        // window start: 16-byte aligned in the reference buffer, 6..21 bytes before the variant's first base;
        // relative to the contig it may start up to 21 bytes early (front padding / previous contig: blanked below)
        const int mis = (int)(clo & 15);
        const int wrel = (((int)p0 - 6 + mis) & ~15) - mis;
        const int o0 = (int)p0 - wrel;                        // 6..21
        uint32_t* wrow = win + tid * kWinStride;
        {
            const uint8_t* cbase = a.ref + clo - 32;          // uniform; lane offsets stay non-negative
            const uint4* src = reinterpret_cast<const uint4*>(cbase + (uint32_t)(wrel + 32));
            const uint4 x0 = src[0], x1 = src[1], x2 = src[2];
            wrow[0] = x0.x; wrow[1] = x0.y; wrow[2] = x0.z; wrow[3] = x0.w;
            wrow[4] = x1.x; wrow[5] = x1.y; wrow[6] = x1.z; wrow[7] = x1.w;
            wrow[8] = x2.x; wrow[9] = x2.y; wrow[10] = x2.z; wrow[11] = x2.w;
        }
        // allele bytes: substitutions need ref[0] and alt[0]; indels the tail of the longer allele, fetched as
        // one 8-byte load at its second base (the pool is padded by 16 bytes)
        const uint32_t lo_off = ins ? ao : ro;
        const int ln = ins ? al : rl;
        typedef uint64_t __attribute__((aligned(1))) u64_unaligned;
        const uint64_t tail = *reinterpret_cast<const u64_unaligned*>(apool + (indel ? lo_off + 1 : ro));
        const uint32_t alt0 = apool[ao];
        __syncthreads();                                      // plan visible

        // ---- stage the slices this tile can touch: wave w copies segments w, w+4, w+8, w+12
        const int abl = a.ablate;       // profiling only: 2 joins, 4 quantise, 8 window features, 16 append
        const int staged = rfl(plan.staged);
        if (prof_on && tid == 0) { const unsigned long long now_ = clock64(); prof_lds[1] += now_ - prof_t; prof_t = now_; }
        if (staged && !(abl & 2)) {
            // every load of the wave's (up to four) segments is issued before the first pool write: one HBM round
            // trip instead of one per 64 elements
            const int wave = tid >> 6;
            int sx[4][3];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int seg = wave + 4 * q;
                const int sg_ = seg < kSeg4 ? seg : 0;
                const int cnt = seg < kSeg4 ? rfl(plan.segCnt[sg_]) : 0;
                const int src0 = rfl(plan.segSrc[sg_]);
                const int glo = rfl(plan.segLo[sg_]), ghi = rfl(plan.segHi[sg_]);
                const int t = sg_ < 7 ? sg_ : sg_ - 7;
                const int32_t* src;
                uint32_t stride = 1;
                if (t == kJoin3 - 1) { src = reinterpret_cast<const int32_t*>(a.bl); stride = 2; }
                else {
                    const TrackView& tv = table_view(a, t);
                    src = sg_ < 7 ? tv.starts : tv.ends;
                }
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    const int kk = lane + 64 * m;
                    const int g = src0 + kk;
                    int x = g < glo ? INT_MIN : INT_MAX;
                    if (kk < cnt && g >= glo && g < ghi) x = src[(uint32_t)g * stride];
                    sx[q][m] = x;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int seg = wave + 4 * q;
                const int sg_ = seg < kSeg4 ? seg : 0;
                const int cnt = seg < kSeg4 ? rfl(plan.segCnt[sg_]) : 0;
                const int dst = rfl(plan.segDst[sg_]);
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    const int kk = lane + 64 * m;
                    if (kk < cnt) pool[dst + kk] = sx[q][m];
                }
                if (cnt > 192) {                              // long slice (dense table): the rest, serially
                    const int src0 = rfl(plan.segSrc[sg_]);
                    const int glo = rfl(plan.segLo[sg_]), ghi = rfl(plan.segHi[sg_]);
                    const int t = sg_ < 7 ? sg_ : sg_ - 7;
                    const int32_t* src;
                    uint32_t stride = 1;
                    if (t == kJoin3 - 1) { src = reinterpret_cast<const int32_t*>(a.bl); stride = 2; }
                    else {
                        const TrackView& tv = table_view(a, t);
                        src = sg_ < 7 ? tv.starts : tv.ends;
                    }
#pragma unroll 1
                    for (int kk = lane + 192; kk < cnt; kk += 64) {
                        const int g = src0 + kk;
                        int x = g < glo ? INT_MIN : INT_MAX;
                        if (g >= glo && g < ghi) x = src[(uint32_t)g * stride];
                        pool[dst + kk] = x;
                    }
                }
            }
        }

        // ---- contig-edge lanes blank the window bytes that lie outside their contig (reads as N)
        const bool edge = wrel < 0 || (uint32_t)(wrel + kWinBytes) > clen;
        if (edge) {
#pragma unroll 1
            for (int q = 0; q < kWinDw; ++q) {
                uint32_t m = 0;
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const int gi = wrel + 4 * q + bb;
                    m |= (gi >= 0 && (uint32_t)gi < clen) ? (0xffu << (8 * bb)) : 0u;
                }
                wrow[q] &= m;
            }
        }
        wrow[12] = 0;                                         // pad dword: read (and shifted out) by the packed fetches
        const uint8_t* wb = reinterpret_cast<const uint8_t*>(wrow);
        const uint32_t wrow_b = lds_addr(wrow);
        // reference base at contig offset p0 + d (0 outside the contig); window first, HBM beyond it
        auto ref_at = [&](int d) -> int {
            const int o = o0 + d;
            if (o >= 0 && o < kWinBytes) return wb[o];
            const int64_t gi = (int64_t)p0 + d;
            return (gi >= 0 && gi < (int64_t)clen) ? (int)a.ref[clo + gi] : 0;
        };
        // 8 window bytes starting at byte offset `o` (0 <= o, o + 8 <= 49): three dwords, two byte-aligns
        auto win8 = [&](int o, uint32_t& w0, uint32_t& w1) {
            const uint32_t ad = wrow_b + ((uint32_t)o & ~3u);
            const uint32_t d0 = lds_u32(ad), d1 = lds_u32(ad + 4), d2 = lds_u32(ad + 8);
            w0 = bytes_at(d1, d0, (uint32_t)o & 3u);
            w1 = bytes_at(d2, d1, (uint32_t)o & 3u);
        };

        // ---- is_hmer_indel.  so = window byte of the first base after the variant's alleles
        // (insertion/substitution: pos+1; deletion: pos+len(ref)); an hmer indel's run starts there.
        const int d_so = (indel && !ins) ? rl : 1;
        const int so = o0 + d_so;
        int hmer_len = 0, hmer_nuc = 0, run = 0;
        const uint32_t ab0 = (uint32_t)tail & 0xffu;          // indel: second base of the longer allele; else ref[0]
        if (indel && !(abl & 8)) {
            const uint32_t bb = ab0;
            const uint32_t rep = bb * 0x01010101u;
            // every base after the first equals bb: positions lo_off+1 .. lo_off+ln-1 (ln >= 2)
            const int nt = ln - 1 < 8 ? ln - 1 : 8;           // bytes of `tail` that belong to the allele
            const uint64_t diff = tail ^ (((uint64_t)rep << 32) | rep);
            const uint64_t msk = nt >= 8 ? ~0ull : ((1ull << (8 * nt)) - 1);
            bool mono = (diff & msk) == 0;
            if (ln > 9)
                for (int k = 9; k < ln; ++k) mono &= apool[lo_off + k] == bb;
            const uint32_t pstart = p0 + (uint32_t)d_so;
            if (mono && pstart < clen) {
                if (so + 12 <= kWinBytes) {
                    // 12-byte look-ahead on packed bytes: count the leading bytes equal to bb
                    const uint32_t ad = wrow_b + ((uint32_t)so & ~3u);
                    const uint32_t d0 = lds_u32(ad), d1 = lds_u32(ad + 4), d2 = lds_u32(ad + 8), d3 = lds_u32(ad + 12);
                    const uint32_t sh = (uint32_t)so & 3u;
                    const uint32_t z0 = lead_zero_bytes(bytes_at(d1, d0, sh) ^ rep);
                    const uint32_t z1 = lead_zero_bytes(bytes_at(d2, d1, sh) ^ rep);
                    const uint32_t z2 = lead_zero_bytes(bytes_at(d3, d2, sh) ^ rep);
                    const int nrun = (int)(z0 < 4 ? z0 : 4 + (z1 < 4 ? z1 : 4 + z2));
                    run = nrun;
                    if (nrun == 12)
                        while (ref_at(d_so + run) == (int)bb && pstart + (uint32_t)run < clen) ++run;
                } else {
                    while (pstart + (uint32_t)run < clen && ref_at(d_so + run) == (int)bb) ++run;
                }
                const uint32_t room = clen - pstart;           // an N run may not run past the contig end
                if ((uint32_t)run > room) run = (int)room;
                if (run > 0) {
                    hmer_len = run + (ins ? 0 : rl - 1);
                    hmer_nuc = (int)bb;
                }
            }
        }
        const bool is_h = indel && hmer_len > 0;
        const int group = !indel ? 0 : (is_h ? 1 : 2);

        // ---- record slots: one returning atomic per wave and group, consumed after the quantisation
        const bool pg_ok = (okbits >> group) & 1;
        const bool mine = live && pg_ok;
        unsigned slot_base = 0, grank = 0;
        {
            const int shard = tile & (kShards - 1);
            unsigned long long m[UGVC_N_GROUPS];
#pragma unroll
            for (int g = 0; g < UGVC_N_GROUPS; ++g) m[g] = __ballot(mine && group == g);
            const unsigned long long mg = lane == 0 ? m[0] : (lane == 1 ? m[1] : m[2]);
            unsigned got = 0;
            if (lane < UGVC_N_GROUPS && mg != 0 && !(abl & 16))
                got = atomicAdd(&v.counters[(lane * kShards + shard) * kCounterStride], (unsigned)__popcll(mg));
            const unsigned long long mm = group == 0 ? m[0] : (group == 1 ? m[1] : m[2]);
            grank = __popcll(mm & ((1ull << lane) - 1));
            slot_base = got;
        }

        // ---- get_motif_around (5), gc_content (10): bases at pos-5 .. pos+5 as packed bytes
        int lm = 0, rm = 0, lmb4 = 0, rmb0 = 0;
        bool motif_n = false;
        float gc = 0.0f;
        const int d_r = is_h ? d_so + run : (indel ? rl : 1);   // right motif: pos+1 | pos+len(ref) | past the run
        if (abl & 8) {
            lm = 1; rm = 2; lmb4 = 1; rmb0 = 2;
        } else {
            constexpr uint32_t kW5 = 125u | (25u << 8) | (5u << 16) | (1u << 24);
            const bool fastw = o0 >= 5 && p0 >= 5 && p0 + 6 <= clen;        // pos-5 .. pos+5 inside window and contig
            if (fastw) {
                uint32_t m0, m1, m2;
                {
                    const int o = o0 - 5;
                    const uint32_t ad = wrow_b + ((uint32_t)o & ~3u);
                    const uint32_t d0 = lds_u32(ad), d1 = lds_u32(ad + 4), d2 = lds_u32(ad + 8), d3 = lds_u32(ad + 12);
                    const uint32_t sh = (uint32_t)o & 3u;
                    m0 = bytes_at(d1, d0, sh); m1 = bytes_at(d2, d1, sh); m2 = bytes_at(d3, d2, sh);
                }
                // left motif: W[0..4] (substitution) or W[1..5] (indel)
                const uint32_t l4 = indel ? bytes_at(m1, m0, 1) : m0;
                const uint32_t l5 = indel ? (m1 >> 8) & 0xffu : m1 & 0xffu;
                lm = (int)(__builtin_amdgcn_udot4(l4, kW5, 0u, false) * 5u + l5);
                lmb4 = (int)l5;
                motif_n = nonzero_bytes(l4) != 0x80808080u || l5 == 0;
                // gc over W[1..10]: bytes 1..3 of m0, all of m1, bytes 0..2 of m2; every base that is neither A (1)
                // nor T (4) counts (N with G/C, as the oracle)
                auto gc_flags = [](uint32_t x) { return nonzero_bytes(x ^ 0x01010101u) & nonzero_bytes(x ^ 0x04040404u); };
                const int gc_cnt = __builtin_popcount(gc_flags(m0) & 0x80808000u) + __builtin_popcount(gc_flags(m1)) +
                                   __builtin_popcount(gc_flags(m2) & 0x00808080u);
                gc = gc_lut[gc_cnt * 11 + 10];
            } else {
                int gc_cnt = 0, gc_len = 0;
#pragma unroll 1
                for (int k = 0; k < kGcWindow; ++k) {
                    const uint32_t pw = p0 + 1 - kGcWindow / 2 + k;   // wraps below 0 -> fails the bound test
                    const bool inb = pw < clen;
                    const int bb = ref_at(k + 1 - 5);
                    gc_len += inb;
                    gc_cnt += inb && bb != 1 && bb != 4;
                }
                gc = gc_lut[gc_cnt * 11 + gc_len];
#pragma unroll 1
                for (int k = 0; k < kMotif; ++k) {
                    const int b = ref_at((indel ? k + 1 : k) - 5);
                    lm = lm * 5 + b;
                    motif_n |= b == 0;
                    lmb4 = b;
                }
            }
            if (o0 + d_r + 8 <= kWinBytes + 1 && o0 + d_r >= 0) {
                uint32_t r4, r5;
                win8(o0 + d_r, r4, r5);
                r5 &= 0xffu;
                rm = (int)(__builtin_amdgcn_udot4(r4, kW5, 0u, false) * 5u + r5);
                rmb0 = (int)(r4 & 0xffu);
                motif_n |= nonzero_bytes(r4) != 0x80808080u || r5 == 0;
            } else {
#pragma unroll 1
                for (int k = 0; k < kMotif; ++k) {
                    const int b = ref_at(d_r + k);
                    rm = rm * 5 + b;
                    motif_n |= b == 0;
                    if (k == 0) rmb0 = b;
                }
            }
        }

        // ---- cycle skip
        int css = 3;
        if (!indel && !(abl & 8)) {
            if (rl == 1) {
                const int rb = (int)ab0, abase = (int)alt0;
                if (motif_n || rb == 0 || abase == 0) css = 0;
                else css = css_lds[((lmb4 - 1) << 6) | ((rb - 1) << 4) | ((abase - 1) << 2) | (rmb0 - 1)];
            } else {
                bool has_n = motif_n;
                for (int k = 0; k < rl; ++k) has_n |= apool[ro + k] == 0 || apool[ao + k] == 0;
                if (has_n) css = 0;
                else {
                    // multi-base substitution (rare): the generic flow-key walk over left motif + allele + right motif
                    auto lmot = [&](int q) -> int { return ref_at(q - kMotif); };          // pos-5 .. pos-1
                    auto rmot = [&](int q) -> int { return ref_at(d_r + q); };
                    auto seq_r = [&](int k) -> int {
                        if (k < kMotif) return lmot(k);
                        if (k < kMotif + rl) return apool[ro + k - kMotif];
                        return rmot(k - kMotif - rl);
                    };
                    auto seq_a = [&](int k) -> int {
                        if (k < kMotif) return lmot(k);
                        if (k < kMotif + rl) return apool[ao + k - kMotif];
                        return rmot(k - kMotif - rl);
                    };
                    css = cycle_skip_walk(rl + 2 * kMotif, a.flow, seq_r, seq_a);
                }
            }
        }
        __syncthreads();                                      // staged slices visible
        if (prof_on && tid == 0) { const unsigned long long now_ = clock64(); prof_lds[2] += now_ - prof_t; prof_t = now_; }

        // ---- joins: rank among the starts of every table + blacklist positions, descents in lock-step over
        // sentinel-padded slices: a step is add, ds_read, compare, select
        uint8_t flags = 0;
        bool inside_run = false, close_run = false;
        if (abl & 2) {
        } else if (staged) {
            uint32_t p[kJoin3];
            int tb_[kJoin3];
#pragma unroll
            for (int t = 0; t < kJoin3; ++t) {
                p[t] = (uint32_t)rfl(plan.p0[t]);
                tb_[t] = rfl(plan.bits[t]);
            }
            const int bits = rfl(plan.maxbits);
            for (int s = bits - 1; s >= 0; --s) {
                const uint32_t step = 4u << s;
                int x[kJoin3];
                uint32_t cand[kJoin3];
#pragma unroll
                for (int t = 0; t < kJoin3; ++t)
                    if (s < tb_[t]) { cand[t] = p[t] + step; x[t] = lds_i32(cand[t]); }
#pragma unroll
                for (int t = 0; t < kJoin3; ++t)
                    if (s < tb_[t]) p[t] = x[t] < pos ? cand[t] : p[t];
            }
            // p[t] = address of starts[sg-1], sg = #starts < pos; sentinels stand in for every range check
            {
                const uint32_t dE = (uint32_t)rfl(plan.dE[0]);
                if (a.has_runs) {
                    const int s1v = lds_i32(p[0]), s0v = lds_i32(p[0] + 4);
                    const int e1v = lds_i32(p[0] + dE);
                    const bool ins_run = e1v >= pos;
                    const uint32_t pe = p[0] + dE + (e1v < pos ? 4u : 0u);        // address of ends[eg]
                    const int ee = lds_i32(pe), em = lds_i32(pe - 4);
                    const int Di = a.hpol_dist;
                    const uint32_t D = (uint32_t)Di;
                    // |pos - x| < D without overflow on the sentinels: pos - x + (D-1) in [0, 2D-2]
                    auto near = [&](int x) { return Di > 0 && ((uint32_t)pos - (uint32_t)x) + (D - 1) <= 2 * D - 2; };
                    const bool cd = near(s1v) || near(s0v) || near(em) || near(ee);
                    inside_run = ins_run;
                    close_run = cd && !ins_run;
                }
            }
#pragma unroll
            for (int t = 1; t < kJoin3 - 1; ++t) {
                if (t - 1 >= a.n_tracks) continue;
                const uint32_t dE = (uint32_t)rfl(plan.dE[t]);
                const int e1v = lds_i32(p[t] + dE), e2v = lds_i32(p[t] + dE - 4);
                const bool in = e1v >= pos && e2v < pos;
                flags |= in ? (uint8_t)(1u << (UGVC_FLAG_TRACK0_SHIFT + t - 1)) : 0;
            }
            if (a.n_bl > 0) {
                const int x = lds_i32(p[kJoin3 - 1] + 4);
                if (x == pos) flags |= UGVC_FLAG_COHORT_FP;
            }
        } else {
            // dense tile (slices exceed the pool): the same tests on the HBM copies, contig range from the plan
#pragma unroll 1
            for (int t = 0; t < kJoin3 - 1; ++t) {
                if (!table_present(a, t)) continue;
                const TrackView& tv = table_view(a, t);
                const int plo = rfl(plan.plo[t]), phi = rfl(plan.phi[t]);
                const int sg = lb_i32_g(tv.starts, rfl(plan.lo[t]), rfl(plan.hi[t]), pos);
                const bool valid = sg > plo;
                const int e1v = valid ? tv.ends[sg - 1] : 0;
                if (t == 0) {
                    const bool ins_run = valid && e1v >= pos;
                    if (phi > plo) {
                        const int eg = valid ? sg - 1 + (e1v < pos ? 1 : 0) : sg;
                        const int D = a.hpol_dist;
                        auto near = [&](int x) { const int d = pos - x; return (d < 0 ? -d : d) < D; };
                        bool cd = (valid && near(tv.starts[sg - 1])) || (sg <= phi - 1 && near(tv.starts[sg]));
                        cd = cd || (eg - 1 >= plo && near(tv.ends[eg - 1])) || (eg <= phi - 1 && near(tv.ends[eg]));
                        inside_run = ins_run;
                        close_run = cd && !ins_run;
                    }
                } else {
                    const bool in = valid && e1v >= pos && (sg - 1 == plo || tv.ends[sg - 2] < pos);
                    flags |= in ? (uint8_t)(1u << (UGVC_FLAG_TRACK0_SHIFT + t - 1)) : 0;
                }
            }
            if (a.n_bl > 0) {
                const uint64_t key = ((uint64_t)c << 32) | (uint32_t)pos;
                const int r = lb_u64_g(a.bl, rfl(plan.lo[kJoin3 - 1]), rfl(plan.hi[kJoin3 - 1]), key);
                if (r < (int)a.n_bl && a.bl[r] == key) flags |= UGVC_FLAG_COHORT_FP;
            }
        }
        if (a.mark_hpol && (inside_run || close_run)) flags |= UGVC_FLAG_HPOL_RUN;
        if (live) a.flags[i] = flags;
        if (!pg_ok && live) {              // no model for this variant type: score 0, PASS
            a.score[i] = 0.f;
            a.filter[i] = UGVC_FILTER_PASS;
        }
        if (prof_on && tid == 0) { const unsigned long long now_ = clock64(); prof_lds[3] += now_ - prof_t; prof_t = now_; }

        // ---- quantise: feature -> rank among the group's sorted thresholds
        const float vaf = dp > 0 ? __fdiv_rn((float)ada, (float)dp) : 0.0f;
        const uint4* dsc = desc_lds + group * kMaxFeatures;
        // the three code dwords accumulate in the lane's window row (free now), one LDS atomic OR per feature:
        // no per-feature dword masks in scalar registers, two vector instructions per code
        {
            // the seven 0/1 features sit in fixed bits 25..31 of dword 2; rank code == value wherever the group's
            // model tests them below 1 (boolmask3), else constant 0.  Plain stores first, atomics after.
            const uint32_t bits7 = (inside_run ? 1u : 0u) | (close_run ? 2u : 0u) | ((uint32_t)(flags >> UGVC_FLAG_TRACK0_SHIFT) << 2);
            const uint32_t bm = group == 0 ? v.boolmask3[0] : (group == 1 ? v.boolmask3[1] : v.boolmask3[2]);
            wrow[0] = 0; wrow[1] = 0;
            wrow[2] = (pg_ok && !(abl & 4)) ? (bits7 & bm) << 25 : 0u;
        }
        auto put = [&](const uint4& d, uint32_t code) {
            __hip_atomic_fetch_or(reinterpret_cast<UGVC_LDS uint32_t*>((uintptr_t)(wrow_b + d.z)), code << d.w,
                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        };
        const bool upper = (gbtbits >> group) & 1;
        if (pg_ok && !(abl & 4)) {
            {
                const float fx[4] = {qual, sor, vaf, gc};
                const int fj[4] = {0, 1, 5, 13};
                const uint32_t thr_b = lds_addr(thr_lds);
                uint32_t q[4], qend[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint4 d = dsc[fj[k]];
                    q[k] = thr_b + 4u * (d.x & 0xFFFFFu) - 4;
                    qend[k] = q[k] + 4u * d.y;
                }
                const int fb0 = v.thr_bits4[0], fb1 = v.thr_bits4[1], fb2 = v.thr_bits4[2], fb3 = v.thr_bits4[3];
                const int fbm = max(max(fb0, fb1), max(fb2, fb3));
                for (int s = fbm - 1; s >= 0; --s) {
                    uint32_t cand[4];
                    float t[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        cand[k] = q[k] + (4u << s);
                        t[k] = lds_f32(cand[k]);
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const bool lt = upper ? t[k] <= fx[k] : t[k] < fx[k];
                        q[k] = ((int32_t)(qend[k] - cand[k]) >= 0 && lt) ? cand[k] : q[k];
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint4 d = dsc[fj[k]];
                    uint32_t cd = (q[k] + 4 - (thr_b + 4u * (d.x & 0xFFFFFu))) >> 2;
                    if (fx[k] != fx[k]) cd = d.y;                 // NaN compares false: always the right branch
                    put(d, cd);
                }
            }
            int iv[kMaxFeatures];
            iv[0] = iv[1] = iv[5] = iv[13] = 0;
            iv[2] = dp; iv[3] = adr; iv[4] = ada; iv[6] = gq; iv[7] = classify; iv[8] = indel_length;
            iv[9] = hmer_len; iv[10] = hmer_nuc; iv[11] = lm; iv[12] = rm; iv[14] = css;
#pragma unroll
            for (int j = 15; j < kMaxFeatures; ++j) iv[j] = 0;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int j0 = half == 0 ? 2 : 9, j1 = half == 0 ? 9 : 15;
                uint32_t code[kMaxFeatures];
                bool slow = false;
#pragma unroll
                for (int j = j0; j < j1; ++j) {
                    if (j == 5 || j == 13) continue;
                    if (j >= F) break;
                    const uint4 d = dsc[j];
                    const uint32_t len = d.y;
                    const uint32_t x = (uint32_t)iv[j];
                    const uint32_t idx = x < len ? x : len - 1;
                    code[j] = v.lut[(d.x & 0xFFFFFu) + idx];
                    slow |= x >= len && ((d.x >> 30) == 1 || (int32_t)x < 0);        // past a cut table, or negative
                }
                if (slow) {                                       // value beyond the LUT: search the thresholds in HBM
#pragma unroll
                    for (int j = j0; j < j1; ++j) {
                        if (j == 5 || j == 13) continue;
                        if (j >= F) break;
                        const uint4 d = dsc[j];
                        if ((uint32_t)iv[j] >= d.y && ((d.x >> 30) == 1 || iv[j] < 0)) {
                            const FeatDesc fd = v.desc[group * kMaxFeatures + j];
                            const uint32_t toff = fd.thr & 0xFFFFF, tlen = fd.thr >> 20;
                            const float x = (float)iv[j];
                            uint32_t bb = 0, len = tlen;
                            while (len > 0) {
                                const uint32_t hf = len >> 1;
                                const float t = v.thr[toff + bb + hf];
                                const bool lt = upper ? t <= x : t < x;
                                bb = lt ? bb + hf + 1 : bb;
                                len = lt ? len - hf - 1 : hf;
                            }
                            code[j] = bb;
                        }
                    }
                }
#pragma unroll
                for (int j = j0; j < j1; ++j) {
                    if (j == 5 || j == 13) continue;
                    if (j >= F) break;
                    put(dsc[j], code[j]);
                }
            }
        }
        if (prof_on && tid == 0) { const unsigned long long now_ = clock64(); prof_lds[4] += now_ - prof_t; prof_t = now_; }

        // ---- append {codes, variant index} to the group's sharded record list
        {
            const unsigned b0 = __shfl(slot_base, 0), b1 = __shfl(slot_base, 1), b2 = __shfl(slot_base, 2);
            const unsigned sb = group == 0 ? b0 : (group == 1 ? b1 : b2);
            const int shard = tile & (kShards - 1);
            const uint32_t c0 = wrow[0], c1 = wrow[1], c2 = wrow[2];
            if (mine && !(abl & 16))
                v.records[group][(size_t)shard * v.shard_cap + sb + grank] = make_uint4(c0, c1, c2, i);
        }
        __syncthreads();                                      // plan / pool are rewritten by the next tile
        if (prof_on && tid == 0) { const unsigned long long now_ = clock64(); prof_lds[5] += now_ - prof_t; prof_t = now_; prof_lds[6] += 1; }
    }
    if (prof_on && tid == 0) {
        for (int k = 0; k < 7; ++k) atomicAdd(&v.prof[k], prof_lds[k]);
    }
}

int launch_filter_v4(ugvc_ctx* ctx, const FilterArgs& a) {
    if (a.n == 0) return 0;
    V2Args v;
    v.f = a;
    if (v2_fill_args(ctx, v, a.n, ctx->n_tiles4)) return -1;
    const int64_t nbr = std::max<int64_t>((int64_t)(v.n_blocks + 1) * 8, UGVC_N_GROUPS * kShards);
    hipLaunchKernelGGL(bracket4_kernel, dim3((unsigned)((nbr + 255) / 256)), dim3(256), 0, ctx->stream, v);
    const int k1_bpc = ((a.ablate >> 12) & 3) ? ((a.ablate >> 12) & 3) : 4;
    const int k1_grid = std::min(v.n_blocks, ctx->n_cus * k1_bpc);
    hipLaunchKernelGGL(featurize4_kernel, dim3((unsigned)k1_grid), dim3(kBlock), 0, ctx->stream, v);
    return launch_forest3(ctx, v, a);
}

}  // namespace ugvc
