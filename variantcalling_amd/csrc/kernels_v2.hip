// v2 fast path kernels (gfx950): K0 join brackets, K1 featurize + lookup + quantise, K2 LDS-resident
// forest walk.  Design notes in ugvc_v2.hpp; semantics identical to kernels_filter.hip / the oracle.
#include "ugvc_v2.hpp"

namespace ugvc {


__device__ __forceinline__ int lb_i32(const int32_t* __restrict__ a, int lo, int hi, int key) {
    int base = lo, len = hi - lo;          // absolute index of the first element >= key in a[lo:hi)
    while (len > 0) {
        const int half = len >> 1;
        const bool lt = a[base + half] < key;
        base = lt ? base + half + 1 : base;
        len = lt ? len - half - 1 : half;
    }
    return base;
}

__device__ __forceinline__ int lb_u64(const uint64_t* __restrict__ a, int lo, int hi, uint64_t key) {
    int base = lo, len = hi - lo;
    while (len > 0) {
        const int half = len >> 1;
        const bool lt = a[base + half] < key;
        base = lt ? base + half + 1 : base;
        len = lt ? len - half - 1 : half;
    }
    return base;
}

__device__ __forceinline__ const int32_t* join_array(const FilterArgs& f, int a, const int32_t*& ptr, bool& present) {
    // a: 0 runs.starts, 1 runs.ends, 2+2t track t starts, 3+2t track t ends
    const int t = (a >> 1) - 1;
    if (a < 2) {
        present = f.has_runs != 0;
        ptr = f.runs.ptr;
        return (a & 1) ? f.runs.ends : f.runs.starts;
    }
    present = t < f.n_tracks;
    ptr = f.tracks[t].ptr;
    return (a & 1) ? f.tracks[t].ends : f.tracks[t].starts;
}

// ---- K0: per-block brackets of every sorted side table -----------------------------------
// brackets[b][a] = global index of the first entry of array a that is >= the first variant of
// block b (in that variant's contig); row n_blocks holds the array lengths.  Because variants
// are sorted, every variant of block b finds its lower bound inside [br[b][a], br[b+1][a]].
__global__ void bracket_kernel(const V2Args v) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nb = v.n_blocks;
    if (gid >= (int64_t)(nb + 1) * kJoinArrays) return;
    const int b = (int)(gid / kJoinArrays), a = (int)(gid - (int64_t)b * kJoinArrays);
    const FilterArgs& f = v.f;
    int out = 0;
    if (a == kJoinArrays - 1) {
        if (f.n_bl > 0) {
            if (b == nb) out = (int)f.n_bl;
            else {
                const int64_t i = (int64_t)b * kBlock;
                out = lb_u64(f.bl, 0, (int)f.n_bl, ((uint64_t)f.contig[i] << 32) | (uint32_t)f.pos[i]);
            }
        }
    } else {
        const int32_t* ptr;
        bool present;
        const int32_t* A = join_array(f, a, ptr, present);
        if (present) {
            if (b == nb) out = ptr[f.n_contigs];
            else {
                const int64_t i = (int64_t)b * kBlock;
                const int c = f.contig[i];
                out = lb_i32(A, ptr[c], ptr[c + 1], f.pos[i]);
            }
        }
    }
    v.brackets[gid] = out;
}

// ---- K1 ------------------------------------------------------------------------------------
constexpr int kThrLds = 4096;

struct StagedTable {            // one sorted i32 side array, either staged in LDS or read from HBM
    const int32_t* g;           // global array
    const int32_t* s;           // LDS copy of g[base : base + count)
    int base, staged;
    __device__ __forceinline__ int at(int i) const { return staged ? s[i - base] : g[i]; }
    __device__ __forceinline__ int lower_bound(int lo, int hi, int key) const {
        int b = lo, len = hi - lo;
        while (len > 0) {
            const int half = len >> 1;
            const bool lt = at(b + half) < key;
            b = lt ? b + half + 1 : b;
            len = lt ? len - half - 1 : half;
        }
        return b;
    }
};

struct Window {                 // 64 reference bases around the variant, staged in LDS as win[dword][lane]
    const uint32_t* w;          // &win[0][tid]
    const uint8_t* __restrict__ codes;
    int64_t wbase, lo, hi;
    __device__ __forceinline__ int at(int64_t i) const {
        if (i < lo || i >= hi) return 0;
        const int64_t off = i - wbase;
        if (off >= 0 && off < 64) return (w[(off >> 2) * kBlock] >> (8 * (off & 3))) & 0xff;
        return codes[i];
    }
};

template <class SeqR, class SeqA>
__device__ __forceinline__ int cycle_skip_generic(int L, const uint8_t flow[4], SeqR seq_r, SeqA seq_a) {
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

__global__ __launch_bounds__(kBlock) void featurize_kernel(const V2Args v) {
    __shared__ uint32_t win[16 * kBlock];                          // 16 KB
    __shared__ int32_t stage[(kJoinArrays - 1) * kStageCap];       // 12 arrays
    __shared__ uint64_t stage_bl[kStageCap];
    __shared__ float thr_lds[kThrLds];
    __shared__ uint4 desc_lds[UGVC_N_GROUPS * kMaxFeatures];
    __shared__ int br_lo[kJoinArrays], br_hi[kJoinArrays], st_base[kJoinArrays], st_ok[kJoinArrays];
    __shared__ int blk_cnt[UGVC_N_GROUPS];
    __shared__ unsigned blk_base[UGVC_N_GROUPS];

    const FilterArgs& a = v.f;
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int64_t i_raw = (int64_t)b * kBlock + tid;
    const bool live = i_raw < a.n;
    const int64_t i = live ? i_raw : a.n - 1;     // idle lanes of the last block shadow the last variant
    const int F = UGVC_N_BASE_FEATURES + a.n_tracks;

    // ---- block-wide tables into LDS
    if (tid < kJoinArrays) {
        const int lo = v.brackets[(int64_t)b * kJoinArrays + tid];
        const int hi = v.brackets[(int64_t)(b + 1) * kJoinArrays + tid];
        const int na = v.brackets[(int64_t)v.n_blocks * kJoinArrays + tid];
        br_lo[tid] = lo;
        br_hi[tid] = hi;
        const int L = lo > 0 ? lo - 1 : 0;
        const int H = hi + 1 < na ? hi + 1 : na;
        st_base[tid] = L;
        st_ok[tid] = (H - L) <= kStageCap ? (H - L) : -1;          // count, or -1 = read from HBM
    }
    for (int k = tid; k < UGVC_N_GROUPS * kMaxFeatures; k += kBlock)
        desc_lds[k] = reinterpret_cast<const uint4*>(v.desc)[k];
    for (int k = tid; k < v.thr_lds_len; k += kBlock) thr_lds[k] = v.thr[k];

    // ---- per-variant columns + reference window
    int c = 0, pos = 1, rl = 1, al = 1;
    uint32_t ro = 0, ao = 0;
    int64_t g0 = 0, clo = 0, chi = 0, wbase = 0;
    {
        c = a.contig[i];
        pos = a.pos[i];
        rl = a.ref_len[i];
        al = a.alt_len[i];
        ro = a.ref_off[i];
        ao = a.alt_off[i];
        clo = a.contig_off[c];
        chi = a.contig_off[c + 1];
        g0 = clo + pos - 1;
        wbase = (g0 - 8) & ~(int64_t)15;
        if (wbase < 0) wbase = 0;
        // the reference buffer is padded by 64 bytes, so the four 16-byte loads never overrun
        const uint4* src = reinterpret_cast<const uint4*>(a.ref + wbase);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 x = src[q];
            win[(4 * q + 0) * kBlock + tid] = x.x;
            win[(4 * q + 1) * kBlock + tid] = x.y;
            win[(4 * q + 2) * kBlock + tid] = x.z;
            win[(4 * q + 3) * kBlock + tid] = x.w;
        }
    }
    __syncthreads();

    // ---- stage the slice of every side table this block can touch
    for (int t = 0; t < kJoinArrays - 1; ++t) {
        const int cnt = st_ok[t];
        if (cnt > 0) {
            const int32_t* ptr;
            bool present;
            const int32_t* A = join_array(a, t, ptr, present);
            const int base = st_base[t];
            for (int k = tid; k < cnt; k += kBlock) stage[t * kStageCap + k] = A[base + k];
        }
    }
    {
        const int cnt = st_ok[kJoinArrays - 1];
        const int base = st_base[kJoinArrays - 1];
        for (int k = tid; k < cnt; k += kBlock) stage_bl[k] = a.bl[base + k];
    }
    __syncthreads();

    Window rw{win + tid, a.ref, wbase, clo, chi};
    const uint8_t* __restrict__ pool = a.alleles;

    // ---- classify_indel / is_hmer_indel
    const bool indel = rl != al;
    const int classify = !indel ? 0 : (rl < al ? 1 : 2);
    const int indel_length = rl < al ? al - rl : rl - al;
    int hmer_len = 0, hmer_nuc = 0;
    if (indel) {
        const uint8_t* alle = pool + (classify == 1 ? ao : ro);
        const int ln = classify == 1 ? al : rl;
        const int bb = alle[1];
        bool mono = true;
        for (int k = 2; k < ln; ++k) mono &= alle[k] == bb;
        const int64_t start = classify == 1 ? g0 + 1 : g0 + rl;
        if (mono && start >= clo && start < chi && rw.at(start) == bb) {
            int64_t j = start + 1;
            while (j < chi && rw.at(j) == bb) ++j;
            hmer_len = (int)(j - start) + (classify == 1 ? 0 : rl - 1);
            hmer_nuc = bb;
        }
    }
    const bool is_h = indel && hmer_len > 0;
    const int group = !indel ? 0 : (is_h ? 1 : 2);

    // ---- motifs, gc
    const int64_t lstart = indel ? g0 - (kMotif - 1) : g0 - kMotif;
    const int64_t rstart = !indel ? g0 + 1 : (is_h ? g0 + 1 + hmer_len : g0 + rl);
    int lmb[kMotif], rmb[kMotif];
    int lm = 0, rm = 0;
    bool motif_n = false;
#pragma unroll
    for (int k = 0; k < kMotif; ++k) {
        lmb[k] = rw.at(lstart + k);
        rmb[k] = rw.at(rstart + k);
        lm = lm * 5 + lmb[k];
        rm = rm * 5 + rmb[k];
        motif_n |= lmb[k] == 0 || rmb[k] == 0;
    }
    int gc_cnt = 0, gc_len = 0;
#pragma unroll
    for (int k = 0; k < kGcWindow; ++k) {
        const int64_t w = g0 + 1 - kGcWindow / 2 + k;
        const bool inb = w >= clo && w < chi;
        const int bb = rw.at(w);
        gc_len += inb;
        gc_cnt += inb && bb != 1 && bb != 4;
    }
    const float gc = gc_len > 0 ? (float)((double)gc_cnt / (double)gc_len) : 0.0f;

    // ---- cycle skip
    int css = 3;
    if (!indel) {
        if (rl == 1) {
            const int rb = pool[ro], ab = pool[ao];
            if (motif_n || rb == 0 || ab == 0) css = 0;
            else css = v.css_lut[((lmb[kMotif - 1] - 1) << 6) | ((rb - 1) << 4) | ((ab - 1) << 2) | (rmb[0] - 1)];
        } else {
            bool has_n = motif_n;
            for (int k = 0; k < rl; ++k) has_n |= pool[ro + k] == 0 || pool[ao + k] == 0;
            if (has_n) css = 0;
            else {
                auto seq_r = [&](int k) -> int {
                    if (k < kMotif) return lmb[0] * (k == 0) + lmb[1] * (k == 1) + lmb[2] * (k == 2) + lmb[3] * (k == 3) + lmb[4] * (k == 4);
                    if (k < kMotif + rl) return pool[ro + k - kMotif];
                    const int q = k - kMotif - rl;
                    return rmb[0] * (q == 0) + rmb[1] * (q == 1) + rmb[2] * (q == 2) + rmb[3] * (q == 3) + rmb[4] * (q == 4);
                };
                auto seq_a = [&](int k) -> int {
                    if (k < kMotif) return lmb[0] * (k == 0) + lmb[1] * (k == 1) + lmb[2] * (k == 2) + lmb[3] * (k == 3) + lmb[4] * (k == 4);
                    if (k < kMotif + rl) return pool[ao + k - kMotif];
                    const int q = k - kMotif - rl;
                    return rmb[0] * (q == 0) + rmb[1] * (q == 1) + rmb[2] * (q == 2) + rmb[3] * (q == 3) + rmb[4] * (q == 4);
                };
                css = cycle_skip_generic(rl + 2 * kMotif, a.flow, seq_r, seq_a);
            }
        }
    }

    // ---- joins against the staged side tables
    uint8_t flags = 0;
    bool inside_run = false, close_run = false;
    bool trk[UGVC_MAX_TRACKS];
#pragma unroll
    for (int t = -1; t < UGVC_MAX_TRACKS; ++t) {
        const int as = 2 * (t + 1), ae = as + 1;
        const bool present = t < 0 ? a.has_runs != 0 : t < a.n_tracks;
        bool inside = false;
        if (present) {
            const TrackView& tv = t < 0 ? a.runs : a.tracks[t < 0 ? 0 : t];
            const int plo = tv.ptr[c], phi = tv.ptr[c + 1];
            StagedTable S{tv.starts, stage + as * kStageCap, st_base[as], st_ok[as] >= 0};
            StagedTable E{tv.ends, stage + ae * kStageCap, st_base[ae], st_ok[ae] >= 0};
            int slo = br_lo[as] > plo ? br_lo[as] : plo, shi = br_hi[as] < phi ? br_hi[as] : phi;
            int elo = br_lo[ae] > plo ? br_lo[ae] : plo, ehi = br_hi[ae] < phi ? br_hi[ae] : phi;
            if (shi < slo) shi = slo;
            if (ehi < elo) ehi = elo;
            const int sg = S.lower_bound(slo, shi, pos);      // global index, #starts < pos within contig
            const int eg = E.lower_bound(elo, ehi, pos);
            inside = (sg - 1) == eg;
            if (t < 0 && phi > plo) {
                const int64_t p = pos, D = a.hpol_dist;
                auto near = [&](int64_t x) { const int64_t d = p - x; return (d < 0 ? -d : d) < D; };
                const int s0 = sg - 1 < plo ? plo : sg - 1, s1 = sg > phi - 1 ? phi - 1 : sg;
                const int e0 = eg - 1 < plo ? plo : eg - 1, e1 = eg > phi - 1 ? phi - 1 : eg;
                bool cd = near(S.at(s0));
                cd |= near(S.at(s1));
                cd |= near(E.at(e0));
                cd |= near(E.at(e1));
                inside_run = inside;
                close_run = cd && !inside;
            }
        }
        if (t >= 0) {
            trk[t] = inside;
            flags |= inside ? (uint8_t)(1u << (UGVC_FLAG_TRACK0_SHIFT + t)) : 0;
        }
    }
    if (a.mark_hpol && (inside_run || close_run)) flags |= UGVC_FLAG_HPOL_RUN;
    if (a.n_bl > 0) {
        const int t = kJoinArrays - 1;
        const uint64_t key = ((uint64_t)c << 32) | (uint32_t)pos;
        const int lo = br_lo[t], hi = br_hi[t];
        const bool staged = st_ok[t] >= 0;
        const int base = st_base[t];
        int bb = lo, len = hi - lo;
        while (len > 0) {
            const int half = len >> 1;
            const uint64_t x = staged ? stage_bl[bb + half - base] : a.bl[bb + half];
            const bool lt = x < key;
            bb = lt ? bb + half + 1 : bb;
            len = lt ? len - half - 1 : half;
        }
        if (bb < (int)a.n_bl && (staged ? stage_bl[bb - base] : a.bl[bb]) == key) flags |= UGVC_FLAG_COHORT_FP;
    }
    if (live) a.flags[i] = flags;

    const PackedGroupView& pg = v.pg[group];
    if (!pg.ok && live) {              // no model for this variant type: score 0, PASS
        a.score[i] = 0.f;
        a.filter[i] = UGVC_FILTER_PASS;
    }

    // ---- quantise: feature value -> rank code among the group's sorted thresholds, bit-packed
    const int dp = a.dp[i], adr = a.ad_ref[i], ada = a.ad_alt[i];
    const float vaf = dp > 0 ? __fdiv_rn((float)ada, (float)dp) : 0.0f;
    float fv[kMaxFeatures];
    int iv[kMaxFeatures];
    fv[0] = a.qual[i]; iv[0] = -1;
    fv[1] = a.sor[i]; iv[1] = -1;
    iv[2] = dp; iv[3] = adr; iv[4] = ada;
    fv[5] = vaf; iv[5] = -1;
    iv[6] = a.gq[i]; iv[7] = classify; iv[8] = indel_length; iv[9] = hmer_len; iv[10] = hmer_nuc;
    iv[11] = lm; iv[12] = rm;
    fv[13] = gc; iv[13] = -1;
    iv[14] = css; iv[15] = inside_run; iv[16] = close_run;
#pragma unroll
    for (int t = 0; t < UGVC_MAX_TRACKS; ++t) iv[UGVC_N_BASE_FEATURES + t] = (t < a.n_tracks && trk[t]) ? 1 : 0;
#pragma unroll
    for (int j = 0; j < kMaxFeatures; ++j)
        if (iv[j] >= 0 || !(j == 0 || j == 1 || j == 5 || j == 13)) fv[j] = (float)iv[j];

    uint32_t c0 = 0, c1 = 0, c2 = 0;
    if (pg.ok) {
        const bool upper = pg.kind == UGVC_MODEL_GBT;
#pragma unroll
        for (int j = 0; j < kMaxFeatures; ++j) {
            if (j >= F) break;
            const uint4 d = desc_lds[group * kMaxFeatures + j];
            const uint32_t kind = d.x >> 30;
            if (kind == 0) continue;
            const uint32_t thr_off = d.z & 0xFFFFF, thr_len = d.z >> 20;
            uint32_t code;
            if (kind == 1 && iv[j] >= 0 && (uint32_t)iv[j] < d.y) {
                code = v.lut[(d.x & 0xFFFFF) + iv[j]];
            } else {
                const float x = fv[j];
                if (x != x) code = thr_len;                 // NaN compares false: always the right branch
                else {
                    const bool in_lds = thr_off + thr_len <= (uint32_t)v.thr_lds_len;
                    uint32_t bb = 0, len = thr_len;
                    while (len > 0) {
                        const uint32_t half = len >> 1;
                        const float t = in_lds ? thr_lds[thr_off + bb + half] : v.thr[thr_off + bb + half];
                        const bool lt = upper ? t <= x : t < x;
                        bb = lt ? bb + half + 1 : bb;
                        len = lt ? len - half - 1 : half;
                    }
                    code = bb;
                }
            }
            const uint32_t dw = d.w & 3, sh = (d.w >> 2) & 31;
            const uint32_t val = code << sh;
            c0 |= dw == 0 ? val : 0;
            c1 |= dw == 1 ? val : 0;
            c2 |= dw == 2 ? val : 0;
        }
    }

    // ---- append {codes, variant index} to the group's record list.  Lists are sharded (shard =
    // block & 255) so the returning global atomics spread over 256 words per group, and a block
    // issues ONE atomic per group (wave counts are pre-aggregated in LDS).
    const int lane = tid & 63;
    const bool mine = live && pg.ok;
    int wave_off = 0;
    unsigned rank = 0;
    if (tid < UGVC_N_GROUPS) blk_cnt[tid] = 0;
    __syncthreads();
#pragma unroll
    for (int g = 0; g < UGVC_N_GROUPS; ++g) {
        const unsigned long long m = __ballot(mine && group == g);
        if (m == 0) continue;
        int off = 0;
        const int leader = __ffsll((long long)m) - 1;
        if (lane == leader) off = atomicAdd(&blk_cnt[g], __popcll(m));
        off = __shfl(off, leader);
        if (mine && group == g) {
            wave_off = off;
            rank = __popcll(m & ((1ull << lane) - 1));
        }
    }
    __syncthreads();
    const int shard = b & (kShards - 1);
    if (tid < UGVC_N_GROUPS && blk_cnt[tid] > 0)
        blk_base[tid] = atomicAdd(&v.counters[(tid * kShards + shard) * kCounterStride], (unsigned)blk_cnt[tid]);
    __syncthreads();
    if (mine)
        v.records[group][(size_t)shard * v.shard_cap + blk_base[group] + wave_off + rank] =
            make_uint4(c0, c1, c2, (uint32_t)i);
}

// ---- K2 ------------------------------------------------------------------------------------
// One workgroup = one variant-type group, whole forest in LDS:
//   nodes   u32  T x 2^D (1-based heap order, slot 0 unused): rank[0:16) | code-plane byte offset[16:32)
//   pairs   f64x2 unique RF leaf payloads, leaf_idx u16 T x 2^D   (GBT: leaf_f32 T x 2^D)
//   planes  u16  per wave [n_planes][64]: the chunk's rank codes, one plane per used feature
// A visit is: ds_read_b32 node -> ds_read_u16 code -> compare -> idx = 2*idx + (code > rank).
template <int KIND, int NT>
__device__ __forceinline__ void k2_batch(const uint32_t* __restrict__ nodes, const uint16_t* __restrict__ planes_lane,
                                         int t, int D, int NL, int (&leaf)[NT]) {
    int idx[NT];
    int tb[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) { idx[k] = 1; tb[k] = (t + k) * NL; }
    for (int d = 0; d < D; ++d) {
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const uint32_t w = nodes[tb[k] + idx[k]];
            const uint32_t code = *reinterpret_cast<const uint16_t*>(reinterpret_cast<const unsigned char*>(planes_lane) + (w >> 16));
            idx[k] = 2 * idx[k] + (code > (w & 0xFFFFu) ? 1 : 0);
        }
    }
#pragma unroll
    for (int k = 0; k < NT; ++k) leaf[k] = tb[k] + idx[k] - NL;
}

__global__ __launch_bounds__(kK2Threads) void forest_kernel(const V2Args v) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned shard_off[kShards + 1];
    __shared__ unsigned totals[UGVC_N_GROUPS];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int n_waves = blockDim.x >> 6;
    // ---- group totals (sum of the 256 shard counters of each group)
    if (tid < UGVC_N_GROUPS) totals[tid] = 0;
    __syncthreads();
    for (int k = tid; k < UGVC_N_GROUPS * kShards; k += blockDim.x) {
        const unsigned cshard = v.counters[k * kCounterStride];
        if (cshard) atomicAdd(&totals[k / kShards], cshard);
    }
    __syncthreads();
    // ---- block -> group: blocks are split over the groups in proportion to count x trees x depth
    const int B = gridDim.x;
    unsigned cnt[UGVC_N_GROUPS];
    double work[UGVC_N_GROUPS], tot = 0.0;
    for (int g = 0; g < UGVC_N_GROUPS; ++g) {
        cnt[g] = v.pg[g].ok ? totals[g] : 0u;
        work[g] = (double)cnt[g] * v.pg[g].T * v.pg[g].D;
        tot += work[g];
    }
    if (tot == 0.0) return;
    int nb[UGVC_N_GROUPS], used = 0, big = 0;
    for (int g = 0; g < UGVC_N_GROUPS; ++g) {
        nb[g] = cnt[g] ? (int)(B * (work[g] / tot) + 0.5) : 0;
        if (cnt[g] && nb[g] < 1) nb[g] = 1;
        used += nb[g];
        if (work[g] > work[big]) big = g;
    }
    nb[big] += B - used;
    if (nb[big] < 1) return;
    int g = 0, lb = blockIdx.x;
    while (g < UGVC_N_GROUPS - 1 && lb >= nb[g]) { lb -= nb[g]; ++g; }
    const PackedGroupView pg = v.pg[g];
    const unsigned n = cnt[g];
    if (n == 0 || nb[g] == 0) return;

    // ---- exclusive scan of this group's shard counts (one wave, 4 shards per lane)
    if (wave == 0) {
        unsigned x[4], s = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { x[k] = v.counters[(g * kShards + lane * 4 + k) * kCounterStride]; s += x[k]; }
        unsigned incl = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned y = __shfl_up(incl, d);
            if (lane >= d) incl += y;
        }
        unsigned run = incl - s;
#pragma unroll
        for (int k = 0; k < 4; ++k) { shard_off[lane * 4 + k] = run; run += x[k]; }
        if (lane == 63) shard_off[kShards] = run;
    }

    // ---- forest of this group into LDS
    const int D = pg.D, NL = 1 << D;
    const size_t n_nodes = (size_t)pg.T * NL;
    uint32_t* nodes = reinterpret_cast<uint32_t*>(smem);
    size_t off = (n_nodes * 4 + 15) & ~(size_t)15;
    double2* pairs = reinterpret_cast<double2*>(smem + off);
    float* leaf_f32 = reinterpret_cast<float*>(smem + off);
    off += pg.kind == UGVC_MODEL_RF ? (size_t)pg.n_pairs * 16 : ((n_nodes * 4 + 15) & ~(size_t)15);
    uint16_t* leaf_idx = reinterpret_cast<uint16_t*>(smem + off);
    if (pg.kind == UGVC_MODEL_RF) off += (n_nodes * 2 + 15) & ~(size_t)15;
    uint16_t* planes_all = reinterpret_cast<uint16_t*>(smem + off);
    for (size_t k = tid; k < n_nodes; k += blockDim.x) nodes[k] = pg.nodes[k];
    if (pg.kind == UGVC_MODEL_RF) {
        for (size_t k = tid; k < (size_t)pg.n_pairs; k += blockDim.x) pairs[k] = pg.pairs[k];
        for (size_t k = tid; k < n_nodes; k += blockDim.x) leaf_idx[k] = pg.leaf_idx[k];
    } else {
        for (size_t k = tid; k < n_nodes; k += blockDim.x) leaf_f32[k] = pg.leaf_f32[k];
    }
    __syncthreads();

    const int P = pg.n_planes;
    uint16_t* planes = planes_all + (size_t)wave * P * 64;
    const uint16_t* planes_lane = planes + lane;
    const unsigned waves = (unsigned)nb[g] * n_waves;
    const uint4* __restrict__ rec = v.records[g];
    const int T = pg.T;
    for (unsigned chunk = (unsigned)lb * n_waves + wave; (uint64_t)chunk * 64 < n; chunk += waves) {
        const unsigned r = chunk * 64 + lane;
        const bool live = r < n;
        // record r lives in shard s = last shard whose offset <= r
        const unsigned rr = live ? r : n - 1;
        int lo = 0, len = kShards;
        while (len > 1) {
            const int half = len >> 1;
            const bool ge = shard_off[lo + half] <= rr;
            lo = ge ? lo + half : lo;
            len = ge ? len - half : half;
        }
        const uint4 q = rec[(size_t)lo * v.shard_cap + (rr - shard_off[lo])];
        // unpack the bit-packed rank codes into this wave's u16 planes
        for (int p = 0; p < P; ++p) {
            const uint32_t pd = pg.plane_desc[p];          // dword[0:2) | bit_off[2:7) | width[7:11)
            const uint32_t dw = pd & 3;
            const uint32_t word = dw == 0 ? q.x : (dw == 1 ? q.y : q.z);
            planes[p * 64 + lane] = (uint16_t)__builtin_amdgcn_ubfe(word, (pd >> 2) & 31, (pd >> 7) & 15);
        }
        double a0 = 0.0, a1 = 0.0;
        float margin = pg.base;
        int t = 0;
        for (; t + 4 <= T; t += 4) {
            int leaf[4];
            k2_batch<0, 4>(nodes, planes_lane, t, D, NL, leaf);
            if (pg.kind == UGVC_MODEL_RF) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { const double2 pv = pairs[leaf_idx[leaf[k]]]; a0 += pv.x; a1 += pv.y; }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) margin += leaf_f32[leaf[k]];
            }
        }
        for (; t < T; ++t) {
            int leaf[1];
            k2_batch<0, 1>(nodes, planes_lane, t, D, NL, leaf);
            if (pg.kind == UGVC_MODEL_RF) { const double2 pv = pairs[leaf_idx[leaf[0]]]; a0 += pv.x; a1 += pv.y; }
            else margin += leaf_f32[leaf[0]];
        }
        float score;
        uint8_t filt;
        if (pg.kind == UGVC_MODEL_RF) {
            const double p0 = a0 / (double)T, p1 = a1 / (double)T;
            score = (float)p1;
            filt = p1 > p0 ? UGVC_FILTER_PASS : UGVC_FILTER_LOW_SCORE;
        } else {
            score = 1.0f / (1.0f + expf(-margin));
            filt = margin > 0.0f ? UGVC_FILTER_PASS : UGVC_FILTER_LOW_SCORE;
        }
        if (live) {
            v.f.score[q.w] = score;
            v.f.filter[q.w] = filt;
        }
    }
}

static size_t k2_lds_bytes(const PackedGroupView& pg, int n_waves) {
    const size_t NL = (size_t)1 << pg.D, n_nodes = (size_t)pg.T * NL;
    size_t b = (n_nodes * 4 + 15) & ~(size_t)15;
    if (pg.kind == UGVC_MODEL_RF) b += (size_t)pg.n_pairs * 16 + ((n_nodes * 2 + 15) & ~(size_t)15);
    else b += (n_nodes * 4 + 15) & ~(size_t)15;
    return b + (size_t)n_waves * pg.n_planes * 128;
}

int launch_filter_v2(ugvc_ctx* ctx, const FilterArgs& a) {
    if (a.n == 0) return 0;
    V2Args v;
    v.f = a;
    if (v2_fill_args(ctx, v, a.n)) return -1;
    UGVC_HIP(hipMemsetAsync(v.counters, 0, (size_t)UGVC_N_GROUPS * kShards * kCounterStride * 4, ctx->stream));
    const int64_t nbr = (int64_t)(v.n_blocks + 1) * kJoinArrays;
    hipLaunchKernelGGL(bracket_kernel, dim3((unsigned)((nbr + 255) / 256)), dim3(256), 0, ctx->stream, v);
    hipLaunchKernelGGL(featurize_kernel, dim3((unsigned)v.n_blocks), dim3(kBlock), 0, ctx->stream, v);
    // K2 geometry: as many waves per workgroup (16, 12, 8) as the largest forest leaves LDS for
    int n_waves = 0;
    size_t lds = 0;
    for (int w : {16, 12, 8, 4}) {
        size_t need = 0;
        for (int g = 0; g < UGVC_N_GROUPS; ++g)
            if (v.pg[g].ok) need = std::max(need, k2_lds_bytes(v.pg[g], w));
        if (need + 2048 <= 160 * 1024) { n_waves = w; lds = need; break; }
    }
    if (n_waves == 0) return fail("internal: packed forest does not fit LDS");
    if (lds) {
        static bool attr_set = false;
        if (!attr_set) {
            UGVC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(forest_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
            attr_set = true;
        }
        hipLaunchKernelGGL(forest_kernel, dim3((unsigned)ctx->n_cus), dim3(n_waves * 64), lds, ctx->stream, v);
    }
    UGVC_HIP(hipGetLastError());
    return 0;
}

}  // namespace ugvc
