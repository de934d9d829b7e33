// SEC (systematic error correction) database: build + apply on the GPU.
//
// What the reference tree fixes: the statistic - scale_contingency_table, correct_multinomial_frequencies,
// multinomial_likelihood, multinomial_likelihood_ratio (/root/reference/ugvc/utils/stats_utils.py:12-70, known answers
// test/unit/utils/test_stats_utils.py:18-110) - and that a call explained by the cohort's noise is tagged "SEC"
// (ugvc/reports/report_utils.py:71-75).  The tools around it (sec_training, correct_systematic_errors;
// ugvc/__main__.py:19,56) are in the absent submodule and undocumented (README.md:12), so the database layout and
// the decision rule below are BUILDER-DEFINED and say so wherever they surface:
//   * database = sorted unique locus keys (contig << 32 | pos, the blacklist key) + k expected counts per locus, the
//     cohort's summed allele tallies (k = 2: ref, alt; k = 3: ref, alt, other);
//   * build   = concatenated per-sample (key, counts) observations -> stable radix sort by key -> segmented sums
//     (the library's own LSD radix sort + scan, kernels_prims.hip; sums by 64-bit integer atomics);
//   * apply   = per resident variant: exact key lookup (two-level: every 64th key first, then one 64-key block);
//     observed = (ad_ref, ad_alt[, max(dp - ad_ref - ad_alt, 0)]); expected optionally rescaled to the observed depth
//     with scale_contingency_table (round half to even, as numpy); ratio = multinomial_likelihood_ratio(observed,
//     expected)[1]; is_sec = ratio >= min_ratio; NaN / 0 off the database.  With `mark` the SEC bit is OR-ed into the
//     resident flags column, so the gathered callset carries it.

#include <algorithm>
#include <cmath>
#include <stdlib.h>
#include <vector>

#include <type_traits>

#include "ugvc_prims.hpp"

namespace ugvc {

constexpr int kSecMaxK = 8;

constexpr int kSecLgTab = 4096;                                   // log(m!) for m = 0..4096 (host std::lgamma), beyond: lgamma()

// (libm's lgamma / log inline into hundreds of instructions and dozens of registers at every call site: out of line - they
// are reached only by depths beyond the tables)
__device__ __attribute__((noinline)) double sec_lgamma_slow(int m) { return lgamma((double)m + 1.0); }
__device__ __attribute__((noinline)) double sec_log_slow(long long m) { return log((double)m); }

__device__ __forceinline__ double sec_log_fact(const double* __restrict__ tab, int m) {
    return m <= kSecLgTab ? tab[m] : sec_lgamma_slow(m);
}

// log pmf(x; n, p(e)) and log pmf(x; n, p(x)) of stats_utils.py:47-70 share lgamma(n+1) - sum lgamma(x_i+1): every
// argument is an integer, so that part is table look-ups; what differs is sum x_i log p_i
__device__ __forceinline__ void sec_log_pmf2(const int* x, const int* e, int k, const double* __restrict__ tab, double& lp_e, double& lp_x) {
    double tot_e = 0.0, tot_x = 0.0;
    int n = 0;
    for (int i = 0; i < k; ++i) { tot_e += (double)e[i] + 1.0; tot_x += (double)x[i] + 1.0; n += x[i]; }
    double g = sec_log_fact(tab, n), se = 0.0, sx = 0.0;
    for (int i = 0; i < k; ++i) {
        if (x[i] > 0) {
            se += (double)x[i] * log(((double)e[i] + 1.0) / tot_e);
            sx += (double)x[i] * log(((double)x[i] + 1.0) / tot_x);
        }
        g -= sec_log_fact(tab, x[i]);
    }
    lp_e = g + se;
    lp_x = g + sx;
}

__global__ __launch_bounds__(256) void sec_apply_kernel(const uint16_t* __restrict__ contig, const int32_t* __restrict__ pos,
                                                        const int32_t* __restrict__ dp, const int32_t* __restrict__ adr,
                                                        const int32_t* __restrict__ ada, int64_t n,
                                                        const uint64_t* __restrict__ keys, const uint64_t* __restrict__ coarse,
                                                        const int32_t* __restrict__ expected, const double* __restrict__ lg_tab, int64_t n_db, int k,
                                                        double min_ratio, int scale, double* __restrict__ ratio, uint8_t* __restrict__ is_sec,
                                                        uint8_t* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t key = ((uint64_t)contig[i] << 32) | (uint32_t)pos[i];
    // lower bound over the sampled keys, then inside the 64-key block it names
    const int64_t nc = (n_db + 63) >> 6;
    int64_t lo = 0, len = nc;
    while (len > 0) {
        const int64_t half = len >> 1;
        if (coarse[lo + half] <= key) { lo += half + 1; len -= half + 1; }
        else len = half;
    }
    double r = __longlong_as_double(0x7ff8000000000000ll);
    uint8_t hit = 0;
    if (lo > 0) {
        const int64_t b0 = (lo - 1) << 6;
        int64_t l2 = b0, n2 = (n_db - b0 < 64 ? n_db - b0 : 64);
        while (n2 > 0) {
            const int64_t half = n2 >> 1;
            if (keys[l2 + half] < key) { l2 += half + 1; n2 -= half + 1; }
            else n2 = half;
        }
        if (l2 < n_db && keys[l2] == key) {
            int a[kSecMaxK], e[kSecMaxK];
            const int r0 = adr[i] > 0 ? adr[i] : 0, a0 = ada[i] > 0 ? ada[i] : 0;
            a[0] = r0;
            a[1] = a0;
            if (k > 2) { const int o = dp[i] - r0 - a0; a[2] = o > 0 ? o : 0; }
            for (int c = 3; c < k; ++c) a[c] = 0;
            long long s = 0, na = 0;
            for (int c = 0; c < k; ++c) { e[c] = expected[l2 * k + c]; s += e[c]; na += a[c]; }
            if (scale && s > 0) {
                const double f = (double)na / (double)s;              // stats_utils.py:24-27: np.round(table * (n / sum))
                for (int c = 0; c < k; ++c) e[c] = (int)rint((double)e[c] * f);
            }
            double lp_e, lp_x;
            sec_log_pmf2(a, e, k, lg_tab, lp_e, lp_x);
            r = exp(lp_e) / exp(lp_x);
            hit = r >= min_ratio ? 1 : 0;
        }
    }
    if (ratio) ratio[i] = r;
    if (is_sec) is_sec[i] = hit;
    if (flags && hit) flags[i] |= UGVC_FLAG_SEC;
}

// sec_log_pmf2 with log(x_i + 1), log(e_i + 1) and the logs of the two totals from a table of log(m) (every argument is a
// small integer): log((e + 1) / tot) = log(e + 1) - log(tot) up to f64 rounding - eight libm logs per call become loads
__device__ __forceinline__ double sec_log_int(const double* __restrict__ tab, long long m) {
    return m <= kSecLgTab ? tab[kSecLgTab + 1 + m] : sec_log_slow(m);
}

__device__ __forceinline__ void sec_log_pmf2_tab(const int* x, const int* e, int k, const double* __restrict__ tab, double& lp_e, double& lp_x) {
    long long tot_e = 0, tot_x = 0;
    int n = 0;
    for (int i = 0; i < k; ++i) { tot_e += (long long)e[i] + 1; tot_x += (long long)x[i] + 1; n += x[i]; }
    const double lte = sec_log_int(tab, tot_e), ltx = sec_log_int(tab, tot_x);
    double g = sec_log_fact(tab, n), se = 0.0, sx = 0.0;
    for (int i = 0; i < k; ++i) {
        if (x[i] > 0) {
            se += (double)x[i] * (sec_log_int(tab, (long long)e[i] + 1) - lte);
            sx += (double)x[i] * (sec_log_int(tab, (long long)x[i] + 1) - ltx);
        }
        g -= sec_log_fact(tab, x[i]);
    }
    lp_e = g + se;
    lp_x = g + sx;
}

// The same verdicts for a callset SORTED by key (every resident callset is: ugvc_variants_upload checks it).  A wave
// takes consecutive tiles of 64 calls; where a tile's first call stands in the database is where the previous tile's
// last call ended, so the 22-step search per call above becomes, per tile, one coalesced fetch of the next 128 keys
// (requested while the previous tile is worked on) and seven steps over them in LDS.  A fresh search - 64 probes per
// step by the whole wave - happens once per wave; a tile whose calls run past the staged keys searches from the carried
// rank in HBM.  (sec_apply_kernel stays as the checker of this one: UGVC_SEC_SIMPLE=1 selects it.)
constexpr int kSecStage = 128;
constexpr int kSecLdsTab = 2048;                                  // log(m!) and log(m) for m <= 2048 in LDS (2 x 16 KB)

// Round 3: what the round-2 kernel spent its time on was the log tables - twelve 8-byte gathers per hit from a 64 KB
// table in L2 (a 64-lane gather of scattered 8-byte words keeps the texture path busy for ~64 cycles), then two f64 exp
// and an f64 division per hit.  Here the first 2048 entries of both tables sit in LDS (depths beyond that read the
// resident table, beyond 4096 libm), and the verdict is taken in the LOG domain - log ratio = lp_e - lp_x against
// log(min_ratio) - so nothing is exponentiated unless the caller downloads the ratios (WANT_RATIO) or the log ratio lies
// within 1e-9 of the threshold / a pmf underflows (then the round-2 expression decides: identical verdicts in every mode).
__device__ __forceinline__ double sec_lf(const double* lt, const double* __restrict__ tab, int m) {
    return m <= kSecLdsTab ? lt[m] : sec_log_fact(tab, m);
}
__device__ __forceinline__ double sec_li(const double* lt, const double* __restrict__ tab, long long m) {
    return m <= kSecLdsTab ? lt[kSecLdsTab + 1 + m] : sec_log_int(tab, m);
}

// exp(lp_e) / exp(lp_x): two f64 exponentials and a division hold ~60 vector registers - out of line, because the scoring
// mode reaches it only for a log ratio within 1e-9 of the threshold or an underflowing pmf
__device__ __attribute__((noinline)) double sec_exact_ratio(double lp_e, double lp_x) { return exp(lp_e) / exp(lp_x); }

template <bool WANT_RATIO, int KT, int BLOCK>                     // KT: the number of count classes when it is 2, 3 or 4 (loops unroll, counts stay in registers); 0 = any k
__global__ __launch_bounds__(BLOCK) void sec_apply_tiles_kernel(const uint16_t* __restrict__ contig, const int32_t* __restrict__ pos,
                                                                    const int32_t* __restrict__ dp, const int32_t* __restrict__ adr,
                                                                    const int32_t* __restrict__ ada, int64_t n,
                                                                    const uint64_t* __restrict__ keys, const int32_t* __restrict__ expected,
                                                                    const double* __restrict__ lg_tab, int64_t n_db, int k, double min_ratio,
                                                                    double log_min, int scale, double* __restrict__ ratio,
                                                                    uint8_t* __restrict__ is_sec, uint8_t* __restrict__ flags, int tiles_per_wave) {
    __shared__ double lt[2 * (kSecLdsTab + 1)];
    __shared__ uint64_t stage_all[BLOCK / 64][kSecStage];
    __shared__ uint2 queue_all[BLOCK / 64][128];
    for (int q = threadIdx.x; q <= kSecLdsTab; q += BLOCK) {
        lt[q] = lg_tab[q];
        lt[kSecLdsTab + 1 + q] = lg_tab[kSecLgTab + 1 + q];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t* stage = stage_all[wave];
    const int64_t n_tiles = (n + 63) >> 6;
    const int64_t t0 = ((int64_t)blockIdx.x * (BLOCK / 64) + wave) * tiles_per_wave;
    const int64_t t1 = t0 + tiles_per_wave < n_tiles ? t0 + tiles_per_wave : n_tiles;
    if (t0 >= t1) return;
    auto key_of = [&](int64_t i) { return ((uint64_t)contig[i] << 32) | (uint32_t)pos[i]; };
    // where the wave's first call stands: lower bound by 64 probes per step
    int64_t Lb = 0;
    {
        const uint64_t key0 = key_of(t0 * 64);
        int64_t b = 0, len = n_db;
        while (len > 0) {
            const int64_t step = (len + 63) >> 6;
            uint64_t x = ~0ull;
            if (lane * step < len) x = keys[b + lane * step];
            const int kk = (int)__popcll(__ballot(x < key0));
            const int64_t end = b + len;
            const int64_t nb = kk > 0 ? b + (kk - 1) * step + 1 : b;
            int64_t ne = b + kk * step;
            ne = ne < end ? ne : end;
            b = nb;
            len = ne > nb ? ne - nb : 0;
        }
        Lb = b;
    }
    auto fetch = [&](int64_t from, uint64_t& a0, uint64_t& a1) {
        a0 = from + lane < n_db ? keys[from + lane] : ~0ull;
        a1 = from + 64 + lane < n_db ? keys[from + 64 + lane] : ~0ull;
    };
    uint2* queue = queue_all[wave];
    int qn = 0;                                                      // hits waiting in the wave's queue (< 64 between tiles)
    // one hit: observed counts of call q.x against the expected counts of database row q.y
    auto work = [&](uint2 q) {
        const int kk = KT ? KT : k;
        const int64_t ci = (int64_t)q.x, l2 = (int64_t)q.y;
        int a[kSecMaxK], e[kSecMaxK];
        const int r0 = adr[ci] > 0 ? adr[ci] : 0, a0 = ada[ci] > 0 ? ada[ci] : 0;
        a[0] = r0;
        a[1] = a0;
        if (kk > 2) { const int o = dp[ci] - r0 - a0; a[2] = o > 0 ? o : 0; }
        for (int c = 3; c < kk; ++c) a[c] = 0;
        long long s = 0, na = 0;
        for (int c = 0; c < kk; ++c) { e[c] = expected[l2 * kk + c]; s += e[c]; na += a[c]; }
        if (scale && s > 0) {
            const double f = (double)na / (double)s;              // stats_utils.py:24-27: np.round(table * (n / sum))
            for (int c = 0; c < kk; ++c) e[c] = (int)rint((double)e[c] * f);
        }
        // sec_log_pmf2_tab with the LDS tables
        long long tot_e = 0, tot_x = 0;
        int nn = 0;
        for (int c = 0; c < kk; ++c) { tot_e += (long long)e[c] + 1; tot_x += (long long)a[c] + 1; nn += a[c]; }
        const double lte = sec_li(lt, lg_tab, tot_e), ltx = sec_li(lt, lg_tab, tot_x);
        double g = sec_lf(lt, lg_tab, nn), se = 0.0, sx = 0.0;
        for (int c = 0; c < kk; ++c) {
            if (a[c] > 0) {
                se += (double)a[c] * (sec_li(lt, lg_tab, (long long)e[c] + 1) - lte);
                sx += (double)a[c] * (sec_li(lt, lg_tab, (long long)a[c] + 1) - ltx);
            }
            g -= sec_lf(lt, lg_tab, a[c]);
        }
        const double lp_e = g + se, lp_x = g + sx, d = lp_e - lp_x;
        const bool exact = WANT_RATIO || !(min_ratio > 0.0) || fabs(d - log_min) <= 1e-9 * fmax(1.0, fabs(log_min)) || lp_e < -700.0 || lp_x < -700.0;
        double r = 0.0;
        uint8_t hit;
        if (exact) {
            r = WANT_RATIO ? exp(lp_e) / exp(lp_x) : sec_exact_ratio(lp_e, lp_x);
            hit = r >= min_ratio ? 1 : 0;
        } else hit = d >= log_min ? 1 : 0;
        if (WANT_RATIO && ratio) ratio[ci] = r;
        if (is_sec) is_sec[ci] = hit;
        if (flags && hit) flags[ci] |= UGVC_FLAG_SEC;
    };
    uint64_t s0, s1;
    fetch(Lb, s0, s1);
    // ONE call site of work(): the loop runs on past the last tile until the queue is empty (two inlined copies of the f64
    // arithmetic - one in the loop, one for the flush - cost 100 spilled vector registers at the 128-register cap)
    for (int64_t t = t0; t < t1 || qn > 0; ++t) {
        if (t < t1) {
            const int64_t i = t * 64 + lane;
            const bool valid = i < n;
            const int64_t ii = valid ? i : n - 1;
            const uint64_t key = key_of(ii);
            const int last = (int)((n - t * 64 < 64 ? n - t * 64 : 64) - 1);
            stage[lane] = s0;
            stage[64 + lane] = s1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint64_t key_max = __shfl(key, last);
            int64_t l2;
            bool found;
            if (stage[kSecStage - 1] >= key_max) {                    // the staged keys reach the tile's last call
                int p = -1;
#pragma unroll
                for (int sb = 64; sb >= 1; sb >>= 1) p = stage[p + sb] < key ? p + sb : p;
                const int r = p + 1;                                   // staged keys below this call's key
                l2 = Lb + r;
                found = stage[r] == key;
            } else {
                int64_t b = Lb, len = n_db - Lb;
                while (len > 0) {
                    const int64_t half = len >> 1;
                    if (keys[b + half] < key) { b += half + 1; len -= half + 1; }
                    else len = half;
                }
                l2 = b;
                found = l2 < n_db && keys[l2] == key;
            }
            __builtin_amdgcn_wave_barrier();
            Lb = __shfl(l2, last);
            if (t + 1 < t1) fetch(Lb, s0, s1);                          // in flight during this tile's arithmetic
            // calls off the database: their outputs now; hits go to the wave's queue and are worked on 64 at a time - ~40 % of
            // the lanes hit in every tile, so the f64 arithmetic below used to run for every wave with 60 % of its lanes idle
            const bool go = found && valid;
            if (valid && !go) {
                if (WANT_RATIO && ratio) ratio[i] = __longlong_as_double(0x7ff8000000000000ll);
                if (is_sec) is_sec[i] = 0;
            }
            const unsigned long long gm = __builtin_amdgcn_ballot_w64(go);
            if (go) queue[qn + (int)__popcll(gm & ((1ull << lane) - 1))] = make_uint2((uint32_t)i, (uint32_t)l2);
            qn += (int)__popcll(gm);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (qn >= 64 || (t + 1 >= t1 && qn > 0)) {
            const int take = qn < 64 ? qn : 64;
            if (lane < take) work(queue[lane]);
            const uint2 rest = queue[64 + lane];
            __builtin_amdgcn_wave_barrier();
            queue[lane] = rest;
            qn -= take;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

__global__ void sec_iota_kernel(uint32_t* idx, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idx[i] = (uint32_t)i;
}

// sorted observations -> loci: head[i] = 1 where a new key begins (a 64-bit word: the segment index is its scan)
__global__ void sec_heads_kernel(const uint64_t* __restrict__ keys, int64_t n, uint64_t* __restrict__ head) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1ull : 0ull;
}

// seg[i] = inclusive scan of the heads: observation i belongs to locus seg[i] - 1.  The locus' key is written by its head,
// its k counts are summed with 64-bit integer atomics (integer sums commute: the result does not depend on the order).
__global__ void sec_segment_sum_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ idx, const uint64_t* __restrict__ seg,
                                       const int32_t* __restrict__ counts, int64_t n, int k, uint64_t* __restrict__ out_keys,
                                       unsigned long long* __restrict__ sums) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t s = seg[i] - 1;
    if (i == 0 || keys[i] != keys[i - 1]) out_keys[s] = keys[i];
    const int32_t* row = counts + (int64_t)idx[i] * k;
    for (int c = 0; c < k; ++c)
        if (row[c]) atomicAdd(&sums[s * k + c], (unsigned long long)row[c]);
}

__global__ void sec_narrow_kernel(const unsigned long long* __restrict__ sums, int64_t n_words, int32_t* __restrict__ out, int* __restrict__ overflow) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_words) return;
    const unsigned long long v = sums[i];
    if (v > 2147483647ull) { *overflow = 1; out[i] = 2147483647; }
    else out[i] = (int32_t)v;
}

}  // namespace ugvc

using namespace ugvc;

extern "C" {

int ugvc_sec_db_build(ugvc_ctx* ctx, const uint64_t* keys, const int32_t* counts, int64_t n_obs, int k, uint64_t* out_keys,
                      int32_t* out_expected, int64_t* out_n) {
    if (!ctx || !out_n || (n_obs > 0 && (!keys || !counts || !out_keys || !out_expected))) return fail("NULL argument");
    if (k < 2 || k > kSecMaxK) return fail("k must be in 2..8");
    if (n_obs < 0 || n_obs >= ((int64_t)1 << 31)) return fail("n_obs out of range");
    *out_n = 0;
    if (n_obs == 0) return 0;
    for (int64_t i = 0; i < n_obs * k; ++i)
        if (counts[i] < 0) return fail("counts must be non-negative");
    UGVC_HIP(hipSetDevice(ctx->device));
    DeviceBuf d_k0, d_k1, d_i0, d_i1, d_cnt, d_seg, d_uk, d_sum, d_out, d_ovf, d_tmp;
    DeviceBuf* all[] = {&d_k0, &d_k1, &d_i0, &d_i1, &d_cnt, &d_seg, &d_uk, &d_sum, &d_out, &d_ovf, &d_tmp};
    int rc = 0;
    const size_t N = (size_t)n_obs;
    const unsigned grid = (unsigned)((n_obs + 255) / 256);
    int64_t n_unique = 0;
    do {
        if ((rc = upload(ctx, d_k0, keys, N * 8)) || (rc = upload(ctx, d_cnt, counts, N * k * 4))) break;
        if ((rc = ensure(d_k1, N * 8)) || (rc = ensure(d_i0, N * 4)) || (rc = ensure(d_i1, N * 4)) || (rc = ensure(d_seg, N * 8)) ||
            (rc = ensure(d_uk, N * 8)) || (rc = ensure(d_sum, N * k * 8)) || (rc = ensure(d_out, N * k * 4)) || (rc = ensure(d_ovf, 4))) break;
        if (hipMemsetAsync(d_ovf.p, 0, 4, ctx->stream) != hipSuccess || hipMemsetAsync(d_sum.p, 0, N * k * 8, ctx->stream) != hipSuccess) {
            rc = fail("hipMemsetAsync failed");
            break;
        }
        UGVC_LAUNCH(sec_iota_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_i0.as<uint32_t>(), n_obs);
        uint64_t* ks = nullptr;
        uint32_t* is = nullptr;
        if ((rc = radix_sort_pairs_u64(ctx, d_tmp, d_k0.as<uint64_t>(), d_k1.as<uint64_t>(), d_i0.as<uint32_t>(), d_i1.as<uint32_t>(), n_obs, &ks, &is))) break;
        UGVC_LAUNCH(sec_heads_kernel, dim3(grid), dim3(256), 0, ctx->stream, ks, n_obs, d_seg.as<uint64_t>());
        if ((rc = scan_u64(ctx, d_tmp, d_seg.as<uint64_t>(), n_obs, true))) break;
        uint64_t last = 0;
        if (copy_out(ctx, &last, d_seg.as<uint64_t>() + (N - 1), 8) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = fail("sec_db_build: device error"); break; }
        n_unique = (int64_t)last;
        UGVC_LAUNCH(sec_segment_sum_kernel, dim3(grid), dim3(256), 0, ctx->stream, ks, is, d_seg.as<uint64_t>(), d_cnt.as<int32_t>(), n_obs, k,
                           d_uk.as<uint64_t>(), d_sum.as<unsigned long long>());
        UGVC_LAUNCH(sec_narrow_kernel, dim3((unsigned)((n_unique * k + 255) / 256)), dim3(256), 0, ctx->stream, d_sum.as<unsigned long long>(),
                           n_unique * k, d_out.as<int32_t>(), d_ovf.as<int>());
        int ovf = 0;
        if (copy_out(ctx, out_keys, d_uk.p, (size_t)n_unique * 8) != hipSuccess ||
            copy_out(ctx, out_expected, d_out.p, (size_t)n_unique * k * 4) != hipSuccess ||
            copy_out(ctx, &ovf, d_ovf.p, 4) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = fail("sec_db_build: device error"); break; }
        if (ovf) { rc = fail("a summed count exceeds int32"); break; }
        *out_n = n_unique;
    } while (0);
    for (DeviceBuf* b : all) if (b->p) dev_free(b->p);
    return rc;
}

int ugvc_sec_db_upload(ugvc_ctx* ctx, const uint64_t* keys, const int32_t* expected, int64_t n_db, int k) {
    if (!ctx) return fail("ctx is NULL");
    if (n_db < 0 || (n_db > 0 && (!keys || !expected))) return fail("bad SEC database arguments");
    if (k < 2 || k > kSecMaxK) return fail("k must be in 2..8");
    for (int64_t i = 1; i < n_db; ++i)
        if (keys[i] <= keys[i - 1]) return fail("SEC database keys must be sorted and unique");
    for (int64_t i = 0; i < n_db * k; ++i)
        if (expected[i] < 0) return fail("expected counts must be non-negative");
    UGVC_HIP(hipSetDevice(ctx->device));
    if (upload(ctx, ctx->sec_keys, keys, (size_t)n_db * 8) || upload(ctx, ctx->sec_exp, expected, (size_t)n_db * k * 4)) return -1;
    std::vector<uint64_t> c((size_t)((n_db + 63) / 64));
    for (size_t j = 0; j < c.size(); ++j) c[j] = keys[j * 64];
    if (upload(ctx, ctx->sec_coarse, c.data(), c.size() * 8)) return -1;
    std::vector<double> lg(2 * ((size_t)kSecLgTab + 1));                              // log(m!) | log(m), m = 0..kSecLgTab
    for (int m = 0; m <= kSecLgTab; ++m) {
        lg[(size_t)m] = std::lgamma((double)m + 1.0);
        lg[(size_t)kSecLgTab + 1 + m] = m > 0 ? std::log((double)m) : 0.0;
    }
    if (upload(ctx, ctx->sec_lgtab, lg.data(), lg.size() * 8)) return -1;
    UGVC_HIP(hipStreamSynchronize(ctx->stream));
    ctx->n_sec = n_db;
    ctx->sec_k = k;
    return 0;
}

int ugvc_sec_apply(ugvc_ctx* ctx, double min_ratio, int scale_expected, int mark, double* ratio, uint8_t* is_sec) {
    if (!ctx) return fail("ctx is NULL");
    if (!(min_ratio >= 0.0)) return fail("min_ratio must be >= 0");
    if (!ctx->sec_k) return fail("no SEC database uploaded (ugvc_sec_db_upload)");
    const int64_t n = ctx->n;
    if (n == 0) return 0;
    if (!ctx->v_pos.p) return fail("no variants resident (ugvc_variants_upload)");
    if (mark && !(ctx->scored && ctx->r_flags.p)) return fail("mark needs the resident flags column of a scoring pass (ugvc_filter_resident)");
    UGVC_HIP(hipSetDevice(ctx->device));
    DeviceBuf d_r, d_s;
    int rc = 0;
    do {
        if ((ratio && (rc = ensure(d_r, (size_t)n * 8))) || (is_sec && (rc = ensure(d_s, (size_t)n)))) break;
        static const bool simple_env = getenv("UGVC_SEC_SIMPLE") != nullptr;
        const bool simple = simple_env || n >= ((int64_t)1 << 32) || ctx->n_sec >= ((int64_t)1 << 32);   // (the hit queue holds 32-bit row numbers)
        if (simple) {
            UGVC_LAUNCH(sec_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->v_contig.as<uint16_t>(),
                               ctx->v_pos.as<int32_t>(), ctx->v_dp.as<int32_t>(), ctx->v_adr.as<int32_t>(), ctx->v_ada.as<int32_t>(), n,
                               ctx->sec_keys.as<uint64_t>(), ctx->sec_coarse.as<uint64_t>(), ctx->sec_exp.as<int32_t>(), ctx->sec_lgtab.as<double>(), ctx->n_sec,
                               ctx->sec_k,
                               min_ratio, scale_expected, ratio ? d_r.as<double>() : nullptr, is_sec ? d_s.as<uint8_t>() : nullptr,
                               mark ? ctx->r_flags.as<uint8_t>() : nullptr);
        } else {
            // consecutive tiles per wave: enough waves for 8 per SIMD on every CU
            const int64_t n_tiles = (n + 63) / 64;
            // workgroups of 512 threads: three per CU by LDS (49 KB each) and by registers (<= 80), and as many waves as are
            // resident at once - one round, every wave the same number of tiles (UGVC_SEC_BLOCK / UGVC_SEC_WAVES: profiling)
            const int blk = getenv("UGVC_SEC_BLOCK") ? atoi(getenv("UGVC_SEC_BLOCK")) : 512;
            const int wpc = getenv("UGVC_SEC_WAVES") ? atoi(getenv("UGVC_SEC_WAVES")) : 24;
            const int block = blk == 1024 ? 1024 : blk == 256 ? 256 : 512;
            const int64_t want_waves = (int64_t)ctx->n_cus * std::max(wpc, 1);
            const int tpw = (int)std::max<int64_t>((n_tiles + want_waves - 1) / want_waves, 1);
            const int64_t n_waves = (n_tiles + tpw - 1) / tpw;
            const unsigned grid = (unsigned)((n_waves + block / 64 - 1) / (block / 64));
            const double log_min = min_ratio > 0.0 ? std::log(min_ratio) : -1e300;
            auto pick_k = [&](auto want, auto bl) {
                constexpr bool W = decltype(want)::value;
                constexpr int B = decltype(bl)::value;
                return ctx->sec_k == 2 ? sec_apply_tiles_kernel<W, 2, B> : ctx->sec_k == 3 ? sec_apply_tiles_kernel<W, 3, B>
                     : ctx->sec_k == 4 ? sec_apply_tiles_kernel<W, 4, B> : sec_apply_tiles_kernel<W, 0, B>;
            };
            auto pick = [&](auto want) {
                return block == 1024 ? pick_k(want, std::integral_constant<int, 1024>{})
                     : block == 256 ? pick_k(want, std::integral_constant<int, 256>{}) : pick_k(want, std::integral_constant<int, 512>{});
            };
            auto kern = ratio ? pick(std::true_type{}) : pick(std::false_type{});
            UGVC_LAUNCH(kern, dim3(grid), dim3(block), 0, ctx->stream, ctx->v_contig.as<uint16_t>(),
                               ctx->v_pos.as<int32_t>(), ctx->v_dp.as<int32_t>(), ctx->v_adr.as<int32_t>(), ctx->v_ada.as<int32_t>(), n,
                               ctx->sec_keys.as<uint64_t>(), ctx->sec_exp.as<int32_t>(), ctx->sec_lgtab.as<double>(), ctx->n_sec, ctx->sec_k,
                               min_ratio, log_min, scale_expected, ratio ? d_r.as<double>() : nullptr, is_sec ? d_s.as<uint8_t>() : nullptr,
                               mark ? ctx->r_flags.as<uint8_t>() : nullptr, tpw);
        }
        if (hipGetLastError() != hipSuccess) { rc = fail("sec_apply: launch failed"); break; }
        // (mark-only calls stay stream-ordered: nothing to wait for on the host)
        if ((ratio && copy_out(ctx, ratio, d_r.p, (size_t)n * 8) != hipSuccess) ||
            (is_sec && copy_out(ctx, is_sec, d_s.p, (size_t)n) != hipSuccess) ||
            ((ratio || is_sec) && hipStreamSynchronize(ctx->stream) != hipSuccess)) { rc = fail("sec_apply: device error"); break; }
    } while (0);
    for (DeviceBuf* b : {&d_r, &d_s}) if (b->p) dev_free(b->p);
    return rc;
}

// `iters` mark-only applications back to back between two events on the context stream (what bench.py --workload sec_apply
// reports: kernel time without the host's launch latency)
int ugvc_timed_sec_apply(ugvc_ctx* ctx, double min_ratio, int scale_expected, int iters, float* ms_total) {
    if (!ctx || !ms_total || iters < 1) return fail("bad arguments");
    UGVC_HIP(hipSetDevice(ctx->device));
    UGVC_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    for (int it = 0; it < iters; ++it)
        if (ugvc_sec_apply(ctx, min_ratio, scale_expected, 1, nullptr, nullptr)) return -1;
    UGVC_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    UGVC_HIP(hipEventSynchronize(ctx->ev1));
    UGVC_HIP(hipEventElapsedTime(ms_total, ctx->ev0, ctx->ev1));
    return 0;
}

}  // extern "C"
