// Auxiliary hot-path kernels: pileup tally (a11), SEC multinomial likelihood ratio (a8),
// bridging-homopolymer SNV un-filter (a12).  gfx950 only.
#include <math.h>
#include <string.h>

#include <algorithm>

#include "ugvc_device.hpp"

namespace ugvc {

// ------------------------------------------------------------------------------------------
// Pileup allele/strand/base-quality tally.  BUILDER-DEFINED: the
// reference only consumes FORMAT/AD, DP, SB, VAF and INFO/SOR
// (test/resources/unit/vcfbed/test_vcftools/header.txt:3379,3391-3398; uses at
// ugvc/pipelines/vcfbed/calibrate_bridging_snvs.py:115-117).
//
// A workgroup owns 256 consecutive loci.  Their observations are one contiguous span of the CSR
// array: it is staged into LDS with coalesced 16-byte loads (the "LDS-staged per-locus read
// window"), then every lane walks its own locus.  An observation costs one ds_read_u16 and ~15
// VALU: the (allele, strand) class is counted one-hot in 8-bit fields (eight classes in two
// dwords, flushed to wide counters every 255 observations), base-quality sums in 32 bits.
// Loci deeper than kPlDeep are tallied by the whole workgroup instead (strided lanes, then shuffle
// reductions of the ten integer counters - exact).
// Measured and dropped this round (5 M loci / 150 M observations, 136 us as it stands): equal runs of observations per
// lane with LDS counters per locus (325 us: some lane crosses a locus boundary in almost every trip, so the wave runs the
// flush path every trip); lanes dealt to loci in order of depth (152 us: a workgroup still ends with its deepest wave);
// the counter walk unrolled by four (149 us); SOR with one f64 division instead of four (no change).
// Measured on 5 M loci / 150 M observations: the previous one-wave-per-64-observations kernel
// (per-observation binary search + 12 shuffles per chunk) took 773 us.
constexpr int kPlBlock = 256;
constexpr int kPlLociPerBlock = 256;
constexpr int kPlCap = 12288;          // staged observations per workgroup, at most (24 KB); the launch picks 8192 / 10240 / 12288
                                       // by the longest 256-locus span of the table (fewer LDS bytes = more workgroups per CU)
constexpr int kPlDeep = 2048;

struct PileupArgs {
    int64_t n_loci;
    const int64_t* off;
    const uint32_t* off32;     // compact layout: 4-byte offsets (fewer than 2^32 observations)
    const uint16_t* obs;
    int32_t* ref_fwd; int32_t* ref_rev; int32_t* alt_fwd; int32_t* alt_rev;
    int32_t* other; int32_t* dp; int32_t* bq_ref; int32_t* bq_alt;
    float* vaf; float* sor;
    uint16_t* c16[5];          // compact layout: the four strand counts and `other` as u16 (no locus deeper than 65535)
};
// Device-side layouts.  Wide: ten 4-byte columns (ref/alt x fwd/rev, other, dp, bq sums, vaf, sor) and 8-byte offsets:
// 108 bytes of traffic per locus at 30 observations.  Compact - the layout SURVEY.md 8(d) prices at 84 B/locus: 4-byte
// offsets, four u16 strand counts + `other` + two u32 base-quality sums + f32 SOR = 22 bytes out; dp (an offset
// difference) and vaf are derived on the host when the columns are downloaded.  Chosen at upload:
// compact whenever no locus is deeper than 65535 and the table holds fewer than 2^32 observations.

__device__ __forceinline__ float sor_from_table(int rf, int rr, int af, int ar) {
    // GATK StrandOddsRatio on the +1 table (oracle.pileup_tally): ln(R + 1/R) + ln(min(a,b)/max(a,b)) - ln(min(c,d)/max(c,d))
    // folded into ONE log.  Round 3: in f32 - the column is f32 and the f64 log + three f64 divisions were ~40 % of a
    // lane's instructions at depth 30; five f32 roundings in front of the log move it by <= ~4e-7 absolute (test: 1e-5).
    const float a = (float)rf + 1.0f, b = (float)rr + 1.0f, c = (float)af + 1.0f, d = (float)ar + 1.0f;
    const float R = __fdiv_rn(a * d, b * c);
    return logf((R + __fdiv_rn(1.0f, R)) * __fdiv_rn(fminf(a, b), fmaxf(a, b)) * __fdiv_rn(fmaxf(c, d), fminf(c, d)));
}

struct PlAcc {                 // wide per-locus counters
    int cf[4], cr[4];          // class counts by allele code 0..3, forward / reverse strand
    int bq0, bq1;
};

// tally observations [k0, k1) of `src` (LDS or HBM) into acc; 8-bit one-hot fields, flushed every 255
template <class Idx, class Src>
__device__ __forceinline__ void pl_walk(Src src, Idx k0, Idx k1, Idx step, PlAcc& acc) {
    Idx k = k0;
    while (k < k1) {
        uint32_t f8 = 0, r8 = 0;                              // four 8-bit class counters each
        const Idx stop = k + 255 * step < k1 ? k + 255 * step : k1;
        // (four reads per trip, issued together, measured slower: 149 vs 141 us per 5 M loci - the walk is bound by
        // instruction issue at 24 waves per CU, not by the LDS round trip)
        for (; k < stop; k += step) {
            const uint32_t o = src(k);
            const uint32_t inc = 1u << ((o & 3u) << 3);
            const bool rev = (o & 4u) != 0;
            f8 += rev ? 0u : inc;
            r8 += rev ? inc : 0u;
            const int bq = (int)(o >> 3);
            acc.bq0 += (o & 3u) == 0 ? bq : 0;
            acc.bq1 += (o & 3u) == 1 ? bq : 0;
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) { acc.cf[a] += (f8 >> (8 * a)) & 0xff; acc.cr[a] += (r8 >> (8 * a)) & 0xff; }
    }
}

// The staged walk of one lane's locus: observations [k0, k1) of the workgroup's LDS span, two per trip (one aligned
// dword read), the (allele, strand) class counted one-hot in ONE 64-bit word of eight 8-bit fields (a 64-bit shift
// and add instead of two selected 32-bit adds), flushed every 254 observations.
__device__ __forceinline__ void pl_walk_staged(const uint16_t* stage, int k0, int k1, PlAcc& acc) {
    // one observation given its class (allele | strand << 2) and base quality.  Round 3: the base-quality sums take their
    // 0 / 1 factors from the one-hot word itself (bit 0 of lo | hi <=> allele 0, bit 8 <=> allele 1) and a multiply-add
    // each, instead of two compares + two selects + two adds; the pair of a dword is taken apart with bit-field
    // extracts instead of being split first (~11 vector instructions an observation, 14 before).
    auto one = [&](uint32_t cls, uint32_t bq, unsigned long long& c8) {
        const unsigned long long t = 1ull << (cls << 3);
        c8 += t;
        const uint32_t u = (uint32_t)t | (uint32_t)(t >> 32);
        // (written as instructions: the compiler turns a multiplication by a 0 / 1 value back into compare + select; the 0 / 1
        // factors are bytes 0 and 1 of `u`, picked by the multiplier's operand selector - no extract instructions)
        uint32_t x0, x1;
        asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(x0) : "v"(bq), "v"(u));
        asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(x1) : "v"(bq), "v"(u));
        acc.bq0 += (int)x0;                                                                  // bq < 2^13
        acc.bq1 += (int)x1;
    };
    int k = k0;
    while (k < k1) {
        unsigned long long c8 = 0;
        const int stop = k + 254 < k1 ? k + 254 : k1;
        if ((k & 1) && k < stop) { const uint32_t o = stage[k]; one(o & 7u, o >> 3, c8); ++k; }
        {   // (one induction variable - the LDS address - instead of an index, a bound test on index + 2 and the address)
            const uint32_t* p = reinterpret_cast<const uint32_t*>(stage + k);
            const int np = (stop - k) >> 1;
            const uint32_t* pe = p + np;
            for (; p != pe; ++p) {
                const uint32_t w = *p;
                one(w & 7u, __builtin_amdgcn_ubfe(w, 3, 13), c8);
                one(__builtin_amdgcn_ubfe(w, 16, 3), w >> 19, c8);
            }
            k += 2 * np;
        }
        if (k < stop) { const uint32_t o = stage[k]; one(o & 7u, o >> 3, c8); ++k; }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            acc.cf[a] += (int)((c8 >> (8 * a)) & 0xff);
            acc.cr[a] += (int)((c8 >> (8 * (a + 4))) & 0xff);
        }
    }
}

template <bool COMPACT>
__device__ __forceinline__ void pl_store(const PileupArgs& a, int64_t l, const PlAcc& c, int dp) {
    const int rf = c.cf[0], rr = c.cr[0], af = c.cf[1], ar = c.cr[1];
    if (COMPACT) {
        a.c16[0][l] = (uint16_t)rf; a.c16[1][l] = (uint16_t)rr; a.c16[2][l] = (uint16_t)af; a.c16[3][l] = (uint16_t)ar;
        a.c16[4][l] = (uint16_t)(c.cf[2] + c.cr[2]);
    } else {
        a.ref_fwd[l] = rf; a.ref_rev[l] = rr; a.alt_fwd[l] = af; a.alt_rev[l] = ar;
        a.other[l] = c.cf[2] + c.cr[2];
        a.dp[l] = dp;
        a.vaf[l] = dp > 0 ? __fdiv_rn((float)(af + ar), (float)dp) : 0.0f;
    }
    a.bq_ref[l] = c.bq0; a.bq_alt[l] = c.bq1;
    a.sor[l] = sor_from_table(rf, rr, af, ar);
}

template <bool COMPACT, int CAP>
__global__ __launch_bounds__(kPlBlock) void pileup_kernel(const PileupArgs a) {
    auto off_at = [&](int64_t l) -> int64_t { return COMPACT ? (int64_t)a.off32[l] : a.off[l]; };
    __shared__ __attribute__((aligned(16))) uint16_t stage[CAP + 8];
    __shared__ int red[10];
    const int tid = threadIdx.x;
    const int64_t l0 = (int64_t)blockIdx.x * kPlLociPerBlock;
    const int nl = (int)((a.n_loci - l0) < kPlLociPerBlock ? (a.n_loci - l0) : kPlLociPerBlock);
    const int64_t o0 = off_at(l0), o1 = off_at(l0 + nl);
    const int64_t base = o0 & ~(int64_t)7;                    // 16-byte aligned start of the staged span
    const bool staged = o1 - base <= CAP;
    if (staged) {
        const uint4* src = reinterpret_cast<const uint4*>(a.obs + base);      // obs buffer is padded by 16 bytes
        uint4* dst = reinterpret_cast<uint4*>(stage);
        const int n16 = (int)((o1 - base + 7) >> 3);
        for (int k = tid; k < n16; k += kPlBlock) dst[k] = src[k];
    }
    __syncthreads();
    const bool mine = tid < nl;
    const int64_t s = mine ? off_at(l0 + tid) : 0, e = mine ? off_at(l0 + tid + 1) : 0;
    const bool deep = mine && (e - s) > kPlDeep;
    if (mine && !deep) {
        PlAcc acc = {{0, 0, 0, 0}, {0, 0, 0, 0}, 0, 0};
        if (staged) pl_walk_staged(stage, (int)(s - base), (int)(e - base), acc);
        else pl_walk<int64_t>([&](int64_t k) -> uint32_t { return a.obs[k]; }, s, e, (int64_t)1, acc);
        pl_store<COMPACT>(a, l0 + tid, acc, (int)(e - s));
    }
    // deep loci: one after the other, the whole workgroup strides over the observations
    unsigned long long dm = __ballot(deep);
    __shared__ unsigned long long deep_mask[kPlBlock / 64];
    if ((tid & 63) == 0) deep_mask[tid >> 6] = dm;
    __syncthreads();
    for (int w = 0; w < kPlBlock / 64; ++w) {
        unsigned long long m = deep_mask[w];
        while (m) {
            const int j = w * 64 + __ffsll((long long)m) - 1;
            m &= m - 1;
            const int64_t ds = off_at(l0 + j), de = off_at(l0 + j + 1);
            if (tid < 10) red[tid] = 0;
            __syncthreads();
            PlAcc acc = {{0, 0, 0, 0}, {0, 0, 0, 0}, 0, 0};
            pl_walk<int64_t>([&](int64_t k) -> uint32_t { return a.obs[k]; }, ds + tid, de, (int64_t)kPlBlock, acc);
            int v[10] = {acc.cf[0], acc.cr[0], acc.cf[1], acc.cr[1], acc.cf[2], acc.cr[2], acc.cf[3], acc.cr[3], acc.bq0, acc.bq1};
#pragma unroll
            for (int q = 0; q < 10; ++q) {
                int x = v[q];
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
                if ((tid & 63) == 0 && x) atomicAdd(&red[q], x);
            }
            __syncthreads();
            if (tid == 0) {
                PlAcc t = {{red[0], red[2], red[4], red[6]}, {red[1], red[3], red[5], red[7]}, red[8], red[9]};
                pl_store<COMPACT>(a, l0 + j, t, (int)(de - ds));
            }
            __syncthreads();
        }
    }
}

int launch_pileup(ugvc_ctx* ctx) {
    if (ctx->pl_n == 0) return 0;
    PileupArgs a;
    memset(&a, 0, sizeof(a));
    a.n_loci = ctx->pl_n;
    a.off = ctx->pl_off.as<int64_t>();
    a.off32 = ctx->pl_off32.as<uint32_t>();
    a.obs = ctx->pl_obsb.as<uint16_t>();
    int32_t* o = ctx->pl_out.as<int32_t>();
    const int64_t n = ctx->pl_n;
    const unsigned grid = (unsigned)((n + kPlLociPerBlock - 1) / kPlLociPerBlock);
    const char* cap_env = getenv("UGVC_PL_CAP");                          // (profiling: staging capacity 8192 / 10240 / 12288)
    if (ctx->pl_compact) {
        // [bq_ref u32 | bq_alt u32 | sor f32 | five u16 count columns]: 22 bytes per locus
        a.bq_ref = o; a.bq_alt = o + n;
        a.sor = reinterpret_cast<float*>(o + 2 * n);
        uint16_t* h = reinterpret_cast<uint16_t*>(o + 3 * n);
        for (int q = 0; q < 5; ++q) a.c16[q] = h + (size_t)q * n;
        const int cap = cap_env ? atoi(cap_env) : (ctx->pl_span <= 8192 ? 8192 : ctx->pl_span <= 10240 ? 10240 : kPlCap);
        auto kern = cap <= 8192 ? pileup_kernel<true, 8192> : cap <= 10240 ? pileup_kernel<true, 10240> : pileup_kernel<true, kPlCap>;
        UGVC_LAUNCH(kern, dim3(grid), dim3(kPlBlock), 0, ctx->stream, a);
    } else {
        a.ref_fwd = o; a.ref_rev = o + n; a.alt_fwd = o + 2 * n; a.alt_rev = o + 3 * n;
        a.other = o + 4 * n; a.dp = o + 5 * n; a.bq_ref = o + 6 * n; a.bq_alt = o + 7 * n;
        a.vaf = reinterpret_cast<float*>(o + 8 * n);
        a.sor = reinterpret_cast<float*>(o + 9 * n);
        UGVC_LAUNCH((pileup_kernel<false, kPlCap>), dim3(grid), dim3(kPlBlock), 0, ctx->stream, a);
    }
    UGVC_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// SEC statistic: multinomial_likelihood / multinomial_likelihood_ratio
// (/root/reference/ugvc/utils/stats_utils.py:31-70): add-one corrected frequencies,
// pmf(x; n, p) = exp(lgamma(n+1) + sum_i (x_i log p_i - lgamma(x_i+1))).
// (libm's lgamma inlines into hundreds of instructions per call site; four inlined copies put the kernel at the 128-register
// cap with 80 spilled registers - one out-of-line copy, same arithmetic)
__device__ __attribute__((noinline)) double lgamma_call(double v) { return lgamma(v); }

__device__ __forceinline__ double log_multinomial_pmf(const int32_t* x, const int32_t* e, int k) {
    double tot = 0.0;
    int n = 0;
    for (int i = 0; i < k; ++i) { tot += (double)e[i] + 1.0; n += x[i]; }
    double lp = lgamma_call((double)n + 1.0);
    for (int i = 0; i < k; ++i) {
        const double p = ((double)e[i] + 1.0) / tot;
        if (x[i] > 0) lp += (double)x[i] * log(p);
        lp -= lgamma_call((double)x[i] + 1.0);
    }
    return lp;
}

__global__ void sec_kernel(const int32_t* __restrict__ actual, const int32_t* __restrict__ expected,
                           int64_t n, int k, double* __restrict__ lik, double* __restrict__ ratio) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t* x = actual + i * k;
    const int32_t* e = expected + i * k;
    const double l = exp(log_multinomial_pmf(x, e, k));
    const double lmax = exp(log_multinomial_pmf(x, x, k));
    lik[i] = l;
    ratio[i] = l / lmax;
}

int launch_sec(ugvc_ctx* ctx, const int32_t* d_actual, const int32_t* d_expected, int64_t n, int k,
               double* d_lik, double* d_ratio) {
    if (n == 0) return 0;
    const unsigned grid = (unsigned)((n + 255) / 256);
    UGVC_LAUNCH(sec_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_actual, d_expected, n, k, d_lik, d_ratio);
    UGVC_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// is_homopolymer_snp + tumor/normal VAF gate
// (/root/reference/ugvc/pipelines/vcfbed/calibrate_bridging_snvs.py:9-66,110-126).
struct BridgingArgs {
    int64_t n;
    const uint16_t* contig; const int32_t* pos; const uint16_t* ref_len; const uint16_t* alt_len;
    const uint32_t* ref_off; const uint32_t* alt_off; const uint8_t* alleles;
    const float* qual; const int32_t* dp;
    const uint8_t* is_pass; const int32_t* ad_alt_sum; const int32_t* bg_ad_alt_sum; const int32_t* bg_dp;
    const uint8_t* ref; const int64_t* contig_off;
    ugvc_bridging_params p;
    uint8_t* out_hmer; uint8_t* out_pass;
};

__global__ void bridging_kernel(const BridgingArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    uint8_t is_hm = 0, pass = 0;
    // :14-20  SNP, not already PASS, qual >= min_initial_qual
    if (a.ref_len[i] == 1 && a.alt_len[i] == 1 && !a.is_pass[i] && (double)a.qual[i] >= a.p.min_initial_qual) {
        const int c = a.contig[i];
        const int64_t lo = a.contig_off[c], hi = a.contig_off[c + 1];
        const int64_t g0 = lo + a.pos[i] - 1;
        const int alt = a.alleles[a.alt_off[i]];
        const int refb = a.alleles[a.ref_off[i]];
        const int h = a.p.min_query_hmer_size;
        // :28-30 window = fetch(contig, pos-h-1, pos+h): h bases each side of the variant base
        int down = 0, up = 0, after = -1, before = -2;   // -1/-2: "" (window exhausted), never equal
        for (int k = 1; k <= h; ++k) {                   // :35-41 reference_seq[h+1:]
            const int64_t j = g0 + k;
            if (j >= hi) break;
            const int b = a.ref[j];
            if (b == alt) ++down;
            else { after = b; break; }
        }
        for (int k = 1; k <= h; ++k) {                   // :43-49 reference_seq[h-1::-1]
            const int64_t j = g0 - k;
            if (j < lo) break;
            const int b = a.ref[j];
            if (b == alt) ++up;
            else { before = b; break; }
        }
        const int hmer_size = 1 + up + down;
        // :51-55: "" == "" counts as equal in the reference, but then it must also equal record.ref
        const bool same_flank = (after == before) || (after == -1 && before == -2);
        const bool tandem = same_flank && before == refb && up == down;
        const int edge = up < down ? up : down;
        if (hmer_size >= h && !tandem && edge >= a.p.min_distance_from_edge) is_hm = 1;   // :56-60
    }
    if (is_hm) {                                        // :114-122
        const double normal_depth = (double)a.bg_dp[i];
        const int dp = a.dp[i];
        if (dp != 0) {
            const double tumor_vaf = (double)a.ad_alt_sum[i] / (double)dp;
            const double normal_vaf = (double)a.bg_ad_alt_sum[i] / fmax(0.01, normal_depth);
            if (tumor_vaf >= a.p.min_tumor_vaf && normal_vaf <= a.p.max_normal_vaf &&
                normal_depth > (double)a.p.min_normal_depth)
                pass = 1;
        }
    }
    a.out_hmer[i] = is_hm;
    a.out_pass[i] = pass;
}

}  // namespace ugvc

using namespace ugvc;

extern "C" {

int ugvc_pileup_upload(ugvc_ctx* ctx, const int64_t* offsets, const uint16_t* obs, int64_t n_loci) {
    if (!ctx || !offsets) return fail("NULL argument");
    if (n_loci < 0) return fail("negative locus count");
    UGVC_HIP(hipSetDevice(ctx->device));
    if (offsets[0] != 0) return fail("offsets[0] must be 0");
    for (int64_t i = 0; i < n_loci; ++i)
        if (offsets[i + 1] < offsets[i]) return fail("offsets must be non-decreasing");
    const int64_t m = offsets[n_loci];
    if (m > 0 && !obs) return fail("NULL observations");
    // compact device layout (4-byte offsets, u16 counts) unless a locus is deeper than 65535 or the table is huge
    int64_t deepest = 0;
    for (int64_t i = 0; i < n_loci; ++i) deepest = std::max(deepest, offsets[i + 1] - offsets[i]);
    ctx->pl_compact = (deepest <= 65535 && m < ((int64_t)1 << 32)) ? 1 : 0;
    // the longest span of observations one workgroup (256 consecutive loci) stages, from its 16-byte aligned start
    int64_t span = 0;
    for (int64_t l0 = 0; l0 < n_loci; l0 += kPlLociPerBlock) {
        const int64_t l1 = std::min(n_loci, l0 + kPlLociPerBlock);
        span = std::max(span, offsets[l1] - (offsets[l0] & ~(int64_t)7));
    }
    ctx->pl_span = span;
    if (ctx->pl_compact) {
        std::vector<uint32_t> o32((size_t)n_loci + 1);
        for (int64_t i = 0; i <= n_loci; ++i) o32[(size_t)i] = (uint32_t)offsets[i];
        if (upload(ctx, ctx->pl_off32, o32.data(), o32.size() * 4)) return -1;
        UGVC_HIP(hipStreamSynchronize(ctx->stream));          // (o32 goes out of scope)
    } else if (upload(ctx, ctx->pl_off, offsets, (size_t)(n_loci + 1) * 8)) return -1;
    if (ensure(ctx->pl_obsb, (size_t)m * 2 + 32)) return -1;            // padded: the kernel stages with 16-byte loads
    if (upload(ctx, ctx->pl_obsb, obs, (size_t)m * 2)) return -1;
    if (ensure(ctx->pl_out, (size_t)n_loci * 10 * 4)) return -1;
    UGVC_HIP(hipStreamSynchronize(ctx->stream));
    ctx->pl_n = n_loci;
    ctx->pl_obs = m;
    return 0;
}

int ugvc_pileup_tally(ugvc_ctx* ctx, const int64_t* offsets, const uint16_t* obs, int64_t n_loci,
                      const ugvc_pileup_out* out) {
    if (!out) return fail("out is NULL");
    if (ugvc_pileup_upload(ctx, offsets, obs, n_loci)) return -1;
    if (launch_pileup(ctx)) return -1;
    const size_t n = (size_t)n_loci;
    int32_t* o = ctx->pl_out.as<int32_t>();
    if (ctx->pl_compact) {
        // 20 bytes per locus come back; dp, other and vaf are derived here (the caller's columns are 4 bytes wide)
        std::vector<uint16_t> c16(5 * n);
        if (n) {
            if (out->bq_ref) UGVC_HIP(copy_out(ctx, out->bq_ref, o, n * 4));
            if (out->bq_alt) UGVC_HIP(copy_out(ctx, out->bq_alt, o + n, n * 4));
            if (out->sor) UGVC_HIP(copy_out(ctx, out->sor, o + 2 * n, n * 4));
            UGVC_HIP(copy_out(ctx, c16.data(), o + 3 * n, n * 10));
        }
        UGVC_HIP(hipStreamSynchronize(ctx->stream));
        int32_t* cols[4] = {out->ref_fwd, out->ref_rev, out->alt_fwd, out->alt_rev};
        for (size_t i = 0; i < n; ++i) {
            const int rf = c16[i], rr = c16[n + i], af = c16[2 * n + i], ar = c16[3 * n + i];
            const int dp = (int)(offsets[i + 1] - offsets[i]);
            const int v[4] = {rf, rr, af, ar};
            for (int q = 0; q < 4; ++q) if (cols[q]) cols[q][i] = v[q];
            if (out->dp) out->dp[i] = dp;
            if (out->other) out->other[i] = c16[4 * n + i];
            if (out->vaf) out->vaf[i] = dp > 0 ? (float)(af + ar) / (float)dp : 0.0f;
        }
        return 0;
    }
    void* dst[10] = {out->ref_fwd, out->ref_rev, out->alt_fwd, out->alt_rev, out->other,
                     out->dp, out->bq_ref, out->bq_alt, out->vaf, out->sor};
    for (int k = 0; k < 10; ++k)
        if (dst[k] && n) UGVC_HIP(copy_out(ctx, dst[k], o + k * n, n * 4));
    UGVC_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int ugvc_timed_pileup(ugvc_ctx* ctx, int iters, float* ms_total) {
    if (!ctx || !ms_total || iters < 1) return fail("bad arguments");
    UGVC_HIP(hipSetDevice(ctx->device));
    UGVC_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    for (int it = 0; it < iters; ++it)
        if (launch_pileup(ctx)) return -1;
    UGVC_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    UGVC_HIP(hipEventSynchronize(ctx->ev1));
    UGVC_HIP(hipEventElapsedTime(ms_total, ctx->ev0, ctx->ev1));
    return 0;
}

int ugvc_sec_likelihood_ratio(ugvc_ctx* ctx, const int32_t* actual, const int32_t* expected, int64_t n_loci,
                              int k, double* likelihood, double* ratio) {
    if (!ctx || !actual || !expected || !likelihood || !ratio) return fail("NULL argument");
    if (k < 1 || k > 64) return fail("k must be in 1..64");
    if (n_loci < 0) return fail("negative locus count");
    for (int64_t i = 0; i < n_loci * k; ++i)
        if (actual[i] < 0 || expected[i] < 0) return fail("counts must be non-negative");
    UGVC_HIP(hipSetDevice(ctx->device));
    DeviceBuf da, de, dl, dr;
    const size_t nb = (size_t)n_loci * k * 4;
    int rc = 0;
    if (upload(ctx, da, actual, nb) || upload(ctx, de, expected, nb) || ensure(dl, (size_t)n_loci * 8) ||
        ensure(dr, (size_t)n_loci * 8))
        rc = -1;
    if (!rc) rc = launch_sec(ctx, da.as<int32_t>(), de.as<int32_t>(), n_loci, k, dl.as<double>(), dr.as<double>());
    if (!rc && n_loci) {
        if (copy_out(ctx, likelihood, dl.p, (size_t)n_loci * 8) != hipSuccess ||
            copy_out(ctx, ratio, dr.p, (size_t)n_loci * 8) != hipSuccess)
            rc = fail("D2H copy failed");
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && !rc) rc = fail("stream sync failed");
    for (DeviceBuf* b : {&da, &de, &dl, &dr})
        if (b->p) dev_free(b->p);
    return rc;
}

int ugvc_bridging_snvs(ugvc_ctx* ctx, const ugvc_variants* v, const uint8_t* is_pass, const int32_t* ad_alt_sum,
                       const int32_t* bg_ad_alt_sum, const int32_t* bg_dp, const ugvc_bridging_params* p,
                       uint8_t* out_hmer_snp, uint8_t* out_pass) {
    if (!ctx || !v || !is_pass || !ad_alt_sum || !bg_ad_alt_sum || !bg_dp || !p || !out_hmer_snp || !out_pass)
        return fail("NULL argument");
    if (p->min_query_hmer_size < 1) return fail("min_query_hmer_size must be >= 1");
    if (ugvc_variants_upload(ctx, v)) return -1;
    const size_t n = (size_t)v->n;
    if (n == 0) return 0;
    DeviceBuf dp_, da, db, dd, oh, op;
    int rc = 0;
    if (upload(ctx, dp_, is_pass, n) || upload(ctx, da, ad_alt_sum, n * 4) || upload(ctx, db, bg_ad_alt_sum, n * 4) ||
        upload(ctx, dd, bg_dp, n * 4) || ensure(oh, n) || ensure(op, n))
        rc = -1;
    if (!rc) {
        BridgingArgs a;
        a.n = v->n;
        a.contig = ctx->v_contig.as<uint16_t>(); a.pos = ctx->v_pos.as<int32_t>();
        a.ref_len = ctx->v_rl.as<uint16_t>(); a.alt_len = ctx->v_al.as<uint16_t>();
        a.ref_off = ctx->v_ro.as<uint32_t>(); a.alt_off = ctx->v_ao.as<uint32_t>();
        a.alleles = ctx->v_alleles.as<uint8_t>(); a.qual = ctx->v_qual.as<float>(); a.dp = ctx->v_dp.as<int32_t>();
        a.is_pass = dp_.as<uint8_t>(); a.ad_alt_sum = da.as<int32_t>(); a.bg_ad_alt_sum = db.as<int32_t>();
        a.bg_dp = dd.as<int32_t>(); a.ref = ctx->ref.as<uint8_t>() + kRefFrontPad; a.contig_off = ctx->contig_off.as<int64_t>();
        a.p = *p; a.out_hmer = oh.as<uint8_t>(); a.out_pass = op.as<uint8_t>();
        UGVC_LAUNCH(bridging_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, a);
        if (hipGetLastError() != hipSuccess) rc = fail("bridging kernel launch failed");
    }
    if (!rc) {
        if (copy_out(ctx, out_hmer_snp, oh.p, n) != hipSuccess ||
            copy_out(ctx, out_pass, op.p, n) != hipSuccess)
            rc = fail("D2H copy failed");
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && !rc) rc = fail("stream sync failed");
    for (DeviceBuf* b : {&dp_, &da, &db, &dd, &oh, &op})
        if (b->p) dev_free(b->p);
    return rc;
}

}  // extern "C"
