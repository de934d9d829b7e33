// Auxiliary hot-path kernels: pileup tally (a11), SEC multinomial likelihood ratio (a8),
// bridging-homopolymer SNV un-filter (a12).  gfx950 only.
#include <math.h>

#include "ugvc_device.hpp"

namespace ugvc {

// ------------------------------------------------------------------------------------------
// Pileup allele/strand/base-quality tally.  BUILDER-DEFINED: the
// reference only consumes FORMAT/AD, DP, SB, VAF and INFO/SOR
// (test/resources/unit/vcfbed/test_vcftools/header.txt:3379,3391-3398; uses at
// ugvc/pipelines/vcfbed/calibrate_bridging_snvs.py:115-117).
//
// One wave walks a contiguous span of read observations with coalesced 2-byte loads; every
// lane classifies its observation (allele x strand), the wave builds one ballot mask per
// class, and the lanes that own a locus (one lane per locus of the span) popcount the class
// masks restricted to their locus' lane range.  Base-quality sums use a segmented
// shuffle-scan.  A workgroup stages its loci' CSR offsets in LDS once.
constexpr int kPlBlock = 256;
constexpr int kPlLociPerBlock = 256;

struct PileupArgs {
    int64_t n_loci;
    const int64_t* off;
    const uint16_t* obs;
    int32_t* ref_fwd; int32_t* ref_rev; int32_t* alt_fwd; int32_t* alt_rev;
    int32_t* other; int32_t* dp; int32_t* bq_ref; int32_t* bq_alt;
    float* vaf; float* sor;
};

__device__ __forceinline__ float sor_from_table(int rf, int rr, int af, int ar) {
    // GATK StrandOddsRatio on the +1 table (oracle.pileup_tally)
    const double a = rf + 1.0, b = rr + 1.0, c = af + 1.0, d = ar + 1.0;
    const double R = (a * d) / (b * c);
    const double s = log(R + 1.0 / R) + log(fmin(a, b) / fmax(a, b)) - log(fmin(c, d) / fmax(c, d));
    return (float)s;
}

__global__ __launch_bounds__(kPlBlock) void pileup_kernel(const PileupArgs a) {
    // cnt[locus][class 0..5], bq[locus][allele 0..1] accumulated in LDS with wave-level
    // pre-aggregation (ballot + popcount), then one thread per locus finalises.
    __shared__ int64_t soff[kPlLociPerBlock + 1];
    __shared__ int cnt[kPlLociPerBlock * 6];
    __shared__ int bqs[kPlLociPerBlock * 2];
    const int tid = threadIdx.x;
    const int64_t l0 = (int64_t)blockIdx.x * kPlLociPerBlock;
    const int nl = (int)((a.n_loci - l0) < kPlLociPerBlock ? (a.n_loci - l0) : kPlLociPerBlock);
    for (int j = tid; j <= nl; j += kPlBlock) soff[j] = a.off[l0 + j];
    for (int j = tid; j < kPlLociPerBlock * 6; j += kPlBlock) cnt[j] = 0;
    for (int j = tid; j < kPlLociPerBlock * 2; j += kPlBlock) bqs[j] = 0;
    __syncthreads();
    const int64_t o0 = soff[0], o1 = soff[nl];
    const int lane = tid & 63;
    // each wave takes 64-observation chunks of the block's span, round-robin
    for (int64_t base = o0 + (int64_t)(tid >> 6) * 64; base < o1; base += (kPlBlock / 64) * 64) {
        const int64_t j = base + lane;
        const bool live = j < o1;
        const unsigned o = live ? a.obs[j] : 0u;
        const int allele = o & 3, strand = (o >> 2) & 1, bq = o >> 3;
        // locus of this observation: binary search in the LDS offsets (largest l: soff[l] <= j)
        int lo = 0, len = nl;
        while (len > 1) {
            const int half = len >> 1;
            const bool ge = soff[lo + half] <= j;
            lo = ge ? lo + half : lo;
            len = ge ? len - half : half;
        }
        const int loc = live ? lo : -1;
        // segment structure inside the wave: head lane of each locus run
        const int prev = __shfl_up(loc, 1);
        const bool head = live && (lane == 0 || prev != loc);
        const unsigned long long heads = __ballot(head);
        const unsigned long long livem = __ballot(live);
        // lane range [lane, end) of my segment, valid for head lanes
        const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1)) << (lane + 1);
        const int end = above ? __ffsll((long long)above) - 1 : 64;
        const unsigned long long seg = (end == 64 ? ~0ull : ((1ull << end) - 1)) & ~((1ull << lane) - 1) & livem;
        const int cls = allele * 2 + strand;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const unsigned long long m = __ballot(live && cls == k);
            if (head) {
                const int c = __popcll(m & seg);
                if (c) atomicAdd(&cnt[loc * 6 + k], c);
            }
        }
        // base-quality sums per allele (ref, alt): segmented inclusive scan by shuffles
        int vr = (live && allele == 0) ? bq : 0;
        int va = (live && allele == 1) ? bq : 0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int ur = __shfl_down(vr, d), ua = __shfl_down(va, d);
            const int ul = __shfl_down(loc, d);
            if (lane + d < 64 && ul == loc) { vr += ur; va += ua; }
        }
        if (head) {
            if (vr) atomicAdd(&bqs[loc * 2 + 0], vr);
            if (va) atomicAdd(&bqs[loc * 2 + 1], va);
        }
    }
    __syncthreads();
    if (tid < nl) {
        const int rf = cnt[tid * 6 + 0], rr = cnt[tid * 6 + 1], af = cnt[tid * 6 + 2], ar = cnt[tid * 6 + 3];
        const int ot = cnt[tid * 6 + 4] + cnt[tid * 6 + 5];
        const int64_t l = l0 + tid;
        const int dp = (int)(soff[tid + 1] - soff[tid]);
        a.ref_fwd[l] = rf; a.ref_rev[l] = rr; a.alt_fwd[l] = af; a.alt_rev[l] = ar;
        a.other[l] = ot; a.dp[l] = dp;
        a.bq_ref[l] = bqs[tid * 2]; a.bq_alt[l] = bqs[tid * 2 + 1];
        a.vaf[l] = dp > 0 ? __fdiv_rn((float)(af + ar), (float)dp) : 0.0f;
        a.sor[l] = sor_from_table(rf, rr, af, ar);
    }
}

int launch_pileup(ugvc_ctx* ctx) {
    if (ctx->pl_n == 0) return 0;
    PileupArgs a;
    a.n_loci = ctx->pl_n;
    a.off = ctx->pl_off.as<int64_t>();
    a.obs = ctx->pl_obsb.as<uint16_t>();
    int32_t* o = ctx->pl_out.as<int32_t>();
    const int64_t n = ctx->pl_n;
    a.ref_fwd = o; a.ref_rev = o + n; a.alt_fwd = o + 2 * n; a.alt_rev = o + 3 * n;
    a.other = o + 4 * n; a.dp = o + 5 * n; a.bq_ref = o + 6 * n; a.bq_alt = o + 7 * n;
    a.vaf = reinterpret_cast<float*>(o + 8 * n);
    a.sor = reinterpret_cast<float*>(o + 9 * n);
    const unsigned grid = (unsigned)((n + kPlLociPerBlock - 1) / kPlLociPerBlock);
    hipLaunchKernelGGL(pileup_kernel, dim3(grid), dim3(kPlBlock), 0, ctx->stream, a);
    UGVC_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// SEC statistic: multinomial_likelihood / multinomial_likelihood_ratio
// (/root/reference/ugvc/utils/stats_utils.py:31-70): add-one corrected frequencies,
// pmf(x; n, p) = exp(lgamma(n+1) + sum_i (x_i log p_i - lgamma(x_i+1))).
__device__ __forceinline__ double log_multinomial_pmf(const int32_t* x, const int32_t* e, int k) {
    double tot = 0.0;
    int n = 0;
    for (int i = 0; i < k; ++i) { tot += (double)e[i] + 1.0; n += x[i]; }
    double lp = lgamma((double)n + 1.0);
    for (int i = 0; i < k; ++i) {
        const double p = ((double)e[i] + 1.0) / tot;
        if (x[i] > 0) lp += (double)x[i] * log(p);
        lp -= lgamma((double)x[i] + 1.0);
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
    hipLaunchKernelGGL(sec_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_actual, d_expected, n, k, d_lik, d_ratio);
    UGVC_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// is_homopolymer_snp + tumor/normal VAF gate
// (/root/reference/ugvc/pipelines/vcfbed/calibrate_bridging_snvs.py:9-66,110-126).
struct BridgingArgs {
    int64_t n;
    const uint8_t* contig; const int32_t* pos; const uint16_t* ref_len; const uint16_t* alt_len;
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
    if (upload(ctx, ctx->pl_off, offsets, (size_t)(n_loci + 1) * 8)) return -1;
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
    void* dst[10] = {out->ref_fwd, out->ref_rev, out->alt_fwd, out->alt_rev, out->other,
                     out->dp, out->bq_ref, out->bq_alt, out->vaf, out->sor};
    for (int k = 0; k < 10; ++k)
        if (dst[k] && n) UGVC_HIP(hipMemcpyAsync(dst[k], o + k * n, n * 4, hipMemcpyDeviceToHost, ctx->stream));
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
        if (hipMemcpyAsync(likelihood, dl.p, (size_t)n_loci * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(ratio, dr.p, (size_t)n_loci * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
            rc = fail("D2H copy failed");
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && !rc) rc = fail("stream sync failed");
    for (DeviceBuf* b : {&da, &de, &dl, &dr})
        if (b->p) (void)hipFree(b->p);
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
        a.contig = ctx->v_contig.as<uint8_t>(); a.pos = ctx->v_pos.as<int32_t>();
        a.ref_len = ctx->v_rl.as<uint16_t>(); a.alt_len = ctx->v_al.as<uint16_t>();
        a.ref_off = ctx->v_ro.as<uint32_t>(); a.alt_off = ctx->v_ao.as<uint32_t>();
        a.alleles = ctx->v_alleles.as<uint8_t>(); a.qual = ctx->v_qual.as<float>(); a.dp = ctx->v_dp.as<int32_t>();
        a.is_pass = dp_.as<uint8_t>(); a.ad_alt_sum = da.as<int32_t>(); a.bg_ad_alt_sum = db.as<int32_t>();
        a.bg_dp = dd.as<int32_t>(); a.ref = ctx->ref.as<uint8_t>(); a.contig_off = ctx->contig_off.as<int64_t>();
        a.p = *p; a.out_hmer = oh.as<uint8_t>(); a.out_pass = op.as<uint8_t>();
        hipLaunchKernelGGL(bridging_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, a);
        if (hipGetLastError() != hipSuccess) rc = fail("bridging kernel launch failed");
    }
    if (!rc) {
        if (hipMemcpyAsync(out_hmer_snp, oh.p, n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(out_pass, op.p, n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
            rc = fail("D2H copy failed");
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && !rc) rc = fail("stream sync failed");
    for (DeviceBuf* b : {&dp_, &da, &db, &dd, &oh, &op})
        if (b->p) (void)hipFree(b->p);
    return rc;
}

}  // extern "C"
