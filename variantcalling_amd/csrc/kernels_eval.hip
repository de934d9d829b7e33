// Score evaluation on the GPU: the direct consumers of tree_score / FILTER.
//   * ugvc_eval_counts: per variant category, labelled calls before and after filtering - the integer counts
//     under the accuracy table of `train_models_pipeline --evaluate_concordance` (evaluate.accuracy_table;
//     group names test/resources/system/test_evaluate_concordance/expected.out.stats.csv:1-10, bins
//     ugvc/reports/report_utils.py:508-538) - computed on the RESIDENT FILTER column, nothing is downloaded;
//   * ugvc_pr_curve: the cumulative curve of ReportUtils.__calc_performance
//     (ugvc/reports/report_utils.py:494-504): stable ascending sort by score (the library's own LSD radix sort,
//     kernels_prims.hip, on an order-preserving 64-bit key), running tp / fp counts (one scan of the two counters
//     packed in a 64-bit word), recall / precision / f1 per position with the formulas of
//     ugvc/utils/stats_utils.py:76-138 in f64 (bit-equal to the host numpy).
#include "ugvc_prims.hpp"

namespace ugvc {

constexpr int kEvalCats = 16;

// out[cat][0..3] = {labelled true, labelled false, labelled true & passing, labelled false & passing}
__global__ __launch_bounds__(256) void eval_counts_kernel(const uint8_t* __restrict__ filter, const int8_t* __restrict__ label,
                                                          const uint16_t* __restrict__ cats, int64_t n, unsigned long long* out) {
    __shared__ unsigned int acc[kEvalCats * 4];
    for (int k = threadIdx.x; k < kEvalCats * 4; k += blockDim.x) acc[k] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x; i0 < n; i0 += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = i0 + threadIdx.x;
        const bool live = i < n;
        const int lab = live ? label[i] : -1;
        const unsigned cm = live && lab >= 0 ? cats[i] : 0u;
        const bool pass = live && filter[i] == UGVC_FILTER_PASS;      // the score's verdict, as evaluate.accuracy_table takes it
        const bool t = lab == 1;
#pragma unroll 1
        for (int c = 0; c < kEvalCats; ++c) {
            const bool in = (cm >> c) & 1u;
            if (__builtin_amdgcn_ballot_w64(in) == 0) continue;
            const unsigned long long b0 = __builtin_amdgcn_ballot_w64(in && t), b1 = __builtin_amdgcn_ballot_w64(in && !t);
            const unsigned long long b2 = __builtin_amdgcn_ballot_w64(in && t && pass), b3 = __builtin_amdgcn_ballot_w64(in && !t && pass);
            if (lane == 0) {
                if (b0) atomicAdd(&acc[c * 4 + 0], (unsigned)__popcll(b0));
                if (b1) atomicAdd(&acc[c * 4 + 1], (unsigned)__popcll(b1));
                if (b2) atomicAdd(&acc[c * 4 + 2], (unsigned)__popcll(b2));
                if (b3) atomicAdd(&acc[c * 4 + 3], (unsigned)__popcll(b3));
            }
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < kEvalCats * 4; k += blockDim.x)
        if (acc[k]) atomicAdd(&out[k], (unsigned long long)acc[k]);
}

// order-preserving map of an f64 onto u64 (ascending): -0 joins +0, every NaN sorts last (numpy's order)
__device__ __forceinline__ uint64_t f64_key(double x) {
    if (x != x) return ~0ull;
    x += 0.0;
    const uint64_t u = (uint64_t)__double_as_longlong(x);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

__global__ void pr_keys_kernel(const double* __restrict__ s, int64_t n, uint64_t* keys, uint32_t* idx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { keys[i] = f64_key(s[i]); idx[i] = (uint32_t)i; }
}

// tp flag in the low word, fp flag in the high word: one scan carries both running counts (n < 2^31: no carry between them)
__global__ void pr_flags_kernel(const uint32_t* __restrict__ idx, const uint8_t* __restrict__ cls, int64_t n, uint64_t* __restrict__ tpfp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const int c = cls[idx[i]]; tpfp[i] = (uint64_t)(c == 1) | ((uint64_t)(c == 2) << 32); }
}

#pragma clang fp contract(off)
__device__ __forceinline__ double one_minus_ratio(double a, double b, double if_zero) {   // get_precision / get_recall
    const double den = a + b;
    return den == 0.0 ? if_zero : 1.0 - a / den;
}

__global__ void pr_finish_kernel(const uint32_t* __restrict__ idx, const double* __restrict__ s, const uint64_t* __restrict__ ctpfp,
                                 int64_t n, int64_t i_tp, int64_t i_fp, int64_t i_fn,
                                 double* s_sorted, double* recall, double* precision, double* f1, int32_t* order) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    const int64_t ctp = (int64_t)(ctpfp[i] & 0xFFFFFFFFull), cfp = (int64_t)(ctpfp[i] >> 32);
    const double c_fn = (double)(i_fn + ctp), c_tp = (double)(i_tp - ctp), c_fp = (double)(i_fp - cfp);
    const double r = one_minus_ratio(c_fn, c_tp, nan), p = one_minus_ratio(c_fp, c_tp, nan);
    double f = (p + r == 0.0) ? 0.0 : 2.0 * p * r / (p + r);
    if (p != p || r != r) f = nan;
    s_sorted[i] = s[idx[i]];
    recall[i] = r; precision[i] = p; f1[i] = f;
    if (order) order[i] = (int32_t)idx[i];
}

}  // namespace ugvc

using namespace ugvc;

extern "C" {

int ugvc_eval_counts(ugvc_ctx* ctx, const int8_t* label, const uint16_t* cat_bits, int64_t out[16][4]) {
    if (!ctx || !label || !cat_bits || !out) return fail("NULL argument");
    const int64_t n = ctx->n;
    for (int c = 0; c < kEvalCats; ++c) for (int k = 0; k < 4; ++k) out[c][k] = 0;
    if (n == 0) return 0;
    if (!ctx->r_filter.p || !ctx->scored) return fail("no scored variants resident: run ugvc_filter_resident after ugvc_variants_upload");
    UGVC_HIP(hipSetDevice(ctx->device));
    DeviceBuf d_lab, d_cat, d_out;
    int rc = 0;
    do {
        if ((rc = upload(ctx, d_lab, label, (size_t)n))) break;
        if ((rc = upload(ctx, d_cat, cat_bits, (size_t)n * 2))) break;
        if ((rc = ensure(d_out, kEvalCats * 4 * 8))) break;
        if (hipMemsetAsync(d_out.p, 0, kEvalCats * 4 * 8, ctx->stream) != hipSuccess) { rc = fail("hipMemsetAsync failed"); break; }
        const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, (int64_t)ctx->n_cus * 8);
        UGVC_LAUNCH(eval_counts_kernel, dim3(grid), dim3(256), 0, ctx->stream, ctx->r_filter.as<uint8_t>(),
                           d_lab.as<int8_t>(), d_cat.as<uint16_t>(), n, d_out.as<unsigned long long>());
        if (copy_out(ctx, out, d_out.p, kEvalCats * 4 * 8) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = fail("eval_counts: device error"); break; }
    } while (0);
    for (DeviceBuf* b : {&d_lab, &d_cat, &d_out}) if (b->p) dev_free(b->p);
    return rc;
}

int ugvc_pr_curve(ugvc_ctx* ctx, const double* score, const uint8_t* cls, int64_t n, int64_t initial_tp, int64_t initial_fp,
                  int64_t initial_fn, double* sorted_score, double* recall, double* precision, double* f1, int32_t* order,
                  float* ms_device) {
    if (!ctx || !score || !cls || !sorted_score || !recall || !precision || !f1) return fail("NULL argument");
    if (n < 0 || n >= ((int64_t)1 << 31)) return fail("n out of range");
    if (ms_device) *ms_device = 0.f;
    if (n == 0) return 0;
    UGVC_HIP(hipSetDevice(ctx->device));
    DeviceBuf d_s, d_cls, d_k0, d_k1, d_i0, d_i1, d_tf, d_o, d_ord, d_tmp;
    DeviceBuf* all[] = {&d_s, &d_cls, &d_k0, &d_k1, &d_i0, &d_i1, &d_tf, &d_o, &d_ord, &d_tmp};
    int rc = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    do {
        const size_t N = (size_t)n;
        if ((rc = upload(ctx, d_s, score, N * 8)) || (rc = upload(ctx, d_cls, cls, N))) break;
        if ((rc = ensure(d_k0, N * 8)) || (rc = ensure(d_k1, N * 8)) || (rc = ensure(d_i0, N * 4)) || (rc = ensure(d_i1, N * 4)) ||
            (rc = ensure(d_tf, N * 8)) || (rc = ensure(d_o, N * 8 * 4)) || (rc = ensure(d_ord, N * 4))) break;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { rc = fail("hipEventCreate failed"); break; }
        const unsigned grid = (unsigned)((n + 255) / 256);
        (void)hipEventRecord(e0, ctx->stream);
        UGVC_LAUNCH(pr_keys_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_s.as<double>(), n, d_k0.as<uint64_t>(), d_i0.as<uint32_t>());
        uint64_t* ks = nullptr;
        uint32_t* is = nullptr;
        if ((rc = radix_sort_pairs_u64(ctx, d_tmp, d_k0.as<uint64_t>(), d_k1.as<uint64_t>(), d_i0.as<uint32_t>(), d_i1.as<uint32_t>(), n, &ks, &is))) break;
        UGVC_LAUNCH(pr_flags_kernel, dim3(grid), dim3(256), 0, ctx->stream, is, d_cls.as<uint8_t>(), n, d_tf.as<uint64_t>());
        if ((rc = scan_u64(ctx, d_tmp, d_tf.as<uint64_t>(), n, true))) break;
        double* o = d_o.as<double>();
        UGVC_LAUNCH(pr_finish_kernel, dim3(grid), dim3(256), 0, ctx->stream, is, d_s.as<double>(),
                           d_tf.as<uint64_t>(), n, initial_tp, initial_fp, initial_fn, o, o + N, o + 2 * N, o + 3 * N,
                           order ? d_ord.as<int32_t>() : nullptr);
        (void)hipEventRecord(e1, ctx->stream);
        bool ok = copy_out(ctx, sorted_score, o, N * 8) == hipSuccess &&
                  copy_out(ctx, recall, o + N, N * 8) == hipSuccess &&
                  copy_out(ctx, precision, o + 2 * N, N * 8) == hipSuccess &&
                  copy_out(ctx, f1, o + 3 * N, N * 8) == hipSuccess;
        if (ok && order) ok = copy_out(ctx, order, d_ord.p, N * 4) == hipSuccess;
        if (!ok || hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = fail("pr_curve: device error"); break; }
        if (ms_device) (void)hipEventElapsedTime(ms_device, e0, e1);
    } while (0);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    for (DeviceBuf* b : all) if (b->p) dev_free(b->p);
    return rc;
}

}  // extern "C"
