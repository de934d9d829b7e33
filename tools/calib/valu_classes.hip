// Which vector instructions of gfx950 run at the "fast" rate (~2 cycles per wave-64 instruction and SIMD) and which at the "slow" one
// (~4)?  (round 6: tools/calib/valu_issue.hip found v_add_u32 at 2.5 and shifts / compares / SDWA / v_lshl_add at 4.2-4.6 with four
// waves per SIMD.)  Same harness: 16-wave workgroups (4 per SIMD), REPS x 32 independent instructions of one kind per wave, cycles per
// instruction and SIMD from the span of a workgroup's slowest wave, median over 256 workgroups.
// hipcc -O3 --offload-arch=gfx950 valu_classes.hip -o valu_classes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

#define R4(X) X X X X
#define BODY2(OP) R4(asm volatile(OP " %0, %8, %0\n" OP " %1, %8, %1\n" OP " %2, %8, %2\n" OP " %3, %8, %3\n" OP " %4, %8, %4\n" OP " %5, %8, %5\n" OP " %6, %8, %6\n" OP " %7, %8, %7\n" \
    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));)
#define BODY2VCC(OP) R4(asm volatile(OP " %0, vcc, %8, %0\n" OP " %1, vcc, %8, %1\n" OP " %2, vcc, %8, %2\n" OP " %3, vcc, %8, %3\n" OP " %4, vcc, %8, %4\n" OP " %5, vcc, %8, %5\n" OP " %6, vcc, %8, %6\n" OP " %7, vcc, %8, %7\n" \
    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b) : "vcc");)
#define BODY3(OP) R4(asm volatile(OP " %0, %8, %0, %9\n" OP " %1, %8, %1, %9\n" OP " %2, %8, %2, %9\n" OP " %3, %8, %3, %9\n" OP " %4, %8, %4, %9\n" OP " %5, %8, %5, %9\n" OP " %6, %8, %6, %9\n" OP " %7, %8, %7, %9\n" \
    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));)
#define BODY1(OP) R4(asm volatile(OP " %0, %8\n" OP " %1, %8\n" OP " %2, %8\n" OP " %3, %8\n" OP " %4, %8\n" OP " %5, %8\n" OP " %6, %8\n" OP " %7, %8\n" \
    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));)

template <int KIND>
__global__ void k(unsigned long long* out, unsigned* sink, int reps) {
    unsigned a[8], b = threadIdx.x * 2654435761u, c = threadIdx.x + 7;
    for (int q = 0; q < 8; ++q) a[q] = threadIdx.x + q;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        if constexpr (KIND == 0) { BODY2("v_add_u32_e32") }
        else if constexpr (KIND == 1) { BODY2("v_sub_u32_e32") }
        else if constexpr (KIND == 2) { BODY2("v_and_b32_e32") }
        else if constexpr (KIND == 3) { BODY2("v_or_b32_e32") }
        else if constexpr (KIND == 4) { BODY2("v_xor_b32_e32") }
        else if constexpr (KIND == 5) { BODY2("v_min_u32_e32") }
        else if constexpr (KIND == 6) { BODY2("v_max_i32_e32") }
        else if constexpr (KIND == 7) { BODY1("v_mov_b32_e32") }
        else if constexpr (KIND == 8) { BODY2("v_lshlrev_b32_e32") }
        else if constexpr (KIND == 9) { BODY2("v_lshrrev_b32_e32") }
        else if constexpr (KIND == 10) { BODY2("v_mul_u32_u24_e32") }
        else if constexpr (KIND == 11) { BODY2VCC("v_add_co_u32_e32") }
        else if constexpr (KIND == 12) { BODY2VCC("v_sub_co_u32_e32") }
        else if constexpr (KIND == 13) { BODY3("v_add3_u32") }
        else if constexpr (KIND == 14) { BODY3("v_and_or_b32") }
        else if constexpr (KIND == 15) { BODY3("v_lshl_or_b32") }
        else if constexpr (KIND == 16) { BODY3("v_bfe_u32") }
        else if constexpr (KIND == 17) { BODY3("v_alignbit_b32") }
        else if constexpr (KIND == 18) { BODY3("v_perm_b32") }
        else if constexpr (KIND == 19) { BODY3("v_mad_u32_u24") }
        else if constexpr (KIND == 20) { BODY3("v_bfi_b32") }
        else if constexpr (KIND == 21) { BODY2("v_add_u16_e32") }
        else if constexpr (KIND == 22) { BODY2("v_sub_u16_e32") }
        else if constexpr (KIND == 23) { BODY2("v_add_f32_e32") }
        else if constexpr (KIND == 24) { BODY2("v_mul_f32_e32") }
        else if constexpr (KIND == 25) { BODY3("v_fma_f32") }
        else if constexpr (KIND == 26) { BODY3("v_min3_u32") }
        else if constexpr (KIND == 27) { BODY3("v_med3_i32") }
        else if constexpr (KIND == 28) { BODY2("v_ashrrev_i32_e32") }
        else if constexpr (KIND == 29) { BODY3("v_xad_u32") }
        else if constexpr (KIND == 30) { BODY3("v_add_lshl_u32") }
        else if constexpr (KIND == 31) { BODY3("v_sad_u32") }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    unsigned s = c;
    for (int q = 0; q < 8; ++q) s ^= a[q];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
static void run(const char* what) {
    const int n_wg = 256, reps = 1024, waves = 16;
    unsigned long long* d_out; unsigned* d_sink;
    (void)hipMalloc(&d_out, sizeof(unsigned long long) * n_wg * waves);
    (void)hipMalloc(&d_sink, 4 * n_wg * 1024);
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(k<KIND>, dim3(n_wg), dim3(waves * 64), 0, 0, d_out, d_sink, reps);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)n_wg * waves);
    (void)hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * n_wg * waves, hipMemcpyDeviceToHost);
    std::vector<double> span((size_t)n_wg);
    for (int b = 0; b < n_wg; ++b) {
        double mx = 0;
        for (int w = 0; w < waves; ++w) mx = std::max(mx, (double)h[(size_t)b * waves + w]);
        span[(size_t)b] = mx;
    }
    std::sort(span.begin(), span.end());
    printf("%-22s %5.2f cycles per instruction and SIMD\n", what, span[(size_t)n_wg / 2] / ((double)reps * 32 * (waves / 4.0)));
    (void)hipFree(d_out); (void)hipFree(d_sink);
}

int main() {
    run<0>("v_add_u32"); run<1>("v_sub_u32"); run<2>("v_and_b32"); run<3>("v_or_b32"); run<4>("v_xor_b32"); run<5>("v_min_u32"); run<6>("v_max_i32");
    run<7>("v_mov_b32"); run<8>("v_lshlrev_b32"); run<9>("v_lshrrev_b32"); run<28>("v_ashrrev_i32"); run<10>("v_mul_u32_u24"); run<11>("v_add_co_u32 (vcc out)");
    run<12>("v_sub_co_u32 (vcc out)"); run<13>("v_add3_u32"); run<14>("v_and_or_b32"); run<15>("v_lshl_or_b32"); run<16>("v_bfe_u32"); run<17>("v_alignbit_b32");
    run<18>("v_perm_b32"); run<19>("v_mad_u32_u24"); run<20>("v_bfi_b32"); run<21>("v_add_u16"); run<22>("v_sub_u16"); run<23>("v_add_f32"); run<24>("v_mul_f32");
    run<25>("v_fma_f32"); run<26>("v_min3_u32"); run<27>("v_med3_i32"); run<29>("v_xad_u32"); run<30>("v_add_lshl_u32"); run<31>("v_sad_u32");
    return 0;
}
