// Issue rate of common 32-bit VALU instructions on gfx950.  W waves on every SIMD of every CU run a long unrolled stream
// of independent instructions (inline asm: the compiler can neither fold nor reorder them); cycles per wave-instruction
// per SIMD from s_memtime, cross-checked against the kernel's wall time (HIP events).
// hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ __launch_bounds__(1024) void k(uint32_t* out, uint64_t* cyc, int iters) {
    uint32_t a[16];
    for (int q = 0; q < 16; ++q) a[q] = threadIdx.x * 17 + q;
    uint32_t s = (uint32_t)iters | 3u;
    __syncthreads();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#define ADD(q) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[q]) : "v"(s));
#define LSHLADD(q) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[q]) : "v"(s));
#define ANDOR(q) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a[q]) : "v"(s));
#define CMPSEL(q) asm volatile("v_cmp_gt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[q]) : "v"(s) : "vcc");
#define ADDC(q) asm volatile("v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(a[q]) : : "vcc");
#define FMA(q) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[q]) : "v"(s));
#define SDWA(q) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(a[q]) : "v"(s));
#define CMP16(q) asm volatile("v_cmp_lt_u16_e32 vcc, %0, %1" : : "v"(a[q]), "v"(s) : "vcc");
#define CMPSDWA(q) asm volatile("v_cmp_lt_u32_sdwa vcc, %0, %1 src0_sel:WORD_0 src1_sel:DWORD" : : "v"(a[q]), "v"(s) : "vcc");
#define CMP32(q) asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1" : : "v"(a[q]), "v"(s) : "vcc");
#define CMP16ADDC(q) asm volatile("v_cmp_lt_u16_e32 vcc, %0, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(a[q]) : "v"(s) : "vcc");
#define CMPSDWAADDC(q) asm volatile("v_cmp_lt_u32_sdwa vcc, %0, %1 src0_sel:WORD_0 src1_sel:DWORD\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(a[q]) : "v"(s) : "vcc");
#define ADDF64(q) asm volatile("v_add_f64 %0, %0, %1" : "+v"(*(double*)&a[(q) & 14]) : "v"(*(double*)&a[(q) & 14]));
            if (OP == 0) { REP16(ADD) }
            if (OP == 1) { REP16(LSHLADD) }
            if (OP == 2) { REP16(ANDOR) }
            if (OP == 3) { REP16(CMPSEL) }
            if (OP == 4) { REP16(ADDC) }
            if (OP == 5) { REP16(FMA) }
            if (OP == 6) { REP16(SDWA) }
            if (OP == 7) { REP16(CMP16) }
            if (OP == 8) { REP16(CMPSDWA) }
            if (OP == 9) { REP16(CMP32) }
            if (OP == 10) { REP16(CMP16ADDC) }
            if (OP == 11) { REP16(CMPSDWAADDC) }
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    uint32_t x = 0;
    for (int q = 0; q < 16; ++q) x ^= a[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
static void run(const char* name, int waves_per_simd, int instr_per_rep) {
    const int threads = 256 * waves_per_simd, blocks = 256, iters = 4000;
    uint32_t* out; uint64_t* cyc;
    hipMalloc(&out, (size_t)blocks * threads * 4);
    hipMalloc(&cyc, (size_t)blocks * (threads / 64) * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    uint64_t h[64];
    hipMemcpy(h, cyc, sizeof(uint64_t) * (threads / 64), hipMemcpyDeviceToHost);
    double mx = 0;
    for (int w = 0; w < threads / 64; ++w) mx = h[w] > mx ? (double)h[w] : mx;
    const double per_simd_instr = (double)iters * 64 * instr_per_rep * waves_per_simd;   // wave-instructions issued on one SIMD
    printf("%-22s %d waves/SIMD: %.2f ticks per wave-instruction per SIMD; kernel %.1f us -> %.2f ns per instruction per SIMD, %.2f GHz ticks\n",
           name, waves_per_simd, mx / per_simd_instr, ms * 1e3, ms * 1e6 / per_simd_instr, mx / (ms * 1e6));
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0>("v_add_u32", 1, 1); run<0>("v_add_u32", 4, 1);
    run<1>("v_lshl_add_u32", 1, 1); run<1>("v_lshl_add_u32", 4, 1);
    run<2>("v_and_or_b32", 4, 1);
    run<3>("v_cmp + v_cndmask", 4, 2);
    run<4>("v_addc_co_u32", 4, 1);
    run<5>("v_fma_f32", 1, 1); run<5>("v_fma_f32", 4, 1);
    run<6>("v_add_u32_sdwa", 4, 1);
    run<7>("v_cmp_lt_u16_e32", 4, 1);
    run<8>("v_cmp_lt_u32_sdwa", 4, 1);
    run<9>("v_cmp_lt_u32_e32", 4, 1);
    run<10>("v_cmp_u16 + v_addc", 4, 2);
    run<11>("v_cmp_sdwa + v_addc", 4, 2);
    return 0;
}
