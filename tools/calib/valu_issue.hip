// Issue cost of the walk's vector instructions on gfx950, per wave-64 instruction and SIMD (round 6): is an SDWA instruction, a compare
// into an SGPR pair, a v_addc with an SGPR carry-in dearer than a plain VOP2 add?  The scoring pass's walk spends 4 vector instructions per
// tree-node visit, two of them SDWA, and measured 19.7 "vector" cycles per visit in isolation where 4 x 4 = 16 would be the 16-lane
// SIMD's rate.  Every workgroup is W waves (W / 4 per SIMD); a wave runs REPS x 32 independent instructions of one kind between two
// s_memtime reads; cycles per instruction = (ticks of the slowest wave of workgroup 0) / (REPS x 32 x waves per SIMD).
// hipcc -O3 --offload-arch=gfx950 valu_issue.hip -o valu_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define R4(X) X X X X
#define R32(X) R4(R4(X)) R4(R4(X))

template <int KIND>
__global__ void k(unsigned long long* out, unsigned* sink, int reps) {
    unsigned a[8], b = threadIdx.x * 2654435761u, c = threadIdx.x + 7;
    for (int q = 0; q < 8; ++q) a[q] = threadIdx.x + q;
    unsigned long long m = 0x5555555555555555ull;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        if constexpr (KIND == 0) {          // plain VOP2 add
            R4(asm volatile("v_add_u32_e32 %0, %8, %0\n v_add_u32_e32 %1, %8, %1\n v_add_u32_e32 %2, %8, %2\n v_add_u32_e32 %3, %8, %3\n"
                            "v_add_u32_e32 %4, %8, %4\n v_add_u32_e32 %5, %8, %5\n v_add_u32_e32 %6, %8, %6\n v_add_u32_e32 %7, %8, %7\n"
                            : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));)
        } else if constexpr (KIND == 1) {   // SDWA add (the walk's code-address add)
            R4(asm volatile("v_add_u32_sdwa %0, %8, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                            "v_add_u32_sdwa %1, %8, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                            "v_add_u32_sdwa %2, %8, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                            "v_add_u32_sdwa %3, %8, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                            "v_add_u32_sdwa %4, %8, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                            "v_add_u32_sdwa %5, %8, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                            "v_add_u32_sdwa %6, %8, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                            "v_add_u32_sdwa %7, %8, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                            : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));)
        } else if constexpr (KIND == 2) {   // VOP3 shift-add (the walk's node address)
            R4(asm volatile("v_lshl_add_u32 %0, %0, 2, %8\n v_lshl_add_u32 %1, %1, 2, %8\n v_lshl_add_u32 %2, %2, 2, %8\n v_lshl_add_u32 %3, %3, 2, %8\n"
                            "v_lshl_add_u32 %4, %4, 2, %8\n v_lshl_add_u32 %5, %5, 2, %8\n v_lshl_add_u32 %6, %6, 2, %8\n v_lshl_add_u32 %7, %7, 2, %8\n"
                            : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));)
        } else if constexpr (KIND == 3) {   // the walk's pair: SDWA compare into an SGPR pair + v_addc with that carry-in (8 pairs = 16 instructions)
            R4(asm volatile("v_cmp_gt_u16_sdwa s[20:21], %0, %8 src0_sel:WORD_0 src1_sel:WORD_0\n v_addc_co_u32_e64 %0, s[22:23], %0, %0, s[20:21]\n"
                            "v_cmp_gt_u16_sdwa s[24:25], %1, %8 src0_sel:WORD_0 src1_sel:WORD_0\n v_addc_co_u32_e64 %1, s[26:27], %1, %1, s[24:25]\n"
                            "v_cmp_gt_u16_sdwa s[20:21], %2, %8 src0_sel:WORD_0 src1_sel:WORD_0\n v_addc_co_u32_e64 %2, s[22:23], %2, %2, s[20:21]\n"
                            "v_cmp_gt_u16_sdwa s[24:25], %3, %8 src0_sel:WORD_0 src1_sel:WORD_0\n v_addc_co_u32_e64 %3, s[26:27], %3, %3, s[24:25]\n"
                            : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
               asm volatile("v_cmp_gt_u16_sdwa s[20:21], %0, %8 src0_sel:WORD_0 src1_sel:WORD_0\n v_addc_co_u32_e64 %0, s[22:23], %0, %0, s[20:21]\n"
                            "v_cmp_gt_u16_sdwa s[24:25], %1, %8 src0_sel:WORD_0 src1_sel:WORD_0\n v_addc_co_u32_e64 %1, s[26:27], %1, %1, s[24:25]\n"
                            "v_cmp_gt_u16_sdwa s[20:21], %2, %8 src0_sel:WORD_0 src1_sel:WORD_0\n v_addc_co_u32_e64 %2, s[22:23], %2, %2, s[20:21]\n"
                            "v_cmp_gt_u16_sdwa s[24:25], %3, %8 src0_sel:WORD_0 src1_sel:WORD_0\n v_addc_co_u32_e64 %3, s[26:27], %3, %3, s[24:25]\n"
                            : "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
            (void)m;
        } else if constexpr (KIND == 4) {   // the same pair with a plain VOP3 compare
            R4(asm volatile("v_cmp_gt_u32_e64 s[20:21], %0, %8\n v_addc_co_u32_e64 %0, s[22:23], %0, %0, s[20:21]\n"
                            "v_cmp_gt_u32_e64 s[24:25], %1, %8\n v_addc_co_u32_e64 %1, s[26:27], %1, %1, s[24:25]\n"
                            "v_cmp_gt_u32_e64 s[20:21], %2, %8\n v_addc_co_u32_e64 %2, s[22:23], %2, %2, s[20:21]\n"
                            "v_cmp_gt_u32_e64 s[24:25], %3, %8\n v_addc_co_u32_e64 %3, s[26:27], %3, %3, s[24:25]\n"
                            : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
               asm volatile("v_cmp_gt_u32_e64 s[20:21], %0, %8\n v_addc_co_u32_e64 %0, s[22:23], %0, %0, s[20:21]\n"
                            "v_cmp_gt_u32_e64 s[24:25], %1, %8\n v_addc_co_u32_e64 %1, s[26:27], %1, %1, s[24:25]\n"
                            "v_cmp_gt_u32_e64 s[20:21], %2, %8\n v_addc_co_u32_e64 %2, s[22:23], %2, %2, s[20:21]\n"
                            "v_cmp_gt_u32_e64 s[24:25], %3, %8\n v_addc_co_u32_e64 %3, s[26:27], %3, %3, s[24:25]\n"
                            : "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
        } else if constexpr (KIND == 5) {   // compare into VCC + VOP2 addc (implicit VCC)
            R4(asm volatile("v_cmp_gt_u32_e32 vcc, %0, %8\n v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n v_cmp_gt_u32_e32 vcc, %1, %8\n v_addc_co_u32_e32 %1, vcc, %1, %1, vcc\n"
                            "v_cmp_gt_u32_e32 vcc, %2, %8\n v_addc_co_u32_e32 %2, vcc, %2, %2, vcc\n v_cmp_gt_u32_e32 vcc, %3, %8\n v_addc_co_u32_e32 %3, vcc, %3, %3, vcc\n"
                            "v_cmp_gt_u32_e32 vcc, %4, %8\n v_addc_co_u32_e32 %4, vcc, %4, %4, vcc\n v_cmp_gt_u32_e32 vcc, %5, %8\n v_addc_co_u32_e32 %5, vcc, %5, %5, vcc\n"
                            "v_cmp_gt_u32_e32 vcc, %6, %8\n v_addc_co_u32_e32 %6, vcc, %6, %6, vcc\n v_cmp_gt_u32_e32 vcc, %7, %8\n v_addc_co_u32_e32 %7, vcc, %7, %7, vcc\n"
                            : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b) : "vcc");)
        } else if constexpr (KIND == 6) {   // v_alignbit (2 I + sign bit) behind a plain subtract: the SGPR-free index update
            R4(asm volatile("v_sub_u32_e32 %8, %9, %0\n v_alignbit_b32 %0, %0, %8, 31\n v_sub_u32_e32 %8, %9, %1\n v_alignbit_b32 %1, %1, %8, 31\n"
                            "v_sub_u32_e32 %8, %9, %2\n v_alignbit_b32 %2, %2, %8, 31\n v_sub_u32_e32 %8, %9, %3\n v_alignbit_b32 %3, %3, %8, 31\n"
                            "v_sub_u32_e32 %8, %9, %4\n v_alignbit_b32 %4, %4, %8, 31\n v_sub_u32_e32 %8, %9, %5\n v_alignbit_b32 %5, %5, %8, 31\n"
                            "v_sub_u32_e32 %8, %9, %6\n v_alignbit_b32 %6, %6, %8, 31\n v_sub_u32_e32 %8, %9, %7\n v_alignbit_b32 %7, %7, %8, 31\n"
                            : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(c) : "v"(b));)
        }
        else if constexpr (KIND == 7) {   // VOP2 shift
            R4(asm volatile("v_lshlrev_b32_e32 %0, 1, %0\n v_lshlrev_b32_e32 %1, 1, %1\n v_lshlrev_b32_e32 %2, 1, %2\n v_lshlrev_b32_e32 %3, 1, %3\n"
                            "v_lshlrev_b32_e32 %4, 1, %4\n v_lshlrev_b32_e32 %5, 1, %5\n v_lshlrev_b32_e32 %6, 1, %6\n v_lshlrev_b32_e32 %7, 1, %7\n"
                            : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));)
        } else if constexpr (KIND == 8) {   // VOP2 16-bit add
            R4(asm volatile("v_add_u16_e32 %0, %8, %0\n v_add_u16_e32 %1, %8, %1\n v_add_u16_e32 %2, %8, %2\n v_add_u16_e32 %3, %8, %3\n"
                            "v_add_u16_e32 %4, %8, %4\n v_add_u16_e32 %5, %8, %5\n v_add_u16_e32 %6, %8, %6\n v_add_u16_e32 %7, %8, %7\n"
                            : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));)
        } else if constexpr (KIND == 9) {   // compares alone, into VCC (VOPC e32)
            R4(asm volatile("v_cmp_gt_u32_e32 vcc, %0, %8\n v_cmp_gt_u32_e32 vcc, %1, %8\n v_cmp_gt_u32_e32 vcc, %2, %8\n v_cmp_gt_u32_e32 vcc, %3, %8\n"
                            "v_cmp_gt_u32_e32 vcc, %4, %8\n v_cmp_gt_u32_e32 vcc, %5, %8\n v_cmp_gt_u32_e32 vcc, %6, %8\n v_cmp_gt_u32_e32 vcc, %7, %8\n"
                            : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b) : "vcc");)
        } else if constexpr (KIND == 10) {  // compares alone, into SGPR pairs (VOP3)
            R4(asm volatile("v_cmp_gt_u32_e64 s[20:21], %0, %8\n v_cmp_gt_u32_e64 s[22:23], %1, %8\n v_cmp_gt_u32_e64 s[24:25], %2, %8\n v_cmp_gt_u32_e64 s[26:27], %3, %8\n"
                            "v_cmp_gt_u32_e64 s[20:21], %4, %8\n v_cmp_gt_u32_e64 s[22:23], %5, %8\n v_cmp_gt_u32_e64 s[24:25], %6, %8\n v_cmp_gt_u32_e64 s[26:27], %7, %8\n"
                            : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
        } else if constexpr (KIND == 11) {  // v_addc alone, carry-in from a fixed SGPR pair (VOP3)
            R4(asm volatile("v_addc_co_u32_e64 %0, s[22:23], %0, %0, %9\n v_addc_co_u32_e64 %1, s[22:23], %1, %1, %9\n v_addc_co_u32_e64 %2, s[22:23], %2, %2, %9\n v_addc_co_u32_e64 %3, s[22:23], %3, %3, %9\n"
                            "v_addc_co_u32_e64 %4, s[22:23], %4, %4, %9\n v_addc_co_u32_e64 %5, s[22:23], %5, %5, %9\n v_addc_co_u32_e64 %6, s[22:23], %6, %6, %9\n v_addc_co_u32_e64 %7, s[22:23], %7, %7, %9\n"
                            : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "s"(m) : "s22", "s23");)
        } else if constexpr (KIND == 12) {  // VOP2 cndmask (VCC fixed)
            R4(asm volatile("v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cndmask_b32_e32 %3, %3, %8, vcc\n"
                            "v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cndmask_b32_e32 %7, %7, %8, vcc\n"
                            : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b) : "vcc");)
        } else if constexpr (KIND == 13) {  // VOP3-encoded plain add (the same operation as kind 0 in the 64-bit encoding)
            R4(asm volatile("v_add_u32_e64 %0, %8, %0\n v_add_u32_e64 %1, %8, %1\n v_add_u32_e64 %2, %8, %2\n v_add_u32_e64 %3, %8, %3\n"
                            "v_add_u32_e64 %4, %8, %4\n v_add_u32_e64 %5, %8, %5\n v_add_u32_e64 %6, %8, %6\n v_add_u32_e64 %7, %8, %7\n"
                            : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));)
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    unsigned s = 0;
    for (int q = 0; q < 8; ++q) s ^= a[q];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s ^ c;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
static void run(const char* what, int instr_per_rep, int waves_per_wg) {
    unsigned long long* d_out; unsigned* d_sink;
    const int n_wg = 256, reps = 2048;
    (void)hipMalloc(&d_out, sizeof(unsigned long long) * n_wg * 16);
    (void)hipMalloc(&d_sink, 4 * n_wg * 1024);
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(k<KIND>, dim3(n_wg), dim3(waves_per_wg * 64), 0, 0, d_out, d_sink, reps);
    (void)hipDeviceSynchronize();
    unsigned long long h[256 * 16];
    (void)hipMemcpy(h, d_out, sizeof(unsigned long long) * n_wg * waves_per_wg, hipMemcpyDeviceToHost);
    // per workgroup (= per CU: one workgroup each): the span of its slowest wave = when every SIMD of the CU has issued all of its
    // waves' instructions; the MEDIAN over the workgroups (a CU that was handed two workgroups would show twice the span)
    std::vector<double> span((size_t)n_wg);
    for (int b = 0; b < n_wg; ++b) {
        double mx = 0;
        for (int w = 0; w < waves_per_wg; ++w) mx = std::max(mx, (double)h[b * waves_per_wg + w]);
        span[(size_t)b] = mx;
    }
    std::sort(span.begin(), span.end());
    const double per_simd = (double)reps * instr_per_rep * (waves_per_wg / 4.0);
    printf("%-58s %2d waves/SIMD: %5.2f cycles per instruction and SIMD (median workgroup; p10 %5.2f, p90 %5.2f)\n", what, waves_per_wg / 4,
           span[(size_t)n_wg / 2] / per_simd, span[(size_t)n_wg / 10] / per_simd, span[(size_t)n_wg * 9 / 10] / per_simd);
    (void)hipFree(d_out); (void)hipFree(d_sink);
}

int main() {
    for (int w : {4, 8, 16}) {
        run<0>("v_add_u32 (VOP2)", 32, w);
        run<1>("v_add_u32_sdwa (src1 WORD_1)", 32, w);
        run<2>("v_lshl_add_u32 (VOP3)", 32, w);
        run<3>("v_cmp_gt_u16_sdwa -> SGPR pair + v_addc (SGPR carry-in)", 64, w);
        run<4>("v_cmp_gt_u32_e64 -> SGPR pair + v_addc (SGPR carry-in)", 64, w);
        run<5>("v_cmp_gt_u32 -> VCC + v_addc (VCC)", 64, w);
        run<6>("v_sub_u32 + v_alignbit_b32", 64, w);
        run<7>("v_lshlrev_b32 (VOP2)", 32, w);
        run<8>("v_add_u16 (VOP2)", 32, w);
        run<9>("v_cmp_gt_u32 -> VCC (VOPC e32), alone", 32, w);
        run<10>("v_cmp_gt_u32_e64 -> SGPR pairs (VOP3), alone", 32, w);
        run<11>("v_addc_co_u32_e64, carry-in from an SGPR pair, alone", 32, w);
        run<12>("v_cndmask_b32 (VOP2, VCC)", 32, w);
        run<13>("v_add_u32_e64 (the VOP2 add in the 64-bit encoding)", 32, w);
    }
    return 0;
}
