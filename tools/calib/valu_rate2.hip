// Issue rate of gfx950 VALU instructions, second list (round 3): which encodings / operand kinds run at the ~2.25-cycle
// rate of a plain VOP2 add and which at ~4.2 (tools/calib/valu_rate.hip found: three-operand, SDWA, carry, compares).
// 4 waves per SIMD on every CU run a long stream of independent instructions (inline asm); cycles per wave-instruction
// per SIMD from s_memtime.  Also: a tree-node visit written three ways (the scoring pass's inner loop) with its LDS reads.
// hipcc -O3 --offload-arch=gfx950 valu_rate2.hip -o valu_rate2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

#define DEFK(NAME, ASM)                                                                                        \
    __global__ __launch_bounds__(1024) void NAME(uint32_t* out, uint64_t* cyc, int iters) {                    \
        uint32_t a[16];                                                                                        \
        for (int q = 0; q < 16; ++q) a[q] = threadIdx.x * 17 + q;                                              \
        uint32_t s = (uint32_t)iters | 3u;                                                                     \
        uint32_t ss = __builtin_amdgcn_readfirstlane(s);                                                       \
        asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(a[0]), "v"(s) : "vcc");                                \
        __syncthreads();                                                                                       \
        const uint64_t t0 = __builtin_readcyclecounter();                                                      \
        for (int it = 0; it < iters; ++it) {                                                                   \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) { REP16(ASM) }                                       \
        }                                                                                                      \
        const uint64_t t1 = __builtin_readcyclecounter();                                                      \
        uint32_t x = ss;                                                                                       \
        for (int q = 0; q < 16; ++q) x ^= a[q];                                                                \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x;                                                        \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;      \
    }

#define A_AND(q) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[q]) : "v"(s));
#define A_OR(q) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[q]) : "v"(s));
#define A_XOR(q) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[q]) : "v"(s));
#define A_SHL(q) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[q]));
#define A_SHR(q) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[q]));
#define A_ASHR(q) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(a[q]));
#define A_SUB(q) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[q]) : "v"(s));
#define A_MAX(q) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[q]) : "v"(s));
#define A_MIN(q) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[q]) : "v"(s));
#define A_MOV(q) asm volatile("v_mov_b32 %0, %1" : "+v"(a[q]) : "v"(s));
#define A_CNDMASK(q) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[q]) : "v"(s) : "vcc");
#define A_CNDMASK_S(q) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[q]) : "s"(ss) : "vcc");
#define A_ADD_S(q) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[q]) : "s"(ss));
#define A_ADD_LIT(q) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(a[q]));
#define A_ADD_E64(q) asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(a[q]) : "v"(s));
#define A_ADD_U16(q) asm volatile("v_add_u16 %0, %0, %1" : "+v"(a[q]) : "v"(s));
#define A_SUB_U16(q) asm volatile("v_sub_u16 %0, %0, %1" : "+v"(a[q]) : "v"(s));
#define A_MUL24(q) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[q]) : "v"(s));
#define A_MAD24(q) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a[q]) : "v"(s));
#define A_ADD3(q) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[q]) : "v"(s));
#define A_PERM(q) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[q]) : "v"(s));
#define A_BFE(q) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(a[q]));
#define A_ALIGNBIT(q) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a[q]) : "v"(s));
#define A_LSHLOR(q) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[q]) : "v"(s));
#define A_PKADD16(q) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[q]) : "v"(s));
#define A_PKMAD16(q) asm volatile("v_pk_mad_u16 %0, %0, %1, %1" : "+v"(a[q]) : "v"(s));
#define A_PKSUBC(q) asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(a[q]) : "v"(s));
#define A_PKMIN16(q) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[q]) : "v"(s));
#define A_ADDCO(q) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[q]) : "v"(s) : "vcc");
#define A_MOV_SDWA(q) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(a[q]) : "v"(s));
#define A_CND_SDWA(q) asm volatile("v_cndmask_b32_sdwa %0, %0, %1, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "+v"(a[q]) : "v"(s) : "vcc");
#define A_ADD_DPP(q) asm volatile("v_add_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[q]) : "v"(s));
#define A_CMPS(q) asm volatile("v_cmp_lt_u32 s[20:21], %0, %1" : : "v"(a[q]), "v"(s) : "s20", "s21");
#define A_ADDF64(q) asm volatile("v_add_f64 %0, %0, %0" : "+v"(*(double*)&a[(q) & 14]));
#define A_ADDF32(q) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[q]) : "v"(s));
#define A_MULF32(q) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[q]) : "v"(s));
#define A_FMAC(q) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a[q]) : "v"(s));
#define A_CVT(q) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[q]));
#define A_READLANE(q) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(a[q]) : "s20");
#define A_DOT4(q) asm volatile("v_dot4_u32_u8 %0, %0, %1, %1" : "+v"(a[q]) : "v"(s));

DEFK(k_and, A_AND) DEFK(k_or, A_OR) DEFK(k_xor, A_XOR) DEFK(k_shl, A_SHL) DEFK(k_shr, A_SHR) DEFK(k_ashr, A_ASHR)
DEFK(k_sub, A_SUB) DEFK(k_max, A_MAX) DEFK(k_min, A_MIN) DEFK(k_mov, A_MOV) DEFK(k_cnd, A_CNDMASK)
DEFK(k_add_s, A_ADD_S) DEFK(k_add_lit, A_ADD_LIT) DEFK(k_add_e64, A_ADD_E64) DEFK(k_add_u16, A_ADD_U16) DEFK(k_sub_u16, A_SUB_U16)
DEFK(k_mul24, A_MUL24) DEFK(k_mad24, A_MAD24) DEFK(k_add3, A_ADD3) DEFK(k_perm, A_PERM) DEFK(k_bfe, A_BFE) DEFK(k_alignbit, A_ALIGNBIT)
DEFK(k_lshlor, A_LSHLOR) DEFK(k_pkadd16, A_PKADD16) DEFK(k_pkmad16, A_PKMAD16) DEFK(k_pksubc, A_PKSUBC) DEFK(k_pkmin16, A_PKMIN16)
DEFK(k_addco, A_ADDCO) DEFK(k_mov_sdwa, A_MOV_SDWA) DEFK(k_cnd_sdwa, A_CND_SDWA) DEFK(k_add_dpp, A_ADD_DPP) DEFK(k_cmps, A_CMPS)
DEFK(k_addf64, A_ADDF64) DEFK(k_addf32, A_ADDF32) DEFK(k_mulf32, A_MULF32) DEFK(k_fmac, A_FMAC) DEFK(k_cvt, A_CVT) DEFK(k_readlane, A_READLANE)
DEFK(k_dot4, A_DOT4)

// ---- a tree-node visit, three ways, with its two LDS reads (16 independent chains per lane, 4 waves per SIMD) ----------
// A: the round-2 visit: v_lshl_add (address from the heap index), ds_read_b32, v_add_sdwa, ds_read_u16, v_cmp_sdwa, v_addc
// B: address = the carried byte offset itself: ds_read_b32, v_add_sdwa, ds_read_u16, v_cmp_sdwa, v_cndmask (0 | 4), v_lshl_add
// C: as B with a 16-bit VOP2 add for the code address (planes below 64 KB)
#define LDSAS __attribute__((address_space(3)))
__device__ __forceinline__ uint32_t l32(uint32_t a) { return *(LDSAS const uint32_t*)(uintptr_t)a; }
__device__ __forceinline__ uint32_t l16(uint32_t a) { return *(LDSAS const uint16_t*)(uintptr_t)a; }

template <int MODE>
__global__ __launch_bounds__(1024) void k_visit(uint32_t* out, uint64_t* cyc, int iters) {
    extern __shared__ uint32_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    // nodes: 8192 words = rank | plane offset (0..21)*128 << 16; planes: per wave 22 x 128 bytes behind them
    for (int q = tid; q < 8192; q += blockDim.x) lds[q] = (uint32_t)(0x4000 + (q * 37 & 0x7fff)) | ((uint32_t)((q * 7) % 22) * 128u << 16);
    uint16_t* planes = reinterpret_cast<uint16_t*>(lds + 8192) + (tid >> 6) * 22 * 64;
    for (int f = 0; f < 22; ++f) planes[f * 64 + lane] = (uint16_t)(tid * 131 + f * 977);
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(LDSAS uint32_t*)lds;
    uint32_t planes_lane = base + 8192u * 4u + (uint32_t)(tid >> 6) * 22u * 128u + 2u * (uint32_t)lane;
    uint32_t four = 4u, mask = (MODE == 0 || MODE == 3) ? 2047u : 8191u;
    asm volatile("" : "+v"(four), "+v"(planes_lane), "+v"(mask));
    uint32_t x[16];
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q) x[q] = (MODE == 0 || MODE == 3) ? 32u + q + (it & 1) : 4u * (32u + q + (it & 1));
#pragma unroll
        for (int lvl = 0; lvl < 6; ++lvl) {
            uint32_t w[16], c[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (MODE == 3) {                  // VALU only: the four instructions of visit A, no LDS
                    uint32_t ad;
                    asm volatile("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(ad) : "v"(x[q]), "s"(base));
                    w[q] = ad;
                } else if (MODE == 4) {           // LDS only: the two reads of a visit, addresses from one cheap add each
                    w[q] = l32(x[q]);
                } else if (MODE == 0) {
                    uint32_t ad;
                    asm volatile("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(ad) : "v"(x[q]), "s"(base));
                    w[q] = l32(ad);
                } else {
                    w[q] = l32(x[q] + base);      // base == 0: folds away
                }
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                uint32_t p;
                if (MODE == 4) { c[q] = l16(planes_lane + (w[q] >> 20 << 7)); continue; }
                if (MODE == 2) asm volatile("v_lshrrev_b32 %0, 16, %1\n\tv_add_u16 %0, %0, %2" : "=&v"(p) : "v"(w[q]), "v"(planes_lane));
                else asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(p) : "v"(planes_lane), "v"(w[q]));
                c[q] = MODE == 3 ? p : l16(p);
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (MODE == 4) { x[q] = (x[q] + c[q]) & mask & ~3u; continue; }
                if (MODE == 0 || MODE == 3) {
                    asm volatile("v_cmp_gt_u32_sdwa vcc, %1, %2 src0_sel:DWORD src1_sel:WORD_0\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(x[q]) : "v"(c[q]), "v"(w[q]) : "vcc");
                } else {
                    uint32_t t;
                    asm volatile("v_cmp_gt_u32_sdwa vcc, %2, %3 src0_sel:DWORD src1_sel:WORD_0\n\tv_cndmask_b32 %1, 0, %4, vcc\n\tv_lshl_add_u32 %0, %0, 1, %1"
                                 : "+v"(x[q]), "=&v"(t) : "v"(c[q]), "v"(w[q]), "v"(four) : "vcc");
                }
                asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[q]) : "v"(mask));       // (keeps the walk inside the table: one cheap op in every mode)
            }
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    uint32_t r = 0;
    for (int q = 0; q < 16; ++q) r ^= x[q];
    out[blockIdx.x * blockDim.x + tid] = r;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (tid >> 6)] = t1 - t0;
}

// D (round 3, second try): no scalar register in the visit at all.  Node word = rank << 16 | plane offset; the code is read
// into the HIGH half of a register (ds_read_u16_d16_hi: the low half is written as zero on sramecc parts), so
//   d = node - code_hi   is negative  <=>  code > rank   (the plane offset in the low half cannot flip the sign),
//   x' = v_alignbit(x, d, 31) = 2 x + (d < 0):
// v_lshl_add (address), ds_read_b32, v_add_u16 (code address: planes below 64 KB), ds_read_u16_d16_hi, v_sub_u32, v_alignbit
// = 12.8 cycles by the per-instruction table against 16.9 for A.  MODE 1: the same without the LDS reads.
template <int MODE>
__global__ __launch_bounds__(1024) void k_visit_d(uint32_t* out, uint64_t* cyc, int iters) {
    extern __shared__ uint32_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    // planes FIRST (16 waves x 22 x 128 bytes = 45056 bytes: addresses below 64 KB), the 8192 node words behind them
    uint16_t* planes = reinterpret_cast<uint16_t*>(lds) + (tid >> 6) * 22 * 64;
    for (int f = 0; f < 22; ++f) planes[f * 64 + lane] = (uint16_t)((tid * 131 + f * 977) & 0x7fff);
    uint32_t* nodes = lds + 45056 / 4;
    for (int q = tid; q < 8192; q += blockDim.x) nodes[q] = ((uint32_t)(0x2000 + (q * 37 & 0x3fff)) << 16) | ((uint32_t)((q * 7) % 22) * 128u);
    __syncthreads();
    const uint32_t nodes_b = (uint32_t)(uintptr_t)(LDSAS uint32_t*)nodes;
    uint32_t planes_lane = (uint32_t)(uintptr_t)(LDSAS uint16_t*)planes + 2u * (uint32_t)lane;
    uint32_t mask = 2047u;
    asm volatile("" : "+v"(planes_lane), "+v"(mask));
    uint32_t x[16];
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q) x[q] = 32u + q + (it & 1);
        uint32_t c[16];                                     // (zeroed once per walk: the d16_hi reads only ever write the high halves)
#pragma unroll
        for (int q = 0; q < 16; ++q) c[q] = 0;
#pragma unroll
        for (int lvl = 0; lvl < 6; ++lvl) {
            uint32_t w[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                uint32_t ad;
                asm volatile("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(ad) : "v"(x[q]), "s"(nodes_b));
                w[q] = MODE == 1 ? ad : l32(ad);
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                uint32_t p;
                asm volatile("v_add_u16 %0, %1, %2" : "=v"(p) : "v"(w[q]), "v"(planes_lane));
                if (MODE == 1) c[q] = p << 16;
                else asm volatile("ds_read_u16_d16_hi %0, %1" : "+v"(c[q]) : "v"(p));
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (MODE == 0) {
                    // LDS returns in order: result q is there once at most 15 - q reads are outstanding
                    switch (q) {
                        case 0: asm volatile("s_waitcnt lgkmcnt(15)"); break; case 1: asm volatile("s_waitcnt lgkmcnt(14)"); break;
                        case 2: asm volatile("s_waitcnt lgkmcnt(13)"); break; case 3: asm volatile("s_waitcnt lgkmcnt(12)"); break;
                        case 4: asm volatile("s_waitcnt lgkmcnt(11)"); break; case 5: asm volatile("s_waitcnt lgkmcnt(10)"); break;
                        case 6: asm volatile("s_waitcnt lgkmcnt(9)"); break; case 7: asm volatile("s_waitcnt lgkmcnt(8)"); break;
                        case 8: asm volatile("s_waitcnt lgkmcnt(7)"); break; case 9: asm volatile("s_waitcnt lgkmcnt(6)"); break;
                        case 10: asm volatile("s_waitcnt lgkmcnt(5)"); break; case 11: asm volatile("s_waitcnt lgkmcnt(4)"); break;
                        case 12: asm volatile("s_waitcnt lgkmcnt(3)"); break; case 13: asm volatile("s_waitcnt lgkmcnt(2)"); break;
                        case 14: asm volatile("s_waitcnt lgkmcnt(1)"); break; default: asm volatile("s_waitcnt lgkmcnt(0)"); break;
                    }
                }
                uint32_t d;
                asm volatile("v_sub_u32 %0, %1, %2" : "=v"(d) : "v"(w[q]), "v"(c[q]));
                asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(x[q]) : "v"(d));
                asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[q]) : "v"(mask));
            }
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    uint32_t r = 0;
    for (int q = 0; q < 16; ++q) r ^= x[q];
    out[blockIdx.x * blockDim.x + tid] = r;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (tid >> 6)] = t1 - t0;
}

using K = void (*)(uint32_t*, uint64_t*, int);

static void run(K k, const char* name, int waves_per_simd, double instr_per_rep, size_t lds = 0, int iters = 4000, double reps = 64) {
    const int threads = 256 * waves_per_simd, blocks = 256;
    uint32_t* out; uint64_t* cyc;
    hipMalloc(&out, (size_t)blocks * threads * 4);
    hipMalloc(&cyc, (size_t)blocks * (threads / 64) * 8);
    if (lds) hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds, 0, out, cyc, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds, 0, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    uint64_t h[64];
    hipMemcpy(h, cyc, sizeof(uint64_t) * (threads / 64), hipMemcpyDeviceToHost);
    double mx = 0;
    for (int w = 0; w < threads / 64; ++w) mx = h[w] > mx ? (double)h[w] : mx;
    const double per_simd = (double)iters * reps * instr_per_rep * waves_per_simd;
    printf("%-34s %d waves/SIMD: %6.2f ticks per unit per SIMD   (kernel %.1f us, %s)\n", name, waves_per_simd, mx / per_simd, ms * 1e3,
           hipGetErrorString(hipGetLastError()));
    hipFree(out); hipFree(cyc);
}

int main() {
#define R(k, n) run(k, n, 4, 1)
    R(k_and, "v_and_b32"); R(k_or, "v_or_b32"); R(k_xor, "v_xor_b32"); R(k_shl, "v_lshlrev_b32"); R(k_shr, "v_lshrrev_b32"); R(k_ashr, "v_ashrrev_i32");
    R(k_sub, "v_sub_u32"); R(k_max, "v_max_u32"); R(k_min, "v_min_u32"); R(k_mov, "v_mov_b32"); R(k_cnd, "v_cndmask_b32 (vcc)");
    R(k_add_s, "v_add_u32 sgpr src0"); R(k_add_lit, "v_add_u32 literal"); R(k_add_e64, "v_add_u32_e64");
    R(k_add_u16, "v_add_u16"); R(k_sub_u16, "v_sub_u16"); R(k_mul24, "v_mul_u32_u24"); R(k_mad24, "v_mad_u32_u24"); R(k_add3, "v_add3_u32");
    R(k_perm, "v_perm_b32"); R(k_bfe, "v_bfe_u32"); R(k_alignbit, "v_alignbit_b32"); R(k_lshlor, "v_lshl_or_b32");
    R(k_pkadd16, "v_pk_add_u16"); R(k_pkmad16, "v_pk_mad_u16"); R(k_pksubc, "v_pk_sub_u16 clamp"); R(k_pkmin16, "v_pk_min_u16");
    R(k_addco, "v_add_co_u32 (carry out)"); R(k_mov_sdwa, "v_mov_b32_sdwa"); R(k_cnd_sdwa, "v_cndmask_b32_sdwa"); R(k_add_dpp, "v_add_u32_dpp");
    R(k_cmps, "v_cmp_lt_u32 -> sgpr pair"); R(k_addf64, "v_add_f64"); R(k_addf32, "v_add_f32"); R(k_mulf32, "v_mul_f32"); R(k_fmac, "v_fmac_f32");
    R(k_cvt, "v_cvt_f32_u32"); R(k_readlane, "v_readlane_b32"); R(k_dot4, "v_dot4_u32_u8");
    const size_t lds = 8192 * 4 + 16 * 22 * 128;
    // unit = one node visit (6 levels x 16 chains per iteration)
    run(k_visit<0>, "visit A (round 2: 4 VALU + 2 LDS)", 4, 1, lds, 400, 96);
    run(k_visit<1>, "visit B (offset carried: cndmask)", 4, 1, lds, 400, 96);
    run(k_visit<2>, "visit C (B + 16-bit plane add)", 4, 1, lds, 400, 96);
    run(k_visit<3>, "visit A, VALU only (no LDS reads)", 4, 1, lds, 400, 96);
    run(k_visit<4>, "visit, LDS only (2 reads + 2 cheap)", 4, 1, lds, 400, 96);
    run(k_visit_d<0>, "visit D (sub + alignbit, no SGPR)", 4, 1, lds, 400, 96);
    run(k_visit_d<1>, "visit D, VALU only", 4, 1, lds, 400, 96);
    run(k_visit<0>, "visit A (again)", 4, 1, lds, 400, 96);
    run(k_visit_d<0>, "visit D (again)", 4, 1, lds, 400, 96);
    run(k_visit<0>, "visit A", 2, 1, lds, 400, 96);
    run(k_visit<3>, "visit A, VALU only", 2, 1, lds, 400, 96);
    run(k_visit<4>, "visit, LDS only", 2, 1, lds, 400, 96);
    run(k_visit<0>, "visit A", 1, 1, lds, 400, 96);
    run(k_visit<4>, "visit, LDS only", 1, 1, lds, 400, 96);
    return 0;
}
