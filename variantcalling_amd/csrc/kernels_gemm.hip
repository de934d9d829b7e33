// Config C5: tree-ensemble inference on the resident N x F feature matrix, cast as a leaf-matrix GEMM
// on the matrix cores (BASELINE.json configs[4]; SURVEY.md 8(d) "Algorithmic FLOPs for C5"), next to a
// plain traversal kernel over the same matrix.  gfx950 only.
//
// Path-matrix formulation for complete binary trees of depth 6 (I = 63 internal nodes, L = 64 leaves;
// shallower trees are padded, the padding replicates leaf payloads so padded decisions are immaterial):
//   t[n][i] = 1 if variant n goes RIGHT at node i (exact f32 compare on the VALU), else 0;
//   C[i][l] = +1 / -1 if leaf l lies in the right / left subtree of node i, 0 if i is not an ancestor;
//   S = t . C  (N x 64 by 64 x 64, int8 x int8 -> int32 on v_mfma_i32_16x16x64_i8: K = 64 nodes in ONE
//   instruction, 4 instructions for the 64 leaves); leaf l is the exit leaf iff S[n][l] == popcount(l)
//   (every one of its 6 ancestors agrees).  Sums are at most 6 in magnitude: exact.
// The exit leaf's f32 margin is added per tree IN TREE ORDER, so the result is bit-identical to the
// traversal kernel and to the oracle.  C is the same for every tree (heap-ordered complete trees), so
// the B operand lives in 16 VGPRs for the whole kernel.  The GEMM does 2*64*64 = 8192 int-ops per tree and
// variant against 6 node visits for the traversal: it is reported beside it, not instead of it.
#include <atomic>
#include <string.h>
#include "ugvc_v2.hpp"

namespace ugvc {

typedef int i32x4_t __attribute__((ext_vector_type(4)));

constexpr int kGemmThreads = 512;       // 8 waves, one 16-row tile per wave and step
constexpr int kGemmXStride = 25;        // dwords per staged feature row (F <= 22; odd: conflict-free)

struct GemmArgs {
    const float* X;            // N x F row-major (resident feature matrix)
    int F;
    const int32_t* rows;       // optional row index list
    int64_t n;                 // rows to evaluate
    const float2* nodes;       // [T][64]: {threshold, feature-as-int bits}; slot 63 is padding
    const float* leaves;       // [T][64]
    int T, kind;
    float base;
    float* out;                // margin per evaluated row
    int out_by_row = 0;        // forest_rows_kernel: the margin of evaluated row i goes to out[rows[i]] (what forest_gemm3_kernel does) instead of out[i]
};

__device__ __forceinline__ int path_entry(int i, int l) {          // C[i][l], node i = heap index - 1
    if (i >= 63) return 0;
    const int d = 31 - __builtin_clz(i + 1);                        // level of node i
    const int p = i + 1 - (1 << d);                                 // position inside the level
    if ((l >> (6 - d)) != p) return 0;
    return ((l >> (5 - d)) & 1) ? 1 : -1;
}

__global__ __launch_bounds__(kGemmThreads) void forest_gemm_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2* nodes = reinterpret_cast<float2*>(smem);
    float* leaves = reinterpret_cast<float*>(smem + (size_t)g.T * 64 * 8);
    float* xs_all = leaves + (size_t)g.T * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int k = tid; k < g.T * 64; k += kGemmThreads) { nodes[k] = g.nodes[k]; leaves[k] = g.leaves[k]; }
    __syncthreads();
    float* xs = xs_all + wave * 16 * kGemmXStride;

    // B operand: lane holds C[k = (lane>>4)*16 .. +16][leaf = 16*j + (lane&15)] for the four leaf blocks j
    i32x4_t B[4];
    const int col = lane & 15, kb = (lane >> 4) * 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 16; ++q) w[q >> 2] |= (path_entry(kb + q, 16 * j + col) & 0xff) << (8 * (q & 3));
        B[j] = i32x4_t{w[0], w[1], w[2], w[3]};
    }
    int want[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) want[j] = __popc(16 * j + col);

    const int64_t n_tiles = (g.n + 15) / 16;
    const int row = lane & 15;
    for (int64_t tile = (int64_t)blockIdx.x * (kGemmThreads / 64) + wave; tile < n_tiles;
         tile += (int64_t)gridDim.x * (kGemmThreads / 64)) {
        // stage the tile's 16 feature rows (lane -> (row, 4 consecutive features) pieces)
        for (int e = lane; e < 16 * g.F; e += 64) {
            const int r = e / g.F, f = e - r * g.F;
            int64_t gr = tile * 16 + r;
            if (gr >= g.n) gr = g.n - 1;
            const int64_t src = g.rows ? (int64_t)g.rows[gr] : gr;
            xs[r * kGemmXStride + f] = g.X[src * g.F + f];
        }
        const float* xrow = xs + row * kGemmXStride;
        float margin[4] = {g.base, g.base, g.base, g.base};        // rows (lane>>4)*4 + r of the tile
        for (int t = 0; t < g.T; ++t) {
            const float2* tn = nodes + t * 64 + kb;
            int a[4] = {0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float2 nd = tn[q];
                const float x = xrow[__float_as_int(nd.y)];
                const bool left = g.kind == UGVC_MODEL_RF ? x <= nd.x : x < nd.x;
                a[q >> 2] |= (left ? 0 : 1) << (8 * (q & 3));
            }
            const i32x4_t A = i32x4_t{a[0], a[1], a[2], a[3]};
            const i32x4_t zero = i32x4_t{0, 0, 0, 0};
            float hit[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const i32x4_t S = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B[j], zero, 0, 0, 0);
                const float lv = leaves[t * 64 + 16 * j + col];
#pragma unroll
                for (int r = 0; r < 4; ++r) hit[r] += S[r] == want[j] ? lv : 0.0f;    // at most one non-zero term per row
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = hit[r];                                   // exactly one lane of the 16 holds the leaf value
                v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
                margin[r] += v;                                     // tree order, f32: as the traversal
            }
        }
        if (col < 4) {
            const int64_t gr = tile * 16 + (lane >> 4) * 4 + col;
            const float m = col == 0 ? margin[0] : (col == 1 ? margin[1] : (col == 2 ? margin[2] : margin[3]));
            if (gr < g.n) g.out[gr] = m;
        }
    }
}

// ---- round 4: the same GEMM, predicates without LDS gathers ---------------------------------------------------------
// forest_gemm_kernel above builds a row's 64 predicates in four lanes, sixteen each, every one with an LDS read of the node
// and an LDS GATHER of the feature value (the node's feature differs between the four lane groups): ~34 LDS + ~90 vector
// instructions per tree and 16-row tile feed four MFMAs - 203 int8-TOP/s, 0.04 of the matrix cores' rate.
// Here a lane owns ONE ROW of a 64-row tile for the predicates: its F feature values sit in a register vector, a tree's 63
// (threshold, feature) pairs are wave-uniform - scalar loads - so a predicate is: feature value by REGISTER INDEX (the index is
// in a scalar register: s_set_gpr_idx / v_movrels, no LDS), one compare against the scalar threshold, one byte-select write
// into the lane's 64-byte predicate vector.  The vector goes through the wave's LDS scratch once per tree (row-major, 80-byte
// rows) and comes back in the MFMA operand layout for the four 16-row tiles.  The product is formed TRANSPOSED - leaves x rows,
// D = C^T t^T + (-popcount(leaf)) - so that a lane holds sixteen leaves of ONE row per leaf block, the exit leaf (D == 0) is
// picked by a compare + select per element and reduced over the four lanes of its row; the margin is added by the row's own
// lane, in tree order: bit-identical to the traversal and the oracle (same compares, same f32 adds).
typedef float f32x32_t __attribute__((ext_vector_type(32)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
constexpr int kGemm2Threads = 512;
constexpr int kGemm2RowB = 80;          // bytes per staged predicate row (64 + 16: conflict-free 16-byte accesses)

// byte K of `a` := (the lane goes RIGHT at the node) ? 1 : 0 - right = !(x <= thr) for scikit-learn forests, !(x < thr) for
// XGBoost-style ensembles (a NaN feature goes right in both, as in the traversal kernel's compares).  x = xv[fidx]: the compare
// reads the feature vector by REGISTER INDEX (VGPR index mode on its first source: xv's first register + fidx) - no move, no LDS.
// `xv` is passed as an operand of its own so that the 32 registers stay live and contiguous; `x0` is its element 0 - every
// instance leaves a comment `; ugvc_pred x0=vN xv=v[N:N+31]` in the listing, and the BUILD checks it (csrc/Makefile: check-gemm,
// tools/isa/check_pred_asm.py: x0 is the tuple's first register in every instance) before the library is linked; at run time the
// first use per kernel is compared with the scalar traversal, which serves the calls if they differ.
// (s_set_gpr_idx_on writes M0, which clang reserves and warns about when it is clobbered; nothing else in this kernel uses M0)
#pragma clang diagnostic ignored "-Winline-asm"
#define UGVC_PRED_ASM(CMP, BYTE)                                                                                          \
    asm("; ugvc_pred x0=%4 xv=%6\n\ts_set_gpr_idx_on %5, gpr_idx(SRC0)\n\t" CMP " vcc, %4, %3\n\ts_set_gpr_idx_off\n\t"     \
        "v_cndmask_b32_sdwa %0, %1, %2, vcc dst_sel:" BYTE " dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"   \
        : "+v"(a) : "v"(zero), "v"(one), "s"(thr), "v"(x0), "s"(fidx), "v"(xv) : "vcc", "m0")
template <int K, bool RF>
__device__ __forceinline__ void pred_byte(uint32_t& a, const f32x32_t& xv, int fidx, float thr, uint32_t zero, uint32_t one) {
    const float x0 = xv[0];
    if constexpr (RF) {
        if constexpr (K == 0) UGVC_PRED_ASM("v_cmp_nle_f32_e64", "BYTE_0");
        if constexpr (K == 1) UGVC_PRED_ASM("v_cmp_nle_f32_e64", "BYTE_1");
        if constexpr (K == 2) UGVC_PRED_ASM("v_cmp_nle_f32_e64", "BYTE_2");
        if constexpr (K == 3) UGVC_PRED_ASM("v_cmp_nle_f32_e64", "BYTE_3");
    } else {
        if constexpr (K == 0) UGVC_PRED_ASM("v_cmp_nlt_f32_e64", "BYTE_0");
        if constexpr (K == 1) UGVC_PRED_ASM("v_cmp_nlt_f32_e64", "BYTE_1");
        if constexpr (K == 2) UGVC_PRED_ASM("v_cmp_nlt_f32_e64", "BYTE_2");
        if constexpr (K == 3) UGVC_PRED_ASM("v_cmp_nlt_f32_e64", "BYTE_3");
    }
}

#define UGVC_GEMM_CONST __attribute__((address_space(4)))

template <bool RF>
__global__ __launch_bounds__(kGemm2Threads) void forest_gemm2_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* leaves = reinterpret_cast<float*>(smem);                                        // [T][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int k = tid; k < g.T * 16; k += kGemm2Threads) reinterpret_cast<float4*>(leaves)[k] = reinterpret_cast<const float4*>(g.leaves)[k];
    __syncthreads();
    unsigned char* stage = smem + (size_t)g.T * 64 * 4 + (size_t)wave * 64 * kGemm2RowB;  // the wave's 64 x 80-byte predicate rows
    const uint32_t stage_b = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)stage;
    const uint32_t leaves_b = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)leaves;

    // A operand of the transposed product: lane holds C^T[leaf = 16 j + (lane & 15)][k = (lane >> 4) * 16 .. + 16]
    i32x4_t CT[4], bias[4];
    const int col = lane & 15, kb = (lane >> 4) * 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 16; ++q) w[q >> 2] |= (path_entry(kb + q, 16 * j + col) & 0xff) << (8 * (q & 3));
        CT[j] = i32x4_t{w[0], w[1], w[2], w[3]};
        // D[leaf = 16 j + 4 (lane >> 4) + q][row = lane & 15]: zero for the exit leaf, negative elsewhere
        const int l0 = 16 * j + 4 * (lane >> 4);
        bias[j] = i32x4_t{-(int)__popc(l0), -(int)__popc(l0 + 1), -(int)__popc(l0 + 2), -(int)__popc(l0 + 3)};
    }
    const uint32_t zero = 0u, one = 1u;
    const f32x2_t UGVC_GEMM_CONST* const nodes = (const f32x2_t UGVC_GEMM_CONST*)(uintptr_t)g.nodes;   // wave-uniform: scalar loads
    const int64_t n_tiles = (g.n + 63) / 64;
    const int F = g.F;
    for (int64_t tile = (int64_t)blockIdx.x * (kGemm2Threads / 64) + wave; tile < n_tiles; tile += (int64_t)gridDim.x * (kGemm2Threads / 64)) {
        int64_t gr = tile * 64 + lane;
        const bool live = gr < g.n;
        if (!live) gr = g.n - 1;
        const int64_t src = g.rows ? (int64_t)g.rows[gr] : gr;
        const float* xr = g.X + src * F;
        f32x32_t xv;
#pragma unroll
        for (int f = 0; f < 32; ++f) xv[f] = 0.f;
        if ((F & 3) == 0) {
#pragma unroll
            for (int q = 0; q < kMaxFeatures / 4 + 1; ++q)
                if (4 * q < F) {
                    const float4 x4 = reinterpret_cast<const float4*>(xr)[q];
                    xv[4 * q] = x4.x; xv[4 * q + 1] = x4.y; xv[4 * q + 2] = x4.z; xv[4 * q + 3] = x4.w;
                }
        } else {
#pragma unroll
            for (int f = 0; f < kMaxFeatures; ++f)
                if (f < F) xv[f] = xr[f];
        }
        float margin = g.base;
        for (int t = 0; t < g.T; ++t) {
            uint32_t a[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) a[q] = 0u;
            const f32x2_t UGVC_GEMM_CONST* tn = nodes + (size_t)t * 64;
#pragma unroll
            for (int i = 0; i < 63; ++i) {                           // (slot 63 is padding: its path-matrix row is zero)
                const f32x2_t nd = tn[i];
                const float thr = nd[0];
                const int fidx = __float_as_int(nd[1]);              // (< kMaxFeatures: gemm_model; the padding slots name feature 0)
                if ((i & 3) == 0) pred_byte<0, RF>(a[i >> 2], xv, fidx, thr, zero, one);
                else if ((i & 3) == 1) pred_byte<1, RF>(a[i >> 2], xv, fidx, thr, zero, one);
                else if ((i & 3) == 2) pred_byte<2, RF>(a[i >> 2], xv, fidx, thr, zero, one);
                else pred_byte<3, RF>(a[i >> 2], xv, fidx, thr, zero, one);
            }
            // the lane's 64 predicate bytes -> its row of the staging area; back in the operand layout of the four row tiles
            __builtin_amdgcn_wave_barrier();
            const uint32_t wr = stage_b + (uint32_t)lane * kGemm2RowB;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                *(__attribute__((address_space(3))) i32x4_t*)(uintptr_t)(wr + 16u * c) = i32x4_t{(int)a[4 * c], (int)a[4 * c + 1], (int)a[4 * c + 2], (int)a[4 * c + 3]};
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            f32x4_t lv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                lv[j] = *(const __attribute__((address_space(3))) f32x4_t*)(uintptr_t)(leaves_b + (uint32_t)(t * 256 + 64 * j + 16 * (lane >> 4)));
            float mine = 0.f;                                        // the exit leaf's value of THIS lane's row (tile lane >> 4)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const i32x4_t P = *(const __attribute__((address_space(3))) i32x4_t*)(uintptr_t)(stage_b + (uint32_t)((16 * r + col) * kGemm2RowB + kb));
                uint32_t hit = 0u;                                   // bits of the exit leaf's f32 value if one of this lane's 16 leaves is it
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const i32x4_t D = __builtin_amdgcn_mfma_i32_16x16x64_i8(CT[j], P, bias[j], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) hit = D[q] == 0 ? __float_as_uint(lv[j][q]) : hit;
                }
                // the four lanes of a row (same lane & 15) hold three zeros and the value
                hit |= (uint32_t)__shfl_xor((int)hit, 16);
                hit |= (uint32_t)__shfl_xor((int)hit, 32);
                mine = (lane >> 4) == r ? __uint_as_float(hit) : mine;
            }
            margin += mine;                                          // tree order, f32: as the traversal
            __builtin_amdgcn_wave_barrier();
        }
        if (live) g.out[tile * 64 + lane] = margin;
    }
}

// ---- round 5: ONE launch for the three variant-type groups, the exit leaf by a maximum instead of 64 compares -------------
// forest_gemm2_kernel spends, per tree and 64-row tile, ~126 vector instructions on the predicates and ~128 on FINDING the exit
// leaf (a compare + a select for each of a lane's sixteen D elements per row tile) around sixteen MFMAs - and runs once per
// variant-type group, three launches with three tails.  Here (i) the path matrix is scaled by 64 and the bias of leaf l is
// -64 popcount(l) + l, so that D = 64 (agreements - ancestors) + l is the leaf's own INDEX (>= 0) for the exit leaf and negative
// for every other leaf: the exit leaf of a row is the MAXIMUM of its 64 D values - eight v_max3 per row tile and lane, two
// cross-lane maxima - and its margin ONE LDS read by the row's own lane (|64 C| = 64 fits int8; |D| <= 6 x 64 + 63); (ii) a
// workgroup serves ONE group (its leaves in LDS, its nodes by scalar loads), the workgroups are split over the groups by rows x
// trees, margins are written by ROW of the resident matrix.  Same compares, same f32 additions in tree order: bit-identical.
struct Gemm3Args {
    GemmArgs g[UGVC_N_GROUPS];
    int wg_end[UGVC_N_GROUPS];       // workgroups [wg_end[k-1], wg_end[k]) serve group k
};

template <bool RF>
__global__ __launch_bounds__(kGemm2Threads) void forest_gemm3_kernel(const Gemm3Args a3) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int gi = 0;
#pragma unroll
    for (int k = 0; k + 1 < UGVC_N_GROUPS; ++k) gi += (int)blockIdx.x >= a3.wg_end[k] ? 1 : 0;
    gi = __builtin_amdgcn_readfirstlane(gi);
    const GemmArgs& g = a3.g[gi];
    const int wg0 = gi == 0 ? 0 : a3.wg_end[gi - 1];
    const int lb = (int)blockIdx.x - wg0, nb = a3.wg_end[gi] - wg0;
    float* leaves = reinterpret_cast<float*>(smem);                                        // [T][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int k = tid; k < g.T * 16; k += kGemm2Threads) reinterpret_cast<float4*>(leaves)[k] = reinterpret_cast<const float4*>(g.leaves)[k];
    __syncthreads();
    unsigned char* stage = smem + (size_t)g.T * 64 * 4 + (size_t)wave * 64 * kGemm2RowB;
    const uint32_t stage_b = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)stage;
    const uint32_t leaves_b = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)leaves;
    i32x4_t CT[4], bias[4];
    const int col = lane & 15, kb = (lane >> 4) * 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 16; ++q) w[q >> 2] |= ((64 * path_entry(kb + q, 16 * j + col)) & 0xff) << (8 * (q & 3));
        CT[j] = i32x4_t{w[0], w[1], w[2], w[3]};
        const int l0 = 16 * j + 4 * (lane >> 4);                  // D[leaf = l0 + q][row = lane & 15]
        bias[j] = i32x4_t{l0 - 64 * (int)__popc(l0), l0 + 1 - 64 * (int)__popc(l0 + 1), l0 + 2 - 64 * (int)__popc(l0 + 2), l0 + 3 - 64 * (int)__popc(l0 + 3)};
    }
    const uint32_t zero = 0u, one = 1u;
    const f32x2_t UGVC_GEMM_CONST* const nodes = (const f32x2_t UGVC_GEMM_CONST*)(uintptr_t)g.nodes;
    const int64_t n_tiles = (g.n + 63) / 64;
    const int F = g.F;
    for (int64_t tile = (int64_t)lb * (kGemm2Threads / 64) + wave; tile < n_tiles; tile += (int64_t)nb * (kGemm2Threads / 64)) {
        int64_t gr = tile * 64 + lane;
        const bool live = gr < g.n;
        if (!live) gr = g.n - 1;
        const int64_t src = g.rows ? (int64_t)g.rows[gr] : gr;
        const float* xr = g.X + src * F;
        f32x32_t xv;
#pragma unroll
        for (int f = 0; f < 32; ++f) xv[f] = 0.f;
        if ((F & 3) == 0) {
#pragma unroll
            for (int q = 0; q < kMaxFeatures / 4 + 1; ++q)
                if (4 * q < F) {
                    const float4 x4 = reinterpret_cast<const float4*>(xr)[q];
                    xv[4 * q] = x4.x; xv[4 * q + 1] = x4.y; xv[4 * q + 2] = x4.z; xv[4 * q + 3] = x4.w;
                }
        } else {
#pragma unroll
            for (int f = 0; f < kMaxFeatures; ++f)
                if (f < F) xv[f] = xr[f];
        }
        float margin = g.base;
        for (int t = 0; t < g.T; ++t) {
            uint32_t a[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) a[q] = 0u;
            const f32x2_t UGVC_GEMM_CONST* tn = nodes + (size_t)t * 64;
#pragma unroll
            for (int i = 0; i < 63; ++i) {
                const f32x2_t nd = tn[i];
                const float thr = nd[0];
                const int fidx = __float_as_int(nd[1]);
                if ((i & 3) == 0) pred_byte<0, RF>(a[i >> 2], xv, fidx, thr, zero, one);
                else if ((i & 3) == 1) pred_byte<1, RF>(a[i >> 2], xv, fidx, thr, zero, one);
                else if ((i & 3) == 2) pred_byte<2, RF>(a[i >> 2], xv, fidx, thr, zero, one);
                else pred_byte<3, RF>(a[i >> 2], xv, fidx, thr, zero, one);
            }
            __builtin_amdgcn_wave_barrier();
            const uint32_t wr = stage_b + (uint32_t)lane * kGemm2RowB;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                *(__attribute__((address_space(3))) i32x4_t*)(uintptr_t)(wr + 16u * c) = i32x4_t{(int)a[4 * c], (int)a[4 * c + 1], (int)a[4 * c + 2], (int)a[4 * c + 3]};
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            int lx = 0;                                              // the exit leaf of THIS lane's row (row tile lane >> 4)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const i32x4_t P = *(const __attribute__((address_space(3))) i32x4_t*)(uintptr_t)(stage_b + (uint32_t)((16 * r + col) * kGemm2RowB + kb));
                int m = -1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const i32x4_t D = __builtin_amdgcn_mfma_i32_16x16x64_i8(CT[j], P, bias[j], 0, 0, 0);
                    m = max(m, max(max(D[0], D[1]), max(D[2], D[3])));
                }
                // the four lanes of a row (same lane & 15) hold its 64 leaves: the one non-negative value is the exit leaf's index
                m = max(m, __shfl_xor(m, 16));
                m = max(m, __shfl_xor(m, 32));
                lx = (lane >> 4) == r ? m : lx;
            }
            margin += *(const __attribute__((address_space(3))) float*)(uintptr_t)(leaves_b + 4u * (uint32_t)(t * 64 + lx));   // tree order, f32
            __builtin_amdgcn_wave_barrier();
        }
        if (live) g.out[src] = margin;
    }
}

// Traversal over the same matrix and the same dense node table: one lane per row, fixed 6-level walk.
__global__ __launch_bounds__(256) void forest_rows_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2* nodes = reinterpret_cast<float2*>(smem);
    float* leaves = reinterpret_cast<float*>(smem + (size_t)g.T * 64 * 8);
    for (int k = threadIdx.x; k < g.T * 64; k += 256) { nodes[k] = g.nodes[k]; leaves[k] = g.leaves[k]; }
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < g.n; i += (int64_t)gridDim.x * 256) {
        const int64_t src = g.rows ? (int64_t)g.rows[i] : i;
        const float* x = g.X + src * g.F;
        float xv[kMaxFeatures];
#pragma unroll
        for (int f = 0; f < kMaxFeatures; ++f) xv[f] = f < g.F ? x[f] : 0.f;
        float margin = g.base;
        for (int t = 0; t < g.T; ++t) {
            int idx = 1;
#pragma unroll
            for (int d = 0; d < 6; ++d) {
                const float2 nd = nodes[t * 64 + idx - 1];
                const int f = __float_as_int(nd.y);
                float xf = xv[0];
#pragma unroll
                for (int q = 1; q < kMaxFeatures; ++q) xf = f == q ? xv[q] : xf;
                const bool left = g.kind == UGVC_MODEL_RF ? xf <= nd.x : xf < nd.x;
                idx = 2 * idx + (left ? 0 : 1);
            }
            margin += leaves[t * 64 + idx - 64];
        }
        g.out[g.out_by_row ? src : i] = margin;
    }
}

}  // namespace ugvc

using namespace ugvc;

namespace ugvc {
int gemm_model(ugvc_ctx* ctx, int group, std::vector<float2>& nodes, std::vector<float>& leaves, int& T, int& kind, float& base);
}

extern "C" int ugvc_forest_gemm(ugvc_ctx* ctx, int group, const int32_t* rows, int64_t n_rows, int use_mfma, int iters,
                                float* margin_out, float* ms_per_launch) {
    if (!ctx || !margin_out) return fail("NULL argument");
    if (group < 0 || group >= UGVC_N_GROUPS) return fail("group out of range");
    if (iters < 1) iters = 1;
    UGVC_HIP(hipSetDevice(ctx->device));
    const int F = UGVC_N_BASE_FEATURES + ctx->n_tracks;
    if (!ctx->x_mat.p || ctx->x_mat.cap < (size_t)ctx->n * F * 4) return fail("no resident feature matrix (ugvc_feature_matrix first)");
    const int64_t n = rows ? n_rows : ctx->n;
    if (n <= 0) return 0;
    std::vector<float2> nodes;
    std::vector<float> leaves;
    int T = 0, kind = 0;
    float base = 0.f;
    if (gemm_model(ctx, group, nodes, leaves, T, kind, base)) return -1;
    DeviceBuf dn, dl, dr, dout;
    int rc = 0;
    if (upload(ctx, dn, nodes.data(), nodes.size() * sizeof(float2)) || upload(ctx, dl, leaves.data(), leaves.size() * 4) ||
        ensure(dout, (size_t)n * 4))
        rc = -1;
    if (!rc && rows) {
        for (int64_t i = 0; i < n_rows; ++i)
            if (rows[i] < 0 || rows[i] >= ctx->n) { rc = fail("row index out of range"); break; }
        if (!rc && upload(ctx, dr, rows, (size_t)n_rows * 4)) rc = -1;
    }
    if (!rc) {
        GemmArgs g{ctx->x_mat.as<float>(), F, rows ? dr.as<int32_t>() : nullptr, n, dn.as<float2>(), dl.as<float>(),
                   T, kind, base, dout.as<float>()};
        const size_t lds_tables = (size_t)T * 64 * 12;
        // use_mfma: 1 = the round-4 kernel (register-indexed predicates, transposed product), 2 = the round-1 kernel (LDS gathers)
        bool v2 = use_mfma == 1 && !(ctx->kernel_variant & 512);
        // forest_gemm2_kernel reads its predicates' operands through VGPR index mode and relies on where the compiler put the row's
        // 32 feature registers (UGVC_PRED_ASM: checked in the listing of THIS build, not guaranteed by the compiler).  First use per
        // process: the margins of up to 512 rows against the scalar kernel's, bit for bit; on a mismatch the round-1 MFMA kernel
        // (LDS gathers, no register indexing) serves every later call and stderr says so once.
        // (one state PER INSTANTIATION: <true> (RF) and <false> are separately compiled kernels with their own register allocation -
        // validating one says nothing about the other)
        static std::atomic<int> gemm2_states[2];                 // 0 unchecked, 1 agrees, 2 differs, 3 could not be checked (said once)
        std::atomic<int>& gemm2_state = gemm2_states[kind == UGVC_MODEL_RF ? 1 : 0];
        if (v2 && gemm2_state.load() == 0 && lds_tables > 156 * 1024) {
            fprintf(stderr, "[ugvc] forest_gemm2_kernel: the scalar kernel's tables (%zu B) do not fit LDS - the register-indexed predicates of this build run UNCHECKED\n", lds_tables);
            gemm2_state.store(3);
        }
        if (v2 && gemm2_state.load() == 0) {
            const int64_t m = std::min<int64_t>(n, 512);
            DeviceBuf dchk;
            std::vector<float> h((size_t)(2 * m));
            const size_t lds2 = (size_t)T * 64 * 4 + (size_t)(kGemm2Threads / 64) * 64 * kGemm2RowB;
            bool ran = ensure(dchk, (size_t)(2 * m) * 4) == 0 &&
                       hipFuncSetAttribute(reinterpret_cast<const void*>(forest_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024) == hipSuccess &&
                       hipFuncSetAttribute(kind == UGVC_MODEL_RF ? reinterpret_cast<const void*>(forest_gemm2_kernel<true>) : reinterpret_cast<const void*>(forest_gemm2_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024) == hipSuccess;
            if (ran) {
                GemmArgs ga = g, gb = g;
                ga.n = gb.n = m;
                ga.out = dchk.as<float>();
                gb.out = dchk.as<float>() + m;
                UGVC_LAUNCH(forest_rows_kernel, dim3((unsigned)ctx->n_cus * 4), dim3(256), lds_tables, ctx->stream, ga);
                if (kind == UGVC_MODEL_RF) UGVC_LAUNCH(forest_gemm2_kernel<true>, dim3((unsigned)ctx->n_cus * 2), dim3(kGemm2Threads), lds2, ctx->stream, gb);
                else UGVC_LAUNCH(forest_gemm2_kernel<false>, dim3((unsigned)ctx->n_cus * 2), dim3(kGemm2Threads), lds2, ctx->stream, gb);
                ran = copy_out(ctx, h.data(), dchk.p, (size_t)(2 * m) * 4) == hipSuccess &&
                      hipStreamSynchronize(ctx->stream) == hipSuccess && hipGetLastError() == hipSuccess;
            }
            dev_free(dchk.p);
            if (ran) {
                const bool same = memcmp(h.data(), h.data() + m, (size_t)m * 4) == 0;
                gemm2_state.store(same ? 1 : 2);
                if (!same) fprintf(stderr, "[ugvc] forest_gemm2_kernel disagrees with the scalar kernel on this build: using forest_gemm_kernel\n");
            }
        }
        if (gemm2_state.load() == 2) v2 = false;
        const size_t lds = !use_mfma ? lds_tables
                         : v2 ? (size_t)T * 64 * 4 + (size_t)(kGemm2Threads / 64) * 64 * kGemm2RowB
                              : lds_tables + (size_t)(kGemmThreads / 64) * 16 * kGemmXStride * 4;
        if (lds > 156 * 1024) rc = fail("ensemble too large for the LDS-resident GEMM formulation (T * 768 bytes)");
        const void* fn = !use_mfma ? reinterpret_cast<const void*>(forest_rows_kernel)
                       : !v2 ? reinterpret_cast<const void*>(forest_gemm_kernel)
                       : kind == UGVC_MODEL_RF ? reinterpret_cast<const void*>(forest_gemm2_kernel<true>) : reinterpret_cast<const void*>(forest_gemm2_kernel<false>);
        if (!rc) {
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024) != hipSuccess)
                rc = fail("cannot raise the dynamic LDS limit");
        }
        if (!rc) {
            const unsigned grid = (unsigned)ctx->n_cus;
            if (hipEventRecord(ctx->ev0, ctx->stream) != hipSuccess) rc = fail("event record failed");
            for (int it = 0; it < iters && !rc; ++it) {
                if (!use_mfma) UGVC_LAUNCH(forest_rows_kernel, dim3(grid * 4), dim3(256), lds, ctx->stream, g);
                else if (!v2) UGVC_LAUNCH(forest_gemm_kernel, dim3(grid), dim3(kGemmThreads), lds, ctx->stream, g);
                // (two workgroups of eight waves per CU: 110 registers and 66 KB of LDS each)
                else if (kind == UGVC_MODEL_RF) UGVC_LAUNCH(forest_gemm2_kernel<true>, dim3(grid * 2), dim3(kGemm2Threads), lds, ctx->stream, g);
                else UGVC_LAUNCH(forest_gemm2_kernel<false>, dim3(grid * 2), dim3(kGemm2Threads), lds, ctx->stream, g);
            }
            if (!rc && (hipEventRecord(ctx->ev1, ctx->stream) != hipSuccess || hipEventSynchronize(ctx->ev1) != hipSuccess ||
                        hipGetLastError() != hipSuccess))
                rc = fail("forest GEMM launch failed");
            if (!rc && ms_per_launch) {
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
                *ms_per_launch = ms / iters;
            }
        }
        if (!rc && copy_out(ctx, margin_out, dout.p, (size_t)n * 4) != hipSuccess)
            rc = fail("D2H copy failed");
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && !rc) rc = fail("stream sync failed");
    for (DeviceBuf* b : {&dn, &dl, &dr, &dout})
        if (b->p) dev_free(b->p);
    return rc;
}

// All three variant-type groups of the resident feature matrix in ONE launch (forest_gemm3_kernel): rows[k] / n_rows[k] name
// group k's rows of the resident matrix (ascending or not), margin_out[row] receives every named row's margin (other rows are
// left as they were).  Every group with rows must have an additive (XGBoost-style) depth <= 6 ensemble of the same kind.
extern "C" int ugvc_forest_gemm3(ugvc_ctx* ctx, const int32_t* const* rows, const int64_t* n_rows, int iters, float* margin_out,
                                 float* ms_per_launch) {
    if (!ctx || !rows || !n_rows || !margin_out) return fail("NULL argument");
    if (iters < 1) iters = 1;
    UGVC_HIP(hipSetDevice(ctx->device));
    const int F = UGVC_N_BASE_FEATURES + ctx->n_tracks;
    if (!ctx->x_mat.p || ctx->x_mat.cap < (size_t)ctx->n * F * 4) return fail("no resident feature matrix (ugvc_feature_matrix first)");
    Gemm3Args a3;
    memset(&a3, 0, sizeof a3);
    DeviceBuf dn[UGVC_N_GROUPS], dl[UGVC_N_GROUPS], dr[UGVC_N_GROUPS], dout;
    int rc = 0, kind_all = -1, t_max = 0;
    double work[UGVC_N_GROUPS] = {0, 0, 0}, tot = 0.0;
    if (ensure(dout, (size_t)std::max<int64_t>(ctx->n, 1) * 4)) rc = -1;
    if (!rc && copy_in(ctx, dout.p, margin_out, (size_t)ctx->n * 4) != hipSuccess) rc = fail("H2D copy failed");
    for (int k = 0; k < UGVC_N_GROUPS && !rc; ++k) {
        if (n_rows[k] <= 0) continue;
        if (!rows[k]) { rc = fail("NULL row list"); break; }
        for (int64_t i = 0; i < n_rows[k]; ++i)
            if (rows[k][i] < 0 || rows[k][i] >= ctx->n) { rc = fail("row index out of range"); break; }
        if (rc) break;
        std::vector<float2> nodes;
        std::vector<float> leaves;
        int T = 0, kind = 0;
        float base = 0.f;
        if (gemm_model(ctx, k, nodes, leaves, T, kind, base)) { rc = -1; break; }
        if (kind_all >= 0 && kind != kind_all) { rc = fail("ugvc_forest_gemm3: the groups' ensembles must be of one kind"); break; }
        kind_all = kind;
        t_max = std::max(t_max, T);
        if (upload(ctx, dn[k], nodes.data(), nodes.size() * sizeof(float2)) || upload(ctx, dl[k], leaves.data(), leaves.size() * 4) ||
            upload(ctx, dr[k], rows[k], (size_t)n_rows[k] * 4)) { rc = -1; break; }
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = fail("stream sync failed"); break; }   // (nodes / leaves go out of scope)
        a3.g[k] = GemmArgs{ctx->x_mat.as<float>(), F, dr[k].as<int32_t>(), n_rows[k], dn[k].as<float2>(), dl[k].as<float>(), T, kind, base, dout.as<float>()};
        work[k] = (double)n_rows[k] * T;
        tot += work[k];
    }
    if (!rc && tot > 0.0) {
        const size_t lds = (size_t)t_max * 64 * 4 + (size_t)(kGemm2Threads / 64) * 64 * kGemm2RowB;
        if (lds > 156 * 1024) rc = fail("ensemble too large for the LDS-resident GEMM formulation");
        // two workgroups of eight waves per CU, split over the groups by rows x trees (every group with rows gets at least one)
        const int grid = ctx->n_cus * 2;
        int used = 0, big = 0;
        int nbk[UGVC_N_GROUPS] = {0, 0, 0};
        for (int k = 0; k < UGVC_N_GROUPS; ++k) {
            nbk[k] = work[k] > 0 ? std::max(1, (int)(grid * (work[k] / tot) + 0.5)) : 0;
            used += nbk[k];
            if (work[k] > work[big]) big = k;
        }
        nbk[big] = std::max(1, nbk[big] + grid - used);
        int end = 0;
        for (int k = 0; k < UGVC_N_GROUPS; ++k) { end += nbk[k]; a3.wg_end[k] = end; }
        const void* fn = kind_all == UGVC_MODEL_RF ? reinterpret_cast<const void*>(forest_gemm3_kernel<true>) : reinterpret_cast<const void*>(forest_gemm3_kernel<false>);
        if (!rc && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024) != hipSuccess) rc = fail("cannot raise the dynamic LDS limit");
        // first use per process: this kernel shares forest_gemm2_kernel's register-indexed predicates (UGVC_PRED_ASM relies on
        // where the compiler put the row's feature registers) - up to 512 rows of the largest group against the scalar traversal,
        // bit for bit; a build on which they differ fails HERE, loudly, instead of returning margins (ugvc_forest_gemm serves then)
        static std::atomic<int> gemm3_states[2];                 // per instantiation: 0 unchecked, 1 agrees, 2 differs, 3 could not be checked (said once)
        std::atomic<int>& gemm3_state = gemm3_states[kind_all == UGVC_MODEL_RF ? 1 : 0];
        const size_t lds_rows = (size_t)a3.g[big].T * 64 * 12;
        if (!rc && gemm3_state.load() == 0 && lds_rows > 156 * 1024) {
            fprintf(stderr, "[ugvc] forest_gemm3_kernel: the scalar traversal's tables (%zu B) do not fit LDS - the register-indexed predicates of this build run UNCHECKED\n", lds_rows);
            gemm3_state.store(3);
        }
        if (!rc && gemm3_state.load() == 0) {
            const int64_t m = std::min<int64_t>(a3.g[big].n, 512);
            DeviceBuf dchk;
            std::vector<float> h((size_t)m), all((size_t)ctx->n);
            Gemm3Args c3;
            memset(&c3, 0, sizeof c3);
            c3.g[big] = a3.g[big];
            c3.g[big].n = m;
            for (int k = 0; k < UGVC_N_GROUPS; ++k) c3.wg_end[k] = k < big ? 0 : 4;
            GemmArgs ga = a3.g[big];
            ga.n = m;
            bool ran = ensure(dchk, (size_t)m * 4) == 0 &&
                       hipFuncSetAttribute(reinterpret_cast<const void*>(forest_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024) == hipSuccess;
            if (ran) {
                ga.out = dchk.as<float>();
                UGVC_LAUNCH(forest_rows_kernel, dim3(8), dim3(256), lds_rows, ctx->stream, ga);
                if (kind_all == UGVC_MODEL_RF) UGVC_LAUNCH(forest_gemm3_kernel<true>, dim3(4), dim3(kGemm2Threads), lds, ctx->stream, c3);
                else UGVC_LAUNCH(forest_gemm3_kernel<false>, dim3(4), dim3(kGemm2Threads), lds, ctx->stream, c3);
                ran = copy_out(ctx, h.data(), dchk.p, (size_t)m * 4) == hipSuccess &&
                      copy_out(ctx, all.data(), dout.p, (size_t)ctx->n * 4) == hipSuccess &&
                      hipStreamSynchronize(ctx->stream) == hipSuccess && hipGetLastError() == hipSuccess;
            }
            dev_free(dchk.p);
            if (ran) {
                bool same = true;
                for (int64_t i = 0; i < m && same; ++i) same = memcmp(&h[(size_t)i], &all[(size_t)rows[big][i]], 4) == 0;
                if (getenv("UGVC_GEMM3_FORCE_MISMATCH")) same = false;      // (tests: the fallback below)
                gemm3_state.store(same ? 1 : 2);
                if (!same) fprintf(stderr, "[ugvc] forest_gemm3_kernel disagrees with the scalar traversal on this build (register-indexed predicates): "
                                           "the traversal (forest_rows_kernel, one launch per group) serves ugvc_forest_gemm3\n");
            }
        }
        // A build on which the register-indexed predicates read the wrong registers (a compiler update moved the row's feature
        // vector) must not brick config C5's default path: the scalar traversal - bit-identical margins by construction, one launch
        // per group, ~15 % slower - serves every later call (round 5 failed loudly here).
        if (!rc && gemm3_state.load() == 2) {
            for (int k = 0; k < UGVC_N_GROUPS && !rc; ++k)
                if (a3.g[k].n > 0 && (size_t)a3.g[k].T * 64 * 12 > 156 * 1024) rc = fail("ensemble too large for the LDS-resident traversal");
            if (!rc && hipFuncSetAttribute(reinterpret_cast<const void*>(forest_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024) != hipSuccess)
                rc = fail("cannot raise the dynamic LDS limit");
            if (!rc && hipEventRecord(ctx->ev0, ctx->stream) != hipSuccess) rc = fail("event record failed");
            for (int it = 0; it < iters && !rc; ++it)
                for (int k = 0; k < UGVC_N_GROUPS; ++k)
                    if (a3.g[k].n > 0) {
                        GemmArgs gk = a3.g[k];
                        gk.out_by_row = 1;
                        UGVC_LAUNCH(forest_rows_kernel, dim3((unsigned)ctx->n_cus * 4), dim3(256), (size_t)gk.T * 64 * 12, ctx->stream, gk);
                    }
            if (!rc && (hipEventRecord(ctx->ev1, ctx->stream) != hipSuccess || hipEventSynchronize(ctx->ev1) != hipSuccess || hipGetLastError() != hipSuccess))
                rc = fail("forest traversal launch failed");
            if (!rc && ms_per_launch) {
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
                *ms_per_launch = ms / iters;
            }
        } else if (!rc) {
            if (hipEventRecord(ctx->ev0, ctx->stream) != hipSuccess) rc = fail("event record failed");
            for (int it = 0; it < iters && !rc; ++it) {
                if (kind_all == UGVC_MODEL_RF) UGVC_LAUNCH(forest_gemm3_kernel<true>, dim3((unsigned)end), dim3(kGemm2Threads), lds, ctx->stream, a3);
                else UGVC_LAUNCH(forest_gemm3_kernel<false>, dim3((unsigned)end), dim3(kGemm2Threads), lds, ctx->stream, a3);
            }
            if (!rc && (hipEventRecord(ctx->ev1, ctx->stream) != hipSuccess || hipEventSynchronize(ctx->ev1) != hipSuccess || hipGetLastError() != hipSuccess))
                rc = fail("forest GEMM launch failed");
            if (!rc && ms_per_launch) {
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
                *ms_per_launch = ms / iters;
            }
        }
    }
    if (!rc && ctx->n > 0 && copy_out(ctx, margin_out, dout.p, (size_t)ctx->n * 4) != hipSuccess) rc = fail("D2H copy failed");
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && !rc) rc = fail("stream sync failed");
    for (int k = 0; k < UGVC_N_GROUPS; ++k)
        for (DeviceBuf* b : {&dn[k], &dl[k], &dr[k]}) dev_free(b->p);
    dev_free(dout.p);
    return rc;
}
