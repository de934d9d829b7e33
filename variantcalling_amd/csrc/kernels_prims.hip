// Stable LSD radix sort of (u64, u32) pairs + prefix sums, hand-written for gfx950 (see ugvc_prims.hpp).
//
// Sort: 8-bit digits, up to 8 passes.  One census kernel counts all eight digit positions at once (LDS histograms): a
// pass whose digit is shared by every key is skipped - locus keys (contig << 32 | pos) differ in ~4 of 8 bytes, so do
// scores confined to [0, 1].  A real pass is three launches:
//   histogram   a block owns 4096 consecutive pairs; its 256-bin histogram (LDS atomics) goes to hist[digit][block];
//   scan        exclusive prefix sum over hist in digit-major order = where each (digit, block) bucket starts;
//   scatter     the block re-reads its pairs in order, 256 at a time: a wave ranks its 64 keys among equal digits with
//               eight ballots (the lanes that agree on all eight bits are a key's peers; its rank is the number of
//               peers below it), waves of a round are ordered through their per-digit counts in LDS, rounds through
//               a running per-digit count - so equal keys keep their input order (stable), which is what makes the
//               next digit's pass correct and what the PR curve's tie rule relies on.
// Scan: blocked 4096-word tiles (16 per thread), wave shuffles + one LDS hop for the block total, tile totals scanned
// recursively, offsets added back.
#include "ugvc_prims.hpp"

namespace ugvc {

constexpr int kSortThreads = 256;
constexpr int kSortRounds = 16;
constexpr int kSortTile = kSortThreads * kSortRounds;      // pairs per block

__global__ __launch_bounds__(kSortThreads) void sort_census_kernel(const uint64_t* __restrict__ keys, int64_t n, unsigned long long* census) {
    __shared__ unsigned h[8][256];
    for (int q = threadIdx.x; q < 8 * 256; q += blockDim.x) (&h[0][0])[q] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t k = keys[i];
#pragma unroll
        for (int p = 0; p < 8; ++p) atomicAdd(&h[p][(k >> (8 * p)) & 255u], 1u);
    }
    __syncthreads();
    for (int q = threadIdx.x; q < 8 * 256; q += blockDim.x) {
        const unsigned c = (&h[0][0])[q];
        if (c) atomicAdd(&census[q], (unsigned long long)c);
    }
}

__global__ __launch_bounds__(kSortThreads) void sort_hist_kernel(const uint64_t* __restrict__ keys, int64_t n, int shift, uint32_t* __restrict__ hist,
                                                                 int n_blocks) {
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kSortTile;
#pragma unroll 4
    for (int r = 0; r < kSortRounds; ++r) {
        const int64_t i = base + r * kSortThreads + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * n_blocks + blockIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(kSortThreads) void sort_scatter_kernel(const uint64_t* __restrict__ k_in, const uint32_t* __restrict__ v_in,
                                                                    uint64_t* __restrict__ k_out, uint32_t* __restrict__ v_out, int64_t n, int shift,
                                                                    const uint32_t* __restrict__ hist, int n_blocks) {
    __shared__ unsigned start[256];                      // where this block's bucket of every digit begins, plus what earlier rounds put there
    __shared__ unsigned wc[kSortThreads / 64][256];      // this round's count per wave and digit
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    start[tid] = hist[(size_t)tid * n_blocks + blockIdx.x];
    const unsigned long long below = (1ull << lane) - 1;
    const int64_t base = (int64_t)blockIdx.x * kSortTile;
    for (int r = 0; r < kSortRounds; ++r) {
        if (base + (int64_t)r * kSortThreads >= n) break;                     // (uniform)
#pragma unroll
        for (int w = 0; w < kSortThreads / 64; ++w) wc[w][tid] = 0;
        __syncthreads();
        const int64_t i = base + r * kSortThreads + tid;
        const bool live = i < n;
        const uint64_t k = live ? k_in[i] : 0;
        const uint32_t v = live ? v_in[i] : 0;
        const unsigned d = (unsigned)(k >> shift) & 255u;
        unsigned long long peers = __builtin_amdgcn_ballot_w64(live);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(bit);
            peers &= bit ? m : ~m;
        }
        const unsigned rank = (unsigned)__popcll(peers & below);
        if (live && rank == 0) wc[wave][d] = (unsigned)__popcll(peers);
        __syncthreads();
        if (live) {
            unsigned off = start[d] + rank;
            for (int w = 0; w < wave; ++w) off += wc[w][d];
            k_out[off] = k;
            v_out[off] = v;
        }
        __syncthreads();
        unsigned add = 0;
#pragma unroll
        for (int w = 0; w < kSortThreads / 64; ++w) add += wc[w][tid];
        start[tid] += add;
        __syncthreads();
    }
}

// ---- scans ------------------------------------------------------------------------------------------
constexpr int kScanItems = 16;
constexpr int kScanTile = kSortThreads * kScanItems;

__device__ __forceinline__ uint64_t wave_inclusive_u64(uint64_t x, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)x, d), hi = (uint32_t)__shfl_up((int)(uint32_t)(x >> 32), d);
        if (lane >= d) x += ((uint64_t)hi << 32) | lo;
    }
    return x;
}

// every tile scanned on its own (exclusive within the tile, or inclusive); totals[tile] = the tile's sum
__global__ __launch_bounds__(kSortThreads) void scan_tiles_kernel(uint64_t* __restrict__ data, int64_t n, int inclusive, uint64_t* __restrict__ totals) {
    __shared__ uint64_t wsum[kSortThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)tid * kScanItems;
    uint64_t x[kScanItems], s = 0;
#pragma unroll
    for (int q = 0; q < kScanItems; ++q) {
        x[q] = base + q < n ? data[base + q] : 0;
        s += x[q];
    }
    const uint64_t inc = wave_inclusive_u64(s, lane);
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint64_t before = inc - s;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    uint64_t run = before;
#pragma unroll
    for (int q = 0; q < kScanItems; ++q) {
        const uint64_t v = x[q];
        if (base + q < n) data[base + q] = inclusive ? run + v : run;
        run += v;
    }
    if (tid == kSortThreads - 1) totals[blockIdx.x] = run;
}

__global__ void scan_add_kernel(uint64_t* __restrict__ data, int64_t n, const uint64_t* __restrict__ tile_off) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) data[i] += tile_off[i / kScanTile];
}

__global__ void widen_u32_kernel(const uint32_t* __restrict__ in, uint64_t* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}
__global__ void narrow_u32_kernel(const uint64_t* __restrict__ in, uint32_t* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint32_t)in[i];
}

static int scan_rec(ugvc_ctx* ctx, uint64_t* data, int64_t n, bool inclusive, uint64_t* scratch, size_t scratch_words) {
    const int64_t tiles = (n + kScanTile - 1) / kScanTile;
    if ((size_t)tiles > scratch_words) return fail("internal: scan scratch too small");
    UGVC_LAUNCH(scan_tiles_kernel, dim3((unsigned)tiles), dim3(kSortThreads), 0, ctx->stream, data, n, inclusive ? 1 : 0, scratch);
    if (tiles > 1) {
        if (scan_rec(ctx, scratch, tiles, false, scratch + tiles, scratch_words - (size_t)tiles)) return -1;
        UGVC_LAUNCH(scan_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, data, n, scratch);
    }
    return 0;
}

static size_t scan_scratch_words(int64_t n) {
    size_t w = 0;
    while (n > 1) {
        n = (n + kScanTile - 1) / kScanTile;
        w += (size_t)n;
        if (n == 1) break;
    }
    return w + 8;
}

int scan_u64(ugvc_ctx* ctx, DeviceBuf& tmp, uint64_t* data, int64_t n, bool inclusive) {
    if (n <= 0) return 0;
    const size_t words = scan_scratch_words(n);
    if (ensure(tmp, words * 8)) return -1;
    if (scan_rec(ctx, data, n, inclusive, tmp.as<uint64_t>(), words)) return -1;
    UGVC_HIP(hipGetLastError());
    return 0;
}

int radix_sort_pairs_u64(ugvc_ctx* ctx, DeviceBuf& tmp, uint64_t* k0, uint64_t* k1, uint32_t* v0, uint32_t* v1, int64_t n,
                         uint64_t** k_out, uint32_t** v_out) {
    *k_out = k0;
    *v_out = v0;
    if (n <= 1) return 0;
    if (n >= ((int64_t)1 << 32)) return fail("radix sort: n out of range");
    const int n_blocks = (int)((n + kSortTile - 1) / kSortTile);
    const size_t hist_words = (size_t)256 * n_blocks;                         // u32 histogram, scanned as u64 words
    const size_t scan_words = scan_scratch_words((int64_t)hist_words);
    // tmp: census (8 x 256 u64) | hist u32 | hist as u64 | scan scratch
    const size_t off_hist = 8 * 256 * 8, off_wide = off_hist + ((hist_words * 4 + 15) & ~(size_t)15);
    const size_t off_scan = off_wide + hist_words * 8;
    if (ensure(tmp, off_scan + scan_words * 8)) return -1;
    uint8_t* t = static_cast<uint8_t*>(tmp.p);
    unsigned long long* census = reinterpret_cast<unsigned long long*>(t);
    uint32_t* hist = reinterpret_cast<uint32_t*>(t + off_hist);
    uint64_t* wide = reinterpret_cast<uint64_t*>(t + off_wide);
    uint64_t* scratch = reinterpret_cast<uint64_t*>(t + off_scan);
    UGVC_HIP(hipMemsetAsync(census, 0, 8 * 256 * 8, ctx->stream));
    const unsigned cgrid = (unsigned)std::min<int64_t>((n + kSortThreads - 1) / kSortThreads, (int64_t)ctx->n_cus * 8);
    UGVC_LAUNCH(sort_census_kernel, dim3(cgrid), dim3(kSortThreads), 0, ctx->stream, k0, n, census);
    unsigned long long h[8 * 256];
    UGVC_HIP(copy_out(ctx, h, census, sizeof(h)));
    UGVC_HIP(hipStreamSynchronize(ctx->stream));
    uint64_t* ki = k0; uint64_t* ko = k1;
    uint32_t* vi = v0; uint32_t* vo = v1;
    for (int p = 0; p < 8; ++p) {
        bool trivial = false;
        for (int d = 0; d < 256; ++d)
            if (h[p * 256 + d] == (unsigned long long)n) { trivial = true; break; }
        if (trivial) continue;
        const int shift = 8 * p;
        UGVC_LAUNCH(sort_hist_kernel, dim3((unsigned)n_blocks), dim3(kSortThreads), 0, ctx->stream, ki, n, shift, hist, n_blocks);
        const unsigned g = (unsigned)((hist_words + 255) / 256);
        UGVC_LAUNCH(widen_u32_kernel, dim3(g), dim3(256), 0, ctx->stream, hist, wide, (int64_t)hist_words);
        if (scan_rec(ctx, wide, (int64_t)hist_words, false, scratch, scan_words)) return -1;
        UGVC_LAUNCH(narrow_u32_kernel, dim3(g), dim3(256), 0, ctx->stream, wide, hist, (int64_t)hist_words);
        UGVC_LAUNCH(sort_scatter_kernel, dim3((unsigned)n_blocks), dim3(kSortThreads), 0, ctx->stream, ki, vi, ko, vo, n, shift, hist, n_blocks);
        std::swap(ki, ko);
        std::swap(vi, vo);
    }
    UGVC_HIP(hipGetLastError());
    *k_out = ki;
    *v_out = vi;
    return 0;
}

}  // namespace ugvc
