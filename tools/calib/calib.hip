// FETCH_SIZE / WRITE_SIZE calibration on known byte counts (MI355X_MICROARCH.md, HBM section:
// "calibrate on a known byte count in your own access pattern before trusting an absolute").
// Four kernels over a 1 GiB buffer (bigger than the 256 MiB Infinity Cache):
//   calib_read4   coalesced 4 B/lane streaming read      (how the variant columns are read)
//   calib_read16  coalesced 16 B/lane streaming read
//   calib_gather  3 x 16 B/lane at sorted pseudo-random 16-B aligned offsets, one 48-B window per
//                 lane, windows ~620 B apart on average (how the reference windows are read)
//   calib_write4  coalesced 4 B/lane streaming write
// Build: hipcc --offload-arch=gfx950 -O3 -o calib calib.hip ; run under rocprofv3 --pmc FETCH_SIZE.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <random>
#include <algorithm>

__global__ void calib_read4(const uint32_t* __restrict__ p, size_t n, uint32_t* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void calib_read16(const uint4* __restrict__ p, size_t n, uint32_t* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 x = p[i]; acc ^= x.x ^ x.y ^ x.z ^ x.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void calib_gather(const uint8_t* __restrict__ base, const uint32_t* __restrict__ off16, size_t n, uint32_t* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4* s = reinterpret_cast<const uint4*>(base + (size_t)off16[i] * 16);
    uint4 a = s[0], b = s[1], c = s[2];
    uint32_t acc = a.x ^ a.w ^ b.y ^ c.z;
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void calib_write4(uint32_t* __restrict__ p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}

int main() {
    const size_t bytes = 1ull << 30;
    uint8_t* buf; uint32_t* out; uint32_t* offs;
    hipMalloc(&buf, bytes + 64); hipMalloc(&out, 64);
    hipMemset(buf, 1, bytes + 64);
    const size_t ng = 1600000;                      // 1.6 M windows over 1 GiB ~ one per 670 B
    std::vector<uint32_t> h(ng);
    std::mt19937_64 rng(7);
    for (auto& x : h) x = (uint32_t)(rng() % ((bytes - 64) / 16));
    std::sort(h.begin(), h.end());
    hipMalloc(&offs, ng * 4); hipMemcpy(offs, h.data(), ng * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(calib_read4, dim3(4096), dim3(256), 0, 0, (const uint32_t*)buf, bytes / 4, out);
        hipLaunchKernelGGL(calib_read16, dim3(4096), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, out);
        hipLaunchKernelGGL(calib_gather, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, 0, buf, offs, ng, out);
        hipLaunchKernelGGL(calib_write4, dim3(4096), dim3(256), 0, 0, (uint32_t*)buf, bytes / 4);
    }
    hipDeviceSynchronize();
    printf("calib: read4/read16/write4 move %zu bytes each; gather reads %zu windows x 48 B = %zu bytes "
           "(sector-granular estimate %zu bytes at 64 B)\n", bytes, ng, ng * 48, ng * 112);
    return 0;
}
