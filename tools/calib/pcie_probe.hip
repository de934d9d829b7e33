// Host link rates the pipelined boundary (csrc/pipeline.hip) can hope for: pinned H2D as one 200 MB copy, as 96 pieces of
// ~2 MB on one and on two streams, with a concurrent 30 MB D2H, and a pageable H2D for comparison.
// hipcc -O3 --offload-arch=gfx950 pcie_probe.hip -o pcie_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const size_t N = 200u << 20, M = 30u << 20;
    void *hp, *hq, *d, *d2;
    (void)hipHostMalloc(&hp, N, hipHostMallocDefault);
    (void)hipHostMalloc(&hq, M, hipHostMallocDefault);
    void* pg = malloc(N);
    memset(hp, 1, N); memset(pg, 2, N);
    (void)hipMalloc(&d, N); (void)hipMalloc(&d2, M);
    hipStream_t s0, s1, s2;
    (void)hipStreamCreateWithFlags(&s0, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    auto rep = [&](const char* what, size_t bytes, auto&& f) {
        f(); (void)hipDeviceSynchronize();
        double best = 1e9;
        for (int r = 0; r < 5; ++r) { const double t0 = now(); f(); (void)hipDeviceSynchronize(); best = std::min(best, now() - t0); }
        printf("%-64s %7.2f ms  %6.1f GB/s\n", what, best * 1e3, bytes / best / 1e9);
    };
    rep("H2D pinned, one 200 MB copy", N, [&] { (void)hipMemcpyAsync(d, hp, N, hipMemcpyHostToDevice, s0); });
    rep("H2D pinned, 96 pieces, one stream", N, [&] { for (int k = 0; k < 96; ++k) (void)hipMemcpyAsync((char*)d + k * (N / 96), (char*)hp + k * (N / 96), N / 96, hipMemcpyHostToDevice, s0); });
    rep("H2D pinned, 96 pieces, two streams", N, [&] { for (int k = 0; k < 96; ++k) (void)hipMemcpyAsync((char*)d + k * (N / 96), (char*)hp + k * (N / 96), N / 96, hipMemcpyHostToDevice, (k & 1) ? s1 : s0); });
    rep("H2D pinned 200 MB + D2H pinned 30 MB at the same time", N + M, [&] { (void)hipMemcpyAsync(d, hp, N, hipMemcpyHostToDevice, s0); (void)hipMemcpyAsync(hq, d2, M, hipMemcpyDeviceToHost, s2); });
    rep("D2H pinned, one 30 MB copy", M, [&] { (void)hipMemcpyAsync(hq, d2, M, hipMemcpyDeviceToHost, s2); });
    rep("H2D pageable, one 200 MB copy", N, [&] { (void)hipMemcpyAsync(d, pg, N, hipMemcpyHostToDevice, s0); });
    rep("host memcpy 200 MB, one thread", N, [&] { memcpy(hp, pg, N); });
    return 0;
}
