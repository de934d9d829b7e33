// Probe of global_load_lds semantics on gfx950: does the instruction's immediate offset move the LDS
// destination as well as the global source?  (Decides how the v4 kernel lays out its DMA'd window planes.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define LDSP __attribute__((address_space(3)))
#define GLBP __attribute__((address_space(1)))
__global__ void probe(const uint32_t* __restrict__ src, uint32_t* out) {
    __shared__ uint32_t lds[1024];
    const int tid = threadIdx.x;
    for (int k = tid; k < 1024; k += 64) lds[k] = 0xdeadbeefu;
    __syncthreads();
    // lane l reads src[l * 16 + 2] (offset 8 bytes) ; LDS base = lds + 128 dwords
    __builtin_amdgcn_global_load_lds((const GLBP void*)(src + tid * 16), (LDSP void*)(lds + 128), 4, 8, 0);
    __syncthreads();
    for (int k = tid; k < 1024; k += 64) out[k] = lds[k];
}
int main() {
    uint32_t h[1024], *d, *o, r[1024];
    for (int i = 0; i < 1024; ++i) h[i] = i;
    hipMalloc(&d, 4096); hipMalloc(&o, 4096);
    hipMemcpy(d, h, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o);
    hipMemcpy(r, o, 4096, hipMemcpyDeviceToHost);
    int first = -1;
    for (int i = 0; i < 1024; ++i) if (r[i] != 0xdeadbeefu) { first = i; break; }
    printf("first written dword index %d (128 = offset not applied to LDS, 130 = applied), value %u (expect 2), next %u (expect 18)\n",
           first, first >= 0 ? r[first] : 0, first >= 0 ? r[first + 1] : 0);
    return 0;
}
