// micro-benchmark: how fast can a CU pull L2-resident data into LDS with global_load_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(1))) const void gvoid;
typedef __attribute__((address_space(3))) void lvoid;
template <int NT>
__global__ void __launch_bounds__(NT) ingest(const char* src, size_t span, int iters, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    constexpr int NW = NT / 64;
    size_t base = ((size_t)blockIdx.x * 65536) % span;
    for (int it = 0; it < iters; it++) {
        // each wave loads 8 x 1 KiB pieces per iteration (64 KiB per block-iteration at 8 waves)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const char* p = src + (base + (size_t)((j * NW + wave) * 1024 + lane * 16)) % span;
            __builtin_amdgcn_global_load_lds((gvoid*)p, (lvoid*)(lds + ((it & 1) * NW * 8 + j * NW + wave) * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        base = (base + NW * 8192) % span;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = lds[5];
}
int main() {
    const size_t span = 2 << 20;   // 2 MiB working set: L2 resident
    char* d; int* sink;
    hipMalloc(&d, span + (1 << 20)); hipMemset(d, 1, span + (1 << 20)); hipMalloc(&sink, 4096 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int nt : {256, 512}) for (int bpc : {1, 2}) {
        int blocks = 256 * bpc, iters = 2000;
        int smem = 2 * (nt / 64) * 8 * 1024;
        auto k = nt == 256 ? ingest<256> : ingest<512>;
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(nt), smem, 0, d, span, 10, sink);
        hipEventRecord(a);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(nt), smem, 0, d, span, iters, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        double bytes = (double)blocks * iters * (nt / 64) * 8192.0;
        printf("threads %d blocks/CU %d : %.1f GB/s per CU, %.2f TB/s chip (%s)\n", nt, bpc, bytes / ms / 1e6 / 256, bytes / ms / 1e9, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
