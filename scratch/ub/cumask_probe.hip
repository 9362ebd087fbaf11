// Which CUs / XCDs does a CU-masked stream use?  hipExtStreamCreateWithCUMask + a kernel that records HW_REG_XCC_ID / HW_ID.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
__global__ void probe(unsigned* out) {
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    // spin a little so that workgroups spread over all allowed CUs
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 20000) {}
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = xcc; out[blockIdx.x * 2 + 1] = hwid; }
}
int main(int argc, char** argv) {
    const int nwords = 8;
    unsigned mask[nwords] = {0};
    for (int i = 0; i < nwords && i + 1 < argc; i++) mask[i] = strtoul(argv[i + 1], 0, 16);
    hipStream_t st;
    if (hipExtStreamCreateWithCUMask(&st, nwords, mask) != hipSuccess) { printf("create failed\n"); return 1; }
    const int n = 4096;
    unsigned* d; hipMalloc(&d, n * 8);
    hipLaunchKernelGGL(probe, dim3(n), dim3(64), 0, st, d);
    hipStreamSynchronize(st);
    unsigned* h = (unsigned*)malloc(n * 8);
    hipMemcpy(h, d, n * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::map<unsigned, int>> cnt;
    for (int i = 0; i < n; i++) cnt[h[2 * i] & 0xf][(h[2 * i + 1] >> 8) & 0xfff]++;      // xcc -> (cu/sh/se bits of HW_ID) -> count
    for (auto& x : cnt) printf("xcc %u: %zu distinct CUs\n", x.first, x.second.size());
    return 0;
}
