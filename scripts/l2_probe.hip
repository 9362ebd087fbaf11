// energy / rate probes: (0) stream an L2-resident region into registers from every CU (the weight-stream pattern of the resident-patch
// kernels: each wave walks 1-KiB fragments of a `region`-byte array, all CUs the same array); (1) the same bytes from LDS (ds_read_b128
// of a resident 64-KB tile); (2) MFMA only (registers).  Looped by the host for seconds while hwmon power is sampled.
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void __launch_bounds__(512) l2_stream(const char* __restrict__ w, unsigned region, int iters, unsigned* out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    u32x4 acc = {0, 0, 0, 0};
    unsigned off = (unsigned)(wave * 1024 * 16 + lane * 16) % region;
    for (int i = 0; i < iters; i++) {
        u32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            v[k] = *(const u32x4*)(w + off);
            off += 8 * 1024;                 // 8 waves x 1 KiB per step
            off = off >= region ? off - region : off;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) acc ^= v[k];
    }
    if (acc.x == 0x12345678u) out[0] = acc.y ^ acc.z ^ acc.w;
}

__global__ void __launch_bounds__(512) lds_stream(int iters, unsigned* out) {
    __shared__ __attribute__((aligned(16))) char tile[65536];
    for (int i = threadIdx.x; i < 65536 / 16; i += 512) ((u32x4*)tile)[i] = u32x4{(unsigned)i * 2654435761u, (unsigned)i, 7u, (unsigned)~i};
    __syncthreads();
    u32x4 acc = {0, 0, 0, 0};
    unsigned off = threadIdx.x * 16;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            acc ^= *(const u32x4*)(tile + off);
            off = (off + 8192) & 65535;
        }
    }
    if (acc.x == 0x12345678u) out[0] = acc.y ^ acc.z ^ acc.w;
}

__global__ void __launch_bounds__(512) mfma_only(int iters, const float* seed, float* out) {
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int k = 0; k < 8; k++) { a[k] = (__bf16)(seed[(lane * 8 + k) & 1023]); b[k] = (__bf16)(seed[(lane * 8 + k + 512) & 1023]); }
    f32x16 c[4];
    for (int r = 0; r < 4; r++) for (int k = 0; k < 16; k++) c[r][k] = 0.f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 4; r++) c[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[r], 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 4; r++) for (int k = 0; k < 16; k++) s += c[r][k];
    if (s == 1234.5f) out[0] = s;
}

extern "C" int probe_run(int kind, const void* w, unsigned region, int iters, int blocks, void* out, const void* seed, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (kind == 0) hipLaunchKernelGGL(l2_stream, dim3(blocks), dim3(512), 0, st, (const char*)w, region, iters, (unsigned*)out);
    else if (kind == 1) hipLaunchKernelGGL(lds_stream, dim3(blocks), dim3(512), 0, st, iters, (unsigned*)out);
    else hipLaunchKernelGGL(mfma_only, dim3(blocks), dim3(512), 0, st, iters, (const float*)seed, (float*)out);
    return (int)hipGetLastError();
}
