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
    unsigned off = (unsigned)(size_t)(__attribute__((address_space(3))) char*)tile + threadIdx.x * 16;
    for (int i = 0; i < iters; i++) {            // (asm: plain C++ reads of eight repeating addresses are hoisted out of the loop)
        u32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[k]) : "v"(off), "n"(k * 8192));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
        for (int k = 0; k < 8; k++) acc ^= v[k];
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

// MFMA with operands that CHANGE from one instruction to the next (kind 3: eight random register pairs in rotation; kind 4: the same
// with half of the B elements zero, as behind a ReLU) and the 16x16x32 shape (kind 5: constant operands, kind 6: rotating): what a matrix
// instruction costs depends on how many operand bits toggle, and a real kernel never feeds the same registers twice.
// (operands arrive as whole registers from a host-made bf16 table `tab` [2][8][64 lanes][8]: part 0 randn, part 1 the same behind a ReLU;
// the instructions are asm so that the rotation stays as written)
template <bool ZEROS>
__global__ void __launch_bounds__(512) mfma_rot(int iters, const bf16x8* __restrict__ tab, float* out) {
    const int lane = threadIdx.x & 63;
    bf16x8 a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { a[j] = tab[j * 64 + lane]; b[j] = tab[(ZEROS ? 512 : 0) + ((j + 3) & 7) * 64 + (lane ^ 21)]; }
    f32x16 c[4];
#pragma unroll
    for (int r = 0; r < 4; r++) for (int k = 0; k < 16; k++) c[r][k] = 0.f;
    for (int i = 0; i < iters; i += 2) {
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c[j & 3]) : "v"(a[j]), "v"(b[j]));
    }
    float s = 0.f;
    for (int r = 0; r < 4; r++) for (int k = 0; k < 16; k++) s += c[r][k];
    if (s == 1234.5f) out[0] = s;
}
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <bool ROT>
__global__ void __launch_bounds__(512) mfma16(int iters, const bf16x8* __restrict__ tab, float* out) {
    const int lane = threadIdx.x & 63;
    bf16x8 a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { a[j] = tab[(ROT ? j : 0) * 64 + lane]; b[j] = tab[(ROT ? (j + 3) & 7 : 3) * 64 + (lane ^ 21)]; }
    f32x4 c[8];
#pragma unroll
    for (int r = 0; r < 8; r++) for (int k = 0; k < 4; k++) c[r][k] = 0.f;
    for (int i = 0; i < iters; i++) {            // 8 x 16384 flop = 4 x 32768: the same flops per iteration as mfma_only
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c[j]) : "v"(a[j]), "v"(b[j]));
    }
    float s = 0.f;
    for (int r = 0; r < 8; r++) for (int k = 0; k < 4; k++) s += c[r][k];
    if (s == 1234.5f) out[0] = s;
}
// kind 7: 32x32x16 with ONE constant operand pair through the same asm (the baseline of kinds 3 / 4: 4 accumulators in rotation)
__global__ void __launch_bounds__(512) mfma_const_asm(int iters, const bf16x8* __restrict__ tab, float* out) {
    const int lane = threadIdx.x & 63;
    bf16x8 a = tab[lane], b = tab[3 * 64 + (lane ^ 21)];
    f32x16 c[4];
#pragma unroll
    for (int r = 0; r < 4; r++) for (int k = 0; k < 16; k++) c[r][k] = 0.f;
    for (int i = 0; i < iters; i += 2) {
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c[j & 3]) : "v"(a), "v"(b));
    }
    float s = 0.f;
    for (int r = 0; r < 4; r++) for (int k = 0; k < 16; k++) s += c[r][k];
    if (s == 1234.5f) out[0] = s;
}

extern "C" int probe_run(int kind, const void* w, unsigned region, int iters, int blocks, void* out, const void* seed, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (kind == 0) hipLaunchKernelGGL(l2_stream, dim3(blocks), dim3(512), 0, st, (const char*)w, region, iters, (unsigned*)out);
    else if (kind == 1) hipLaunchKernelGGL(lds_stream, dim3(blocks), dim3(512), 0, st, iters, (unsigned*)out);
    else if (kind == 3) hipLaunchKernelGGL(mfma_rot<false>, dim3(blocks), dim3(512), 0, st, iters, (const bf16x8*)w, (float*)out);
    else if (kind == 4) hipLaunchKernelGGL(mfma_rot<true>, dim3(blocks), dim3(512), 0, st, iters, (const bf16x8*)w, (float*)out);
    else if (kind == 5) hipLaunchKernelGGL(mfma16<false>, dim3(blocks), dim3(512), 0, st, iters, (const bf16x8*)w, (float*)out);
    else if (kind == 6) hipLaunchKernelGGL(mfma16<true>, dim3(blocks), dim3(512), 0, st, iters, (const bf16x8*)w, (float*)out);
    else if (kind == 7) hipLaunchKernelGGL(mfma_const_asm, dim3(blocks), dim3(512), 0, st, iters, (const bf16x8*)w, (float*)out);
    else hipLaunchKernelGGL(mfma_only, dim3(blocks), dim3(512), 0, st, iters, (const float*)seed, (float*)out);
    return (int)hipGetLastError();
}
