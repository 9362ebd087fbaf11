// Which packed fp32 instruction form returns wrong lanes beside matrix kernels on gfx950?  One kernel per form: every thread runs a
// dependent chain of ITER instructions on its own four inputs (pure function of the input).  scripts/pk_probe.py runs them beside
// torch.matmul on three streams and compares with the idle result.
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(2))) float f32x2;
#define ITER 256
template <int KIND>
__global__ void __launch_bounds__(256) pk_kernel(const float4* __restrict__ in, float4* __restrict__ out, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float4 v = in[i];
    f32x2 a = {v.x, v.y}, b = {v.z, v.w}, c = {v.y, v.z};
#pragma unroll 8
    for (int k = 0; k < ITER; k++) {
        f32x2 r;
        if (KIND == 0) {          // scalar ops only
            float r0, r1;
            asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %3, %5" : "=&v"(r0), "=&v"(r1) : "v"(a.x), "v"(a.y), "v"(b.x), "v"(b.y));
            r = (f32x2){r0, r1};
        } else if (KIND == 1) {   // plain packed multiply
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=&v"(r) : "v"(a), "v"(b));
        } else if (KIND == 2) {   // packed multiply with a cross-half swizzle
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(r) : "v"(a), "v"(b));
        } else if (KIND == 3) {   // packed add with negated second operand
            asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=&v"(r) : "v"(a), "v"(b));
        } else if (KIND == 4) {   // packed fma, broadcast low half of src0, negated addend
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=&v"(r) : "v"(a), "v"(b), "v"(c));
        } else if (KIND == 5) {   // packed move with half selection
            asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=&v"(r) : "v"(a), "v"(b));
        } else {                  // plain packed fma
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=&v"(r) : "v"(a), "v"(b), "v"(c));
        }
        // keep the values bounded and the chain dependent
        c = b; b = a;
        a = (f32x2){r.x * 0.5f + 0.25f, r.y * 0.5f - 0.25f};
        a.x = fminf(fmaxf(a.x, -4.f), 4.f); a.y = fminf(fmaxf(a.y, -4.f), 4.f);
    }
    out[i] = make_float4(a.x, a.y, b.x, b.y);
}
extern "C" int pk_run(int kind, const void* in, void* out, int n, void* stream) {
    dim3 g((n + 255) / 256), b(256);
    hipStream_t st = (hipStream_t)stream;
    switch (kind) {
#define C(K) case K: hipLaunchKernelGGL(pk_kernel<K>, g, b, 0, st, (const float4*)in, (float4*)out, n); break;
        C(0) C(1) C(2) C(3) C(4) C(5) C(6)
    }
    return (int)hipGetLastError();
}
