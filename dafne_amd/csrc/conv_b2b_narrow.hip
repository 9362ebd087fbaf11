// Bottleneck tail + next bottleneck head in one kernel for the NARROW stage res2 of ResNet-50/101 (gfx950)
// [detectron2 BottleneckBlock, recalled; built by build_dafne_resnet_fpn_backbone, backbone/fpn.py:58-91]:
//
//     Y = relu(conv3_b(T) + bias3 + X)            1x1, 64 -> 256, X = the block's shortcut (identity or projection output)
//     Z = relu(conv1_{b+1}(Y) + bias1)            1x1, 256 -> 64
//
// Unlike res4 (conv_b2b.hip: 1 MB of weights, matrix-pipe / ingest bound) this pair is a pure STREAMING problem:
// 64 KB of weights, 1.3 flop per byte of activation, and unfused the 268-MB map Y (batch 8, 256 x 256) is written by
// conv3 and read straight back by conv1.  So the structure is the opposite of conv_b2b's hand-counted ring:
//   * persistent workgroups (one per CU, 8 waves), BOTH weight matrices live in registers for the whole kernel
//     (conv3: the wave's 32 output channels x K = 64 -> 4 A fragments; conv1: output-channel half `wave & 1` x K = 256
//     -> 16 A fragments);
//   * a workgroup walks 128-pixel tiles; the loads of tile k+1 -- T by DMA into the other half of a 2 x 16-KB LDS ring,
//     the shortcut rows X into 8 registers per lane -- are issued at the start of tile k and awaited at the start of
//     tile k+1, so 80 KB per CU are in flight during all of tile k's work; the stores of tile k (Y: 64 KB, Z: 16 KB) are
//     younger than those loads, and `s_waitcnt vmcnt(10)` = "everything but the last tile's ten stores" never waits for
//     a store (gfx950 counts stores in vmcnt, in order);
//   * tile k: X rows -> the Y buffer in LDS; GEMM1 (T tile as B operand); (acc + bias3) + X -> ReLU -> bf16 in place
//     (conv_b2b's epilogue); Y rows -> HBM as whole 512-B pixel rows; GEMM2 with the Y buffer as B operand (wave =
//     output-channel half x 32-pixel quarter); (acc + bias1) -> ReLU -> bf16 -> a 16-KB staging tile -> Z rows.
// K is walked in ascending order in both GEMMs and the epilogue expressions are those of the separate kernels: results
// are bit-identical to conv3 (+residual) followed by conv1 (tests/test_gpu_conv.py).
// HBM per 128-pixel tile: 16 (T) + 64 (X) in, 64 (Y) + 16 (Z) out = 160 KB against 224 KB for the two launches.
//
// PROJ (block 0 of the stage): the shortcut is itself a 1x1 convolution of the block's 64-channel input X0 (projection,
// no ReLU).  Then X is never materialised: the X0 tile rides the LDS ring next to T (16 KB instead of 64 KB of shortcut
// rows), a third resident weight matrix gives S = bf16(Wsc . X0 + bias_sc) -- rounded to bf16 exactly as the separate
// launch stores it -- and the epilogue adds it from registers.  112 KB per tile against 304 KB for the three launches.
#include <stdlib.h>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((address_space(1))) void gvoid;
typedef __attribute__((address_space(3))) void lvoid;

constexpr int kPx = 128;                 // pixels per tile
constexpr int kSlab = kPx * 128;         // [128 px][64 channels]: 16 KB
constexpr int kCM = 64, kCB = 256;
constexpr int kOffT = 0;                 // 2 x 16 KB ring of T tiles
constexpr int kOffY = 2 * kSlab;         // 4 slabs: the Y tile (shortcut rows first, updated in place)
constexpr int kOffZ = kOffY + 4 * kSlab; // Z staging tile
constexpr int kOffBias = kOffZ + kSlab;  // fp32 [256 conv3 | 64 conv1 | 256 projection]
constexpr int kOffX0 = kOffBias + 4096;  // PROJ: 2 x 16 KB ring of X0 tiles
constexpr int kSmemTotal = kOffBias + 4096;
constexpr int kSmemTotalProj = kOffX0 + 2 * kSlab;
static_assert((2 * kCB + kCM) * 4 <= 4096 && kSmemTotalProj <= 160 * 1024, "LDS budget");
constexpr int kNW = 8, kNT = 512;

struct NarrowDev {
    const char* in;      // bf16 [N, H+2, W+2, 64]
    const char* res;     // bf16 [N, H+2, W+2, 256]; PROJ: the block input X0 [N, H+2, W+2, 64]
    const char* wf;      // bf16 [8 waves][4 steps][64 lanes][8] (conv3) | [2 halves][16 steps][64 lanes][8] (conv1) | PROJ: [8][4][64][8] (projection)
    const float* b3;     // [256]
    const float* b1;     // [64]
    const float* bsc;    // PROJ: [256]
    char* out;           // bf16 [N, H+2, W+2, 256]
    char* next;          // bf16 [N, H+2, W+2, 64]
    int N, H, W, tiles_per_img, tiles;
};

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    f32x2 v = {a, b};
    bf16x2 r = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(unsigned, r);
}

template <bool PROJ>
__global__ void __launch_bounds__(512, 2) conv_b2b_narrow_kernel(NarrowDev P) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, half = lane >> 5;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const int HW = P.H * P.W;
    const int Wp = P.W + 2;
    const float invW = 1.0f / (float)P.W;
    const int G = gridDim.x;
    const int my_tiles = (P.tiles - (int)blockIdx.x + G - 1) / G;

    // haloed pixel index of pixel px of tile t (clamped into the image: rows past the end of a ragged tile alias the last one)
    auto halo_index = [&](int t, int px) {
        const int img = t / P.tiles_per_img;
        int m = (t - img * P.tiles_per_img) * kPx + px;
        m = m < HW ? m : HW - 1;
        const int ho = (int)(((float)m + 0.5f) * invW), wo = m - ho * P.W;    // exact for H*W <= 2^20
        return (unsigned)((img * (P.H + 2) + ho + 1) * Wp + wo + 1);
    };

    // ---- weights -> registers (once); biases -> LDS
    bf16x8 a3[4], a1[16];
    {
        const char* w3 = P.wf + (size_t)wave * 4 * 1024 + lane * 16;
#pragma unroll
        for (int s = 0; s < 4; s++) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a3[s]) : "v"(w3 + s * 1024) : "memory");
        const char* w1 = P.wf + 8 * 4 * 1024 + (size_t)(wave & 1) * 16 * 1024 + lane * 16;
#pragma unroll
        for (int s = 0; s < 16; s++) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a1[s]) : "v"(w1 + s * 1024) : "memory");
        if (tid < kCB + kCM) ((float*)(lds + kOffBias))[tid] = tid < kCB ? P.b3[tid] : P.b1[tid - kCB];
        if (PROJ && tid < kCB) ((float*)(lds + kOffBias))[kCB + kCM + tid] = P.bsc[tid];
    }
    bf16x8 asc[PROJ ? 4 : 1];
    if (PROJ) {
        const char* ws = P.wf + (8 * 4 + 2 * 16) * 1024 + (size_t)wave * 4 * 1024 + lane * 16;
#pragma unroll
        for (int s = 0; s < 4; s++) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(asc[s]) : "v"(ws + s * 1024) : "memory");
    }

    // ---- per-tile loads: T by DMA (16 pieces of 8 px x 128 B; wave w moves pieces w and w + 8), X rows into registers
    u32x4 rr[PROJ ? 1 : 8];
#pragma unroll
    for (int i = 0; i < (PROJ ? 1 : 8); i++) rr[i] = u32x4{0u, 0u, 0u, 0u};
    auto issue_loads = [&](int t, int buf) {
#pragma unroll
        for (int ii = 0; ii < 2; ii++) {
            const int px = (wave + kNW * ii) * 8 + (lane >> 3);
            const unsigned q = (unsigned)(((lane & 7) ^ ((px >> 1) & 7)) * 16);
            const size_t hp = (size_t)halo_index(t, px) * (kCM * 2) + q;
            __builtin_amdgcn_global_load_lds((gvoid*)(P.in + hp), (lvoid*)(lds + kOffT + buf * kSlab + (wave + kNW * ii) * 1024), 16, 0, 0);
            if (PROJ)
                __builtin_amdgcn_global_load_lds((gvoid*)(P.res + hp), (lvoid*)(lds + kOffX0 + buf * kSlab + (wave + kNW * ii) * 1024), 16, 0, 0);
        }
        if (!PROJ) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int idx = tid + kNT * i;
                const char* src = P.res + (size_t)halo_index(t, idx >> 5) * (kCB * 2) + (idx & 31) * 16;
                // "+v": the destination stays the register that carries rr[i] around the tile loop (with "=v" the compiler
                // may load into a fresh register and copy it right behind the still outstanding load)
                asm volatile("global_load_dwordx4 %0, %1, off nt" : "+v"(rr[i]) : "v"(src) : "memory");
            }
        }
    };

    unsigned bs[4];                          // B fragment of k16 step s inside a slab, pixel fragment 0
#pragma unroll
    for (int s = 0; s < 4; s++) bs[s] = (unsigned)(frow * 128 + (((2 * s + half) ^ ((frow >> 1) & 7)) * 16));
    auto barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    const unsigned lbias = lds_base + (unsigned)kOffBias;
    // epilogue 1: the wave's 32 channels are half (wave & 1) of slab (wave >> 1) of the Y buffer
    const unsigned ebase = lds_base + (unsigned)(kOffY + (wave >> 1) * kSlab + frow * 128 + 8 * half);
    const int ct = wave & 1, pt = wave >> 1;                       // GEMM2: output-channel half, 32-pixel quarter

    if (my_tiles > 0) issue_loads((int)blockIdx.x, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < 4; s++) asm volatile("" : "+v"(a3[s]));
#pragma unroll
    for (int s = 0; s < 16; s++) asm volatile("" : "+v"(a1[s]));
    if (PROJ) {
#pragma unroll
        for (int s = 0; s < 4; s++) asm volatile("" : "+v"(asc[s]));
    }

    for (int kk = 0; kk < my_tiles; kk++) {
        const int t = (int)blockIdx.x + kk * G;
        const int buf = kk & 1;
        const int img = t / P.tiles_per_img;
        const int plast = HW - 1 - (t - img * P.tiles_per_img) * kPx;
        // ---- 1. this tile's loads have landed: only the previous tile's 8 + 2 stores are younger
        if (kk > 0) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        // ---- 2. shortcut rows -> Y buffer ([slab][px][64 ch], 16-byte chunk ^ ((px >> 1) & 7))
        if (!PROJ) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                asm volatile("" : "+v"(rr[i]));
                const int idx = tid + kNT * i;
                const int px = idx >> 5, j = idx & 31;
                const unsigned ad = lds_base + (unsigned)(kOffY + (j >> 3) * kSlab + px * 128 + (((j & 7) ^ ((px >> 1) & 7)) * 16));
                asm volatile("ds_write_b128 %0, %1" ::"v"(ad), "v"(rr[i]) : "memory");
            }
        }
        barrier();       // T tile + shortcut rows visible; every wave is done with the previous tile's LDS reads
        // ---- 3. next tile's loads
        if (kk + 1 < my_tiles) issue_loads(t + G, buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        // ---- 4a. PROJ: S = bf16(Wsc . X0 + bias_sc), kept as packed bf16 pairs in the accumulator layout
        u32x2 sres[PROJ ? 4 : 1][4];
        if (PROJ) {
            f32x16 accs[4];
#pragma unroll
            for (int b = 0; b < 4; b++)
#pragma unroll
                for (int k = 0; k < 16; k++) accs[b][k] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; s++) {
                bf16x8 bfr[4];
#pragma unroll
                for (int b = 0; b < 4; b++) bfr[b] = *(const bf16x8*)(lds + kOffX0 + buf * kSlab + b * 4096 + bs[s]);
#pragma unroll
                for (int b = 0; b < 4; b++) accs[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(asc[s], bfr[b], accs[b], 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const float* bp = (const float*)(lds + kOffBias) + kCB + kCM + wave * 32 + 8 * g + 4 * half;
                const float c0 = bp[0], c1 = bp[1], c2 = bp[2], c3 = bp[3];
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    sres[g][b].x = pack_bf16(accs[b][4 * g] + c0, accs[b][4 * g + 1] + c1);
                    sres[g][b].y = pack_bf16(accs[b][4 * g + 2] + c2, accs[b][4 * g + 3] + c3);
                }
            }
        }
        // ---- 4. GEMM1: Y (32 channels of this wave x 128 px) = W3 . T
        f32x16 acc1[4];
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int k = 0; k < 16; k++) acc1[b][k] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            bf16x8 bfr[4];
#pragma unroll
            for (int b = 0; b < 4; b++) bfr[b] = *(const bf16x8*)(lds + kOffT + buf * kSlab + b * 4096 + bs[s]);
#pragma unroll
            for (int b = 0; b < 4; b++) acc1[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[s], bfr[b], acc1[b], 0, 0, 0);
        }
        // ---- 5. (acc + bias3) + X -> ReLU -> bf16, in place in the Y buffer (each wave touches only its own 32 channels)
        {
            typedef __attribute__((ext_vector_type(4))) float f32x4;
            typedef __attribute__((ext_vector_type(2))) float f32x2;
#pragma unroll
            for (int gp = 0; gp < 2; gp++) {
                f32x4 bv[2];
                u32x2 rc[2][4];
                unsigned ead[2];
#pragma unroll
                for (int gg = 0; gg < 2; gg++) {
                    const int g = 2 * gp + gg;
                    ead[gg] = ebase + (unsigned)(((((wave & 1) * 4 + g) ^ ((frow >> 1) & 7))) * 16);
                    const unsigned bad = lbias + (unsigned)((wave * 32 + 8 * g + 4 * half) * 4);
                    if (PROJ) {
                        asm volatile("ds_read_b128 %0, %1" : "=&v"(bv[gg]) : "v"(bad) : "memory");
#pragma unroll
                        for (int b = 0; b < 4; b++) rc[gg][b] = sres[g][b];
                    } else {
                        asm volatile("ds_read_b128 %4, %6\n\tds_read_b64 %0, %5\n\tds_read_b64 %1, %5 offset:4096\n\t"
                                     "ds_read_b64 %2, %5 offset:8192\n\tds_read_b64 %3, %5 offset:12288"
                                     : "=&v"(rc[gg][0]), "=&v"(rc[gg][1]), "=&v"(rc[gg][2]), "=&v"(rc[gg][3]), "=&v"(bv[gg])
                                     : "v"(ead[gg]), "v"(bad)
                                     : "memory");
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(rc[0][0]), "+v"(rc[0][1]), "+v"(rc[0][2]), "+v"(rc[0][3]), "+v"(rc[1][0]), "+v"(rc[1][1]),
                               "+v"(rc[1][2]), "+v"(rc[1][3]), "+v"(bv[0]), "+v"(bv[1])
                             :
                             : "memory");
#pragma unroll
                for (int gg = 0; gg < 2; gg++) {
                    const int g = 2 * gp + gg;
                    const f32x2 blo = {bv[gg][0], bv[gg][1]}, bhi = {bv[gg][2], bv[gg][3]};
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const u32x2 r = rc[gg][b];
                        const f32x2 rlo = {__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u)};
                        const f32x2 rhi = {__uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
                        const f32x2 alo = {acc1[b][4 * g], acc1[b][4 * g + 1]}, ahi = {acc1[b][4 * g + 2], acc1[b][4 * g + 3]};
                        const f32x2 vlo = alo + blo + rlo, vhi = ahi + bhi + rhi;          // (acc + bias) + residual
                        rc[gg][b].x = pack_bf16(fmaxf(vlo[0], 0.f), fmaxf(vlo[1], 0.f));
                        rc[gg][b].y = pack_bf16(fmaxf(vhi[0], 0.f), fmaxf(vhi[1], 0.f));
                    }
                    asm volatile("ds_write_b64 %4, %0\n\tds_write_b64 %4, %1 offset:4096\n\t"
                                 "ds_write_b64 %4, %2 offset:8192\n\tds_write_b64 %4, %3 offset:12288"
                                 ::"v"(rc[gg][0]), "v"(rc[gg][1]), "v"(rc[gg][2]), "v"(rc[gg][3]), "v"(ead[gg]) : "memory");
                }
            }
        }
        barrier();       // the Y tile is complete
        // ---- 6. Y rows -> HBM: pass i of 8, 32 threads write one pixel's 512 B (exactly 8 stores per lane)
#pragma unroll
        for (int i = 0; i < 8; i++) {
            int idx = tid + kNT * i;
            asm volatile("" : "+v"(idx));
            int px = idx >> 5;
            px = px < plast ? px : plast;
            const int j = idx & 31;
            const u32x4 v = *(const u32x4*)(lds + kOffY + (j >> 3) * kSlab + px * 128 + (((j & 7) ^ ((px >> 1) & 7)) * 16));
            __builtin_nontemporal_store(v, (u32x4*)(P.out + (size_t)halo_index(t, px) * (kCB * 2) + j * 16));
        }
        // ---- 7. GEMM2: Z (channel half ct x pixel quarter pt) = W1 . Y
        f32x16 acc2;
#pragma unroll
        for (int k = 0; k < 16; k++) acc2[k] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            bf16x8 bfr[4];
#pragma unroll
            for (int s = 0; s < 4; s++) bfr[s] = *(const bf16x8*)(lds + kOffY + q * kSlab + pt * 4096 + bs[s]);
#pragma unroll
            for (int s = 0; s < 4; s++) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[4 * q + s], bfr[s], acc2, 0, 0, 0);
        }
        // ---- 8. (acc + bias1) -> ReLU -> bf16 -> Z staging tile
        {
            const unsigned zb = lds_base + (unsigned)(kOffZ + (pt * 32 + frow) * 128 + 8 * half);
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const float* bp = (const float*)(lds + kOffBias) + kCB + ct * 32 + 8 * g + 4 * half;
                const float v0 = fmaxf(acc2[4 * g] + bp[0], 0.f), v1 = fmaxf(acc2[4 * g + 1] + bp[1], 0.f);
                const float v2 = fmaxf(acc2[4 * g + 2] + bp[2], 0.f), v3 = fmaxf(acc2[4 * g + 3] + bp[3], 0.f);
                u32x2 pk;
                pk.x = pack_bf16(v0, v1);
                pk.y = pack_bf16(v2, v3);
                const unsigned ad = zb + (unsigned)((((ct * 4 + g) ^ ((frow >> 1) & 7))) * 16);
                asm volatile("ds_write_b64 %0, %1" ::"v"(ad), "v"(pk) : "memory");
            }
        }
        barrier();
        // ---- 9. Z rows -> HBM: 8 threads write one pixel's 128 B (exactly 2 stores per lane)
#pragma unroll
        for (int i = 0; i < 2; i++) {
            int idx = tid + kNT * i;
            asm volatile("" : "+v"(idx));
            int px = idx >> 3;
            px = px < plast ? px : plast;
            const int q = idx & 7;
            const u32x4 v = *(const u32x4*)(lds + kOffZ + px * 128 + ((q ^ ((px >> 1) & 7)) * 16));
            *(u32x4*)(P.next + (size_t)halo_index(t, px) * (kCM * 2) + q * 16) = v;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

}  // namespace

extern "C" {

static int narrow_launch(const void* d_in, const void* d_res, const void* d_wfrag, const float* d_bias3, const float* d_bias_sc,
                         const float* d_bias1, int n_images, int H, int W, void* d_out, void* d_next, void* stream, bool proj) {
    if (!d_in || !d_res || !d_wfrag || !d_bias3 || !d_bias1 || !d_out || !d_next || (proj && !d_bias_sc)) return dafne::fail(DAFNE_E_INVALID, "bottleneck_tail_head_narrow: null argument");
    if (n_images < 1 || H < 1 || W < 1 || (long long)H * W > (1 << 20)) return dafne::fail(DAFNE_E_INVALID, "bottleneck_tail_head_narrow: bad size");
    NarrowDev D;
    D.in = (const char*)d_in; D.res = (const char*)d_res; D.wf = (const char*)d_wfrag; D.b3 = d_bias3; D.b1 = d_bias1; D.bsc = d_bias_sc;
    D.out = (char*)d_out; D.next = (char*)d_next;
    D.N = n_images; D.H = H; D.W = W;
    D.tiles_per_img = (H * W + kPx - 1) / kPx;
    const long long tiles = (long long)D.tiles_per_img * n_images;
    if (tiles > (1ll << 24)) return dafne::fail(DAFNE_E_UNSUPPORTED, "bottleneck_tail_head_narrow: too many tiles");
    D.tiles = (int)tiles;
    DAFNE_MAX_LDS_ONCE(kSmemTotal, (const void*)conv_b2b_narrow_kernel<false>);
    DAFNE_MAX_LDS_ONCE(kSmemTotalProj, (const void*)conv_b2b_narrow_kernel<true>);
    int n_cu = 0;
    if (int rc = dafne::device_cus(&n_cu)) return rc;
    static const int cap = getenv("DAFNE_STREAM_GRID") ? atoi(getenv("DAFNE_STREAM_GRID")) : 0;
    const int lim = cap > 0 && cap < n_cu ? cap : n_cu;
    const int grid = D.tiles < lim ? D.tiles : lim;
    if (proj) hipLaunchKernelGGL(conv_b2b_narrow_kernel<true>, dim3(grid), dim3(kNT), kSmemTotalProj, (hipStream_t)stream, D);
    else hipLaunchKernelGGL(conv_b2b_narrow_kernel<false>, dim3(grid), dim3(kNT), kSmemTotal, (hipStream_t)stream, D);
    return dafne::check_launch("conv_b2b_narrow");
}

int dafne_bottleneck_tail_head_narrow_hip(const void* d_in, const void* d_res, const void* d_wfrag, const float* d_bias3,
                                          const float* d_bias1, int n_images, int H, int W, void* d_out, void* d_next,
                                          void* stream) {
    return narrow_launch(d_in, d_res, d_wfrag, d_bias3, nullptr, d_bias1, n_images, H, W, d_out, d_next, stream, false);
}

int dafne_bottleneck_proj_tail_head_narrow_hip(const void* d_in, const void* d_x0, const void* d_wfrag, const float* d_bias3,
                                               const float* d_bias_sc, const float* d_bias1, int n_images, int H, int W,
                                               void* d_out, void* d_next, void* stream) {
    return narrow_launch(d_in, d_x0, d_wfrag, d_bias3, d_bias_sc, d_bias1, n_images, H, W, d_out, d_next, stream, true);
}

}  // extern "C"
