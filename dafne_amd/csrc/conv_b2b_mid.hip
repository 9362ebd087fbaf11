// Bottleneck tail + next bottleneck head in one kernel for res3 of ResNet-50/101 (gfx950)
// [detectron2 BottleneckBlock, recalled; built by build_dafne_resnet_fpn_backbone, backbone/fpn.py:58-91]:
//
//     Y = relu(conv3_b(T) + bias3 + X)            1x1, 128 -> 512, X = the block's shortcut
//     Z = relu(conv1_{b+1}(Y) + bias1)            1x1, 512 -> 128
//
// Like res2 (conv_b2b_narrow.hip) this pair is HBM-bound (320 KB of activations per 128 pixels against 1 MFLOP per KB), but its
// 256 KB of weights fit neither the registers nor the LDS, so they stream L2 -> LDS while the activations stream
// HBM -> LDS / registers -- and the two streams must not share a `vmcnt` queue: the counter retires in order, so a weight
// piece issued behind the next tile's HBM loads would only be "ready" when those are.  vmcnt is per WAVE, hence a role
// split inside the persistent workgroup (8 waves, one workgroup per CU; all eight do the matrix work):
//   * waves 0-3 ("memory waves") issue every HBM access: the T tile of the next tile by DMA, the shortcut rows X into
//     registers (a whole tile ahead: chunk 0's rows at the start of a tile, chunk 1's in the middle), the Y and Z row stores.
//     Their waits count only their own, younger stores: `vmcnt(24)` / `vmcnt(32)` never wait for a store;
//   * waves 4-7 ("weight waves") issue only the weight DMAs: 16-KB quarter blocks, fragment-major (the LDS image IS the
//     A-operand layout: one conflict-free ds_read_b128 per fragment), through a ring of four, three blocks ahead.
// A tile is 64 pixels; per 256-channel chunk c of Y: shortcut rows -> Y buffer, GEMM1 in two 128-channel halves (K = 128,
// T tile as B operand) each followed by conv_b2b's in-place epilogue, Y rows -> HBM, GEMM2 over the chunk's K range in two
// halves with the Y buffer as B operand (accumulators stay in registers across both chunks); then Z through a staging tile.
// Every wave owns one 32 x 32 MFMA tile (channel quarter `wave & 3`, pixel half `wave >> 2`) in every block.
// K ascends in both GEMMs and the epilogue expressions are those of the separate kernels: bit-identical to conv3
// (+residual) followed by conv1 (tests/test_gpu_conv.py).  HBM per 64-pixel tile: 16 (T) + 64 (X) in, 64 (Y) + 16 (Z) out.
#include <type_traits>

#include <stdlib.h>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((address_space(1))) void gvoid;
typedef __attribute__((address_space(3))) void lvoid;

constexpr int kPx = 64;                  // pixels per tile
constexpr int kSlab = kPx * 128;         // [64 px][64 channels]: 8 KB
constexpr int kCM = 128, kCB = 512;
constexpr int kOffT = 0;                 // 2 x (2 slabs): ring of T tiles, 32 KB
constexpr int kOffY = 4 * kSlab;         // 4 slabs: one 256-channel chunk of Y
constexpr int kOffZ = kOffY + 4 * kSlab; // 2 slabs: Z staging
constexpr int kQB = 16 * 1024;           // weight quarter block: [4 channel quarters][4 k16 steps][64 lanes][16 B]
constexpr int kOffW = kOffZ + 2 * kSlab; // ring of 4 quarter blocks
constexpr int kOffBias = kOffW + 4 * kQB;          // fp32 [512 conv3 | 128 conv1]
constexpr int kSmemTotal = kOffBias + (kCB + kCM) * 4;
constexpr int kNT = 512;
constexpr int kQPT = 16;                 // quarter blocks per tile
static_assert(kSmemTotal <= 160 * 1024, "LDS budget");

struct MidDev {
    const char* in;      // bf16 [N, H+2, W+2, 128]
    const char* res;     // bf16 [N, H+2, W+2, 512]
    const char* wf;      // bf16 [16 quarter blocks][4][4][64][8] (engine.pack_b2b_mid)
    const float* b3;     // [512]
    const float* b1;     // [128]
    char* out;           // bf16 [N, H+2, W+2, 512]
    char* next;          // bf16 [N, H+2, W+2, 128]
    int N, H, W, tiles_per_img, tiles;
};

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    f32x2 v = {a, b};
    bf16x2 r = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(unsigned, r);
}

template <int N>
__device__ __forceinline__ void vm_le() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

__global__ void __launch_bounds__(512, 2) conv_b2b_mid_kernel(MidDev P) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool memw = wave < 4;               // memory wave / weight wave
    const int frow = lane & 31, half = lane >> 5;
    const int ct = wave & 3, pt = wave >> 2;  // this wave's MFMA tile: channel quarter, pixel half
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const int HW = P.H * P.W;
    const int Wp = P.W + 2;
    const float invW = 1.0f / (float)P.W;
    const int G = gridDim.x;
    const int my_tiles = (P.tiles - (int)blockIdx.x + G - 1) / G;

    auto halo_index = [&](int t, int px) {
        const int img = t / P.tiles_per_img;
        int m = (t - img * P.tiles_per_img) * kPx + px;
        m = m < HW ? m : HW - 1;
        const int ho = (int)(((float)m + 0.5f) * invW), wo = m - ho * P.W;    // exact for H*W <= 2^20
        return (unsigned)((img * (P.H + 2) + ho + 1) * Wp + wo + 1);
    };

    if (tid < kCB + kCM) ((float*)(lds + kOffBias))[tid] = tid < kCB ? P.b3[tid] : P.b1[tid - kCB];
    if (tid + kNT < kCB + kCM) ((float*)(lds + kOffBias))[tid + kNT] = P.b1[tid + kNT - kCB];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- memory waves: T tile by DMA (16 pieces of 8 px x 128 B: 2 slabs x 8; wave w moves pieces w, w+4, w+8, w+12) and the
    // shortcut rows of one 256-channel chunk into 8 registers per lane (pass i: pixel (mt + 256 i) >> 5, 16-byte piece & 31)
    const int mt = tid & 255;
    u32x4 rr[2][8];
#pragma unroll
    for (int i = 0; i < 8; i++) rr[0][i] = rr[1][i] = u32x4{0u, 0u, 0u, 0u};
    auto issue_T = [&](int t, int buf) {
#pragma unroll
        for (int ii = 0; ii < 4; ii++) {
            const int piece = wave + 4 * ii;
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int px = (piece & 7) * 8 + (ln >> 3);
            const unsigned q = (unsigned)(((ln & 7) ^ ((px >> 1) & 7)) * 16);
            __builtin_amdgcn_global_load_lds((gvoid*)(P.in + (size_t)halo_index(t, px) * (kCM * 2) + (piece >> 3) * 128 + q),
                                             (lvoid*)(lds + kOffT + buf * 2 * kSlab + piece * 1024), 16, 0, 0);
        }
    };
    auto issue_rows = [&](int t, int c, u32x4 (&r)[8]) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            int idx = mt + 256 * i;
            asm volatile("" : "+v"(idx));            // address recomputed at the point of use, not carried across the tile
            const char* src = P.res + (size_t)halo_index(t, idx >> 5) * (kCB * 2) + c * 512 + (idx & 31) * 16;
            // "+v": the destination is tied to the register that already holds r[i] -- with "=v" the compiler may pick a fresh
            // register and COPY it into the loop-carried one right behind the (still outstanding) load
            asm volatile("global_load_dwordx4 %0, %1, off nt" : "+v"(r[i]) : "v"(src) : "memory");
        }
    };
    auto rows_to_lds = [&](u32x4 (&r)[8]) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            asm volatile("" : "+v"(r[i]));
            int idx = mt + 256 * i;
            asm volatile("" : "+v"(idx));
            const int px = idx >> 5, j = idx & 31;
            const unsigned ad = lds_base + (unsigned)(kOffY + (j >> 3) * kSlab + px * 128 + (((j & 7) ^ ((px >> 1) & 7)) * 16));
            asm volatile("ds_write_b128 %0, %1" ::"v"(ad), "v"(r[i]) : "memory");
        }
    };
    // ---- weight waves: quarter block q (global sequence) -> ring slot q & 3; 16 pieces of 1 KB, 4 per wave
    auto issue_qb = [&](int q) {                                   // q = block number inside a tile (any tile: same weights)
        unsigned voff = (unsigned)lane * 16u;
        asm volatile("" : "+v"(voff));           // opaque per call: otherwise all 64 source addresses are hoisted out of the tile loop and spilled
        const char* src = P.wf + (size_t)(q & (kQPT - 1)) * kQB;
#pragma unroll
        for (int ii = 0; ii < 4; ii++) {
            const int piece = (wave - 4) + 4 * ii;
            __builtin_amdgcn_global_load_lds((gvoid*)(src + piece * 1024 + voff), (lvoid*)(lds + kOffW + (q & 3) * kQB + piece * 1024), 16, 0, 0);
        }
    };

    unsigned bs[4];
#pragma unroll
    for (int s = 0; s < 4; s++) bs[s] = (unsigned)((pt * 32 + frow) * 128 + (((2 * s + half) ^ ((frow >> 1) & 7)) * 16));
    auto barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    const unsigned lbias = lds_base + (unsigned)kOffBias;
    // position of this lane's 4-channel group g of its MFMA tile inside a [slab][px][64 ch] buffer whose channel 0 is
    // channel `c0` of the tile's 128-channel half: slab (c0 + ct*32) >> 6, 16-byte piece ((ct & 1) * 4 + g) ^ swizzle
    const unsigned tile_off = (unsigned)((ct >> 1) * kSlab + (pt * 32 + frow) * 128 + 8 * half);

    const int t0 = (int)blockIdx.x;
    if (my_tiles > 0) {
        if (memw) {
            issue_T(t0, 0);
            issue_rows(t0, 0, rr[0]);
            issue_rows(t0, 1, rr[1]);
        } else {
#pragma unroll
            for (int q = 0; q < 3; q++) issue_qb(q);
        }
    }

    for (int kk = 0; kk < my_tiles; kk++) {
        const int t = t0 + kk * G;
        const int buf = kk & 1;
        const bool has_next = kk + 1 < my_tiles;
        const int img = t / P.tiles_per_img;
        const int plast = HW - 1 - (t - img * P.tiles_per_img) * kPx;
        f32x16 acc1, acc2;
#pragma unroll
        for (int k = 0; k < 16; k++) acc2[k] = 0.f;

        // fully unrolled: every condition on h below is a compile-time one (a rolled loop made the compiler hoist the
        // tile's load / store addresses out of it, spill them, and drain vmcnt at every reload)
        static_for<0, kQPT>([&](auto hc) {
            constexpr int h = decltype(hc)::value;
            constexpr int c = h >> 3;             // Y chunk
            constexpr bool g2 = (h >> 2) & 1;     // GEMM2 block
            constexpr int rp = (h >> 1) & 1;      // GEMM1: 128-channel half of the chunk; GEMM2: K half of the chunk
            constexpr int sh = h & 1;             // which 4 of the block's 8 k16 steps
            // ---- readiness of what this quarter block needs, then the workgroup barrier
            if (memw) {
                if constexpr (h == 0) {           // T tile + chunk 0 rows of THIS tile (issued a tile ago)
                    if (kk == 0) vm_le<8>(); else vm_le<24>();
                }
                if constexpr (h == 8) {           // chunk 1 rows: younger = [Y stores c1 + Z of the previous tile] [next T + rows] Y stores c0
                    if (kk == 0) { if (has_next) vm_le<20>(); else vm_le<8>(); }
                    else { if (has_next) vm_le<32>(); else vm_le<20>(); }
                }
            } else {
                // block h of this tile has landed: the (existing) blocks h+1, h+2 of the sequence are younger
                if (h + 2 < kQPT || has_next) vm_le<8>(); else if (h + 1 < kQPT) vm_le<4>(); else vm_le<0>();
            }
            barrier();
            // ---- issue: weights three blocks ahead; memory traffic of this phase
            if (!memw) {
                if (h + 3 < kQPT || has_next) issue_qb(h + 3);
            } else {
                if constexpr (h == 0) {
                    if (kk > 0) {                                  // Z rows of the previous tile: 64 px x 256 B, 4 stores per lane
                        const int tp = t - G;
                        const int imgp = tp / P.tiles_per_img;
                        const int plastp = HW - 1 - (tp - imgp * P.tiles_per_img) * kPx;
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            int idx = mt + 256 * i;
                            asm volatile("" : "+v"(idx));
                            int px = idx >> 4;
                            px = px < plastp ? px : plastp;
                            const int j = idx & 15;
                            const u32x4 v = *(const u32x4*)(lds + kOffZ + (j >> 3) * kSlab + px * 128 + (((j & 7) ^ ((px >> 1) & 7)) * 16));
                            *(u32x4*)(P.next + (size_t)halo_index(tp, px) * (kCM * 2) + j * 16) = v;
                        }
                    }
                    rows_to_lds(rr[0]);
                    if (has_next) {
                        issue_T(t + G, buf ^ 1);
                        issue_rows(t + G, 0, rr[0]);
                    }
                }
                if constexpr (h == 8) {
                    rows_to_lds(rr[1]);
                    if (has_next) issue_rows(t + G, 1, rr[1]);
                }
                if constexpr (h == 4 || h == 12) {                           // the chunk is complete: Y rows -> HBM, 8 stores per lane
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        int idx = mt + 256 * i;
                        asm volatile("" : "+v"(idx));
                        int px = idx >> 5;
                        px = px < plast ? px : plast;
                        const int j = idx & 31;
                        const u32x4 v = *(const u32x4*)(lds + kOffY + (j >> 3) * kSlab + px * 128 + (((j & 7) ^ ((px >> 1) & 7)) * 16));
                        __builtin_nontemporal_store(v, (u32x4*)(P.out + (size_t)halo_index(t, px) * (kCB * 2) + c * 512 + j * 16));
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- 4 k16 steps
            if constexpr (!g2 && sh == 0) {
#pragma unroll
                for (int k = 0; k < 16; k++) acc1[k] = 0.f;
            }
            {
                const char* wq = lds + kOffW + (h & 3) * kQB + ct * 4096 + lane * 16;
                const char* bsrc = g2 ? lds + kOffY + (2 * rp + sh) * kSlab : lds + kOffT + buf * 2 * kSlab + sh * kSlab;
                bf16x8 af[4], bf[4];
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    af[s] = *(const bf16x8*)(wq + s * 1024);
                    bf[s] = *(const bf16x8*)(bsrc + bs[s]);
                }
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    if constexpr (g2) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], bf[s], acc2, 0, 0, 0);
                    else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], bf[s], acc1, 0, 0, 0);
                }
            }
            // ---- epilogue 1 after the second half of a GEMM1 block: (acc + bias3) + X -> ReLU -> bf16, in place in the Y buffer
            if constexpr (!g2 && sh == 1) {
                typedef __attribute__((ext_vector_type(4))) float f32x4;
                u32x2 rc[4];
                f32x4 bv[4];
                unsigned ead[4];
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    ead[g] = lds_base + (unsigned)(kOffY + 2 * rp * kSlab) + tile_off + (unsigned)(((((ct & 1) * 4 + g) ^ ((frow >> 1) & 7))) * 16);
                    const unsigned bad = lbias + (unsigned)((c * 256 + rp * 128 + ct * 32 + 8 * g + 4 * half) * 4);
                    asm volatile("ds_read_b64 %0, %2\n\tds_read_b128 %1, %3" : "=&v"(rc[g]), "=&v"(bv[g]) : "v"(ead[g]), "v"(bad) : "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(rc[0]), "+v"(rc[1]), "+v"(rc[2]), "+v"(rc[3]), "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3])
                             :
                             : "memory");
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const u32x2 r = rc[g];
                    const float v0 = fmaxf((acc1[4 * g] + bv[g][0]) + __uint_as_float(r.x << 16), 0.f);
                    const float v1 = fmaxf((acc1[4 * g + 1] + bv[g][1]) + __uint_as_float(r.x & 0xffff0000u), 0.f);
                    const float v2 = fmaxf((acc1[4 * g + 2] + bv[g][2]) + __uint_as_float(r.y << 16), 0.f);
                    const float v3 = fmaxf((acc1[4 * g + 3] + bv[g][3]) + __uint_as_float(r.y & 0xffff0000u), 0.f);
                    u32x2 pk;
                    pk.x = pack_bf16(v0, v1);
                    pk.y = pack_bf16(v2, v3);
                    asm volatile("ds_write_b64 %0, %1" ::"v"(ead[g]), "v"(pk) : "memory");
                }
            }
            // ---- epilogue 2 after the tile's last block: (acc + bias1) -> ReLU -> bf16 -> Z staging tile
            if constexpr (h == kQPT - 1) {
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const float* bp = (const float*)(lds + kOffBias) + kCB + ct * 32 + 8 * g + 4 * half;
                    const float v0 = fmaxf(acc2[4 * g] + bp[0], 0.f), v1 = fmaxf(acc2[4 * g + 1] + bp[1], 0.f);
                    const float v2 = fmaxf(acc2[4 * g + 2] + bp[2], 0.f), v3 = fmaxf(acc2[4 * g + 3] + bp[3], 0.f);
                    u32x2 pk;
                    pk.x = pack_bf16(v0, v1);
                    pk.y = pack_bf16(v2, v3);
                    const unsigned ad = lds_base + (unsigned)kOffZ + tile_off + (unsigned)(((((ct & 1) * 4 + g) ^ ((frow >> 1) & 7))) * 16);
                    asm volatile("ds_write_b64 %0, %1" ::"v"(ad), "v"(pk) : "memory");
                }
            }
        });
    }
    // ---- Z rows of the last tile
    barrier();
    if (memw && my_tiles > 0) {
        const int tp = t0 + (my_tiles - 1) * G;
        const int imgp = tp / P.tiles_per_img;
        const int plastp = HW - 1 - (tp - imgp * P.tiles_per_img) * kPx;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int idx = mt + 256 * i;
            int px = idx >> 4;
            px = px < plastp ? px : plastp;
            const int j = idx & 15;
            const u32x4 v = *(const u32x4*)(lds + kOffZ + (j >> 3) * kSlab + px * 128 + (((j & 7) ^ ((px >> 1) & 7)) * 16));
            *(u32x4*)(P.next + (size_t)halo_index(tp, px) * (kCM * 2) + j * 16) = v;
        }
    }
}

}  // namespace

extern "C" {

int dafne_bottleneck_tail_head_mid_hip(const void* d_in, const void* d_res, const void* d_wfrag, const float* d_bias3,
                                       const float* d_bias1, int n_images, int H, int W, void* d_out, void* d_next,
                                       void* stream) {
    if (!d_in || !d_res || !d_wfrag || !d_bias3 || !d_bias1 || !d_out || !d_next) return dafne::fail(DAFNE_E_INVALID, "bottleneck_tail_head_mid: null argument");
    if (n_images < 1 || H < 1 || W < 1 || (long long)H * W > (1 << 20)) return dafne::fail(DAFNE_E_INVALID, "bottleneck_tail_head_mid: bad size");
    MidDev D;
    D.in = (const char*)d_in; D.res = (const char*)d_res; D.wf = (const char*)d_wfrag; D.b3 = d_bias3; D.b1 = d_bias1;
    D.out = (char*)d_out; D.next = (char*)d_next;
    D.N = n_images; D.H = H; D.W = W;
    D.tiles_per_img = (H * W + kPx - 1) / kPx;
    const long long tiles = (long long)D.tiles_per_img * n_images;
    if (tiles > (1ll << 24)) return dafne::fail(DAFNE_E_UNSUPPORTED, "bottleneck_tail_head_mid: too many tiles");
    D.tiles = (int)tiles;
    DAFNE_MAX_LDS_ONCE(kSmemTotal, (const void*)conv_b2b_mid_kernel);
    int n_cu = 0;
    if (int rc = dafne::device_cus(&n_cu)) return rc;
    static const int cap = getenv("DAFNE_STREAM_GRID") ? atoi(getenv("DAFNE_STREAM_GRID")) : 0;
    const int lim = cap > 0 && cap < n_cu ? cap : n_cu;
    const int grid = D.tiles < lim ? D.tiles : lim;
    hipLaunchKernelGGL(conv_b2b_mid_kernel, dim3(grid), dim3(kNT), kSmemTotal, (hipStream_t)stream, D);
    return dafne::check_launch("conv_b2b_mid");
}

}  // extern "C"
