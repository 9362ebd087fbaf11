// ResNet BasicStem in one kernel (gfx950): conv 7x7 / stride 2 / pad 3 (3 -> 64, FrozenBN folded) + ReLU +
// max-pool 3x3 / stride 2 / pad 1 [detectron2 BasicStem, recalled; built by build_dafne_resnet_fpn_backbone,
// dafne/modeling/backbone/fpn.py:58-91].
//
// Unfused, the 64-channel half-resolution map (268 MB at batch 8, 1024^2) is written by the convolution and read
// back by the pool: 2/3 of the HBM traffic of the two layers.  Here a workgroup owns a tile of 4 x 32 POOLED
// pixels, computes the 9 x 65 convolution outputs under it (8 % halo recompute) into LDS and pools from there:
// HBM sees the 4-channel image (8 B/px) and the pooled map only.
//
//   * input patch 23 x 136 px x 4 ch bf16 (24.4 KB) of the layout dafne_preprocess_image_hip writes, double
//     buffered: the next tile's patch is fetched into registers under the MFMA phase (persistent workgroups);
//   * implicit GEMM M = 585 conv px (19 fragments of 32), N = 64, K = 7 kh x 8 kw x 4 ch = 224 (kw = 7 and the
//     4th channel carry zero weights): 14 MFMA 32x32x16 steps; a step's B fragment is 16 contiguous bytes of a
//     patch row (2 taps x 4 ch), so im2col is just an LDS address;
//   * 8 waves = 2 channel halves x 4 fragment lanes; a wave keeps its 14 weight fragments in registers for the
//     whole kernel (weights never touch LDS);
//   * CONV1 (round 4): res2.0's first convolution (1x1, 64 -> 64, + ReLU) on the pooled tile while it is still in LDS --
//     the pooled map is written once and not read back by a separate launch (67 MB at batch 8); K order and epilogue
//     expression of the generic kernel: bit-identical to stem_pool -> dafne_conv2d_nhwc_bf16_hip;
//   * K is walked in the same order as the generic kernel's stem path (the skipped kh = 7 steps multiply zero
//     weights), and max of post-ReLU bf16 values is exact: results are bit-identical to conv -> pool.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

constexpr int kTH = 4, kTW = 32;                 // pooled tile
constexpr int kCR = 2 * kTH + 1, kCC = 2 * kTW + 1;   // conv outputs under it: 9 x 65
constexpr int kCPx = kCR * kCC;                  // 585
constexpr int kFrags = (kCPx + 31) / 32;         // 19
constexpr int kPR = 2 * kCR + 5, kPC = 2 * kCC + 6;   // input patch 23 x 136 px (8 B each)
constexpr int kPatchChunks = kPR * kPC / 2;      // 16-byte chunks: 1564
constexpr int kPatchBytes = kPatchChunks * 16;   // 25 024
constexpr int kStgRow = 64 * 2 + 16;             // staged conv pixel: 64 ch bf16 + pad (bank spread, 16-B aligned)
constexpr int kOffStg = 2 * kPatchBytes;
constexpr int kSmem = kOffStg + kCPx * kStgRow;  // 134 288 B
constexpr int kOffPT = kSmem;                    // CONV1: the pooled tile [128 px][128 B], chunk ^ ((px >> 1) & 7)
constexpr int kSmem1 = kOffPT + kTH * kTW * 128; // 150 672 B
static_assert(kSmem1 <= 160 * 1024, "LDS budget");
constexpr int kFetch = (kPatchChunks + 511) / 512;   // chunks per thread: 4

struct StemDev {
    const char* in;      // bf16 [N, H+6, W+6, 4]
    const char* w;       // bf16 [64, 256]: k = (kh 0..7, kw 0..7, c 0..3)
    const float* bias;   // [64]
    char* out;           // bf16 [N, H/4+2, W/4+2, 64]
    const char* w1;      // CONV1: bf16 [64, 64] (cout, cin)
    const float* b1;     // CONV1: [64]
    char* out1;          // CONV1: bf16 [N, H/4+2, W/4+2, 64]
    int N, H, W;         // padded image size (multiples of 4; conv H/2 x W/2, pool H/4 x W/4)
    int tiles_x, tiles_y, tiles;
};

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    f32x2 v = {a, b};
    bf16x2 r = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(unsigned, r);
}

__device__ __forceinline__ unsigned max_u16x2(unsigned a, unsigned b) {
    unsigned r;
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <bool CONV1>
__global__ void __launch_bounds__(512) stem_pool_kernel(StemDev P) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cf = wave & 1, fl = wave >> 1;         // channel half (32 cout), fragment lane: fragments fl, fl+4, ..
    const int frow = lane & 31, half = lane >> 5;
    const int Hi = P.H + 6, Wi = P.W + 6;
    const int Hc = P.H / 2, Wc = P.W / 2, Hq = P.H / 4, Wq = P.W / 4;
    (void)Hc; (void)Wc;

    // weight fragments of this wave: cout row cf*32 + frow, step s = (kh, kw0 = 4 (s & 1)), K half `half`
    bf16x8 af[14];
#pragma unroll
    for (int s = 0; s < 14; s++)
        af[s] = *(const bf16x8*)(P.w + (size_t)(cf * 32 + frow) * 512 + s * 32 + half * 16);

    // conv pixel of this lane in each of the wave's fragments -> byte offset of its (kh = 0, kw0 = 0) operand
    unsigned boff[5];
    bool pvalid_r0[5], pvalid_c0[5];
#pragma unroll
    for (int j = 0; j < 5; j++) {
        int idx = (fl + 4 * j) * 32 + frow;
        idx = idx < kCPx ? idx : kCPx - 1;
        const int r = idx / kCC, c = idx - r * kCC;
        boff[j] = (unsigned)((2 * r * kPC + 2 * c + 2 * half) * 8);
        pvalid_r0[j] = r != 0;
        pvalid_c0[j] = c != 0;
    }

    auto fetch = [&](int tile, u32x4* regs) {
        const int img = tile / (P.tiles_x * P.tiles_y);
        const int tt = tile - img * (P.tiles_x * P.tiles_y);
        const int ty = tt / P.tiles_x, tx = tt - ty * P.tiles_x;
        const int row0 = 4 * ty * kTH - 2, col0 = 4 * tx * kTW - 2;
#pragma unroll
        for (int k = 0; k < kFetch; k++) {
            int i = tid + 512 * k;
            i = i < kPatchChunks ? i : kPatchChunks - 1;
            const int pr = i / (kPC / 2), pc2 = i - pr * (kPC / 2);
            int row = row0 + pr, col = col0 + 2 * pc2;
            row = row < 0 ? 0 : (row > Hi - 1 ? Hi - 1 : row);
            col = col < 0 ? 0 : (col > Wi - 2 ? Wi - 2 : col);
            regs[k] = *(const u32x4*)(P.in + (((size_t)img * Hi + row) * Wi + col) * 8);
        }
    };
    auto stash = [&](int buf, const u32x4* regs) {
#pragma unroll
        for (int k = 0; k < kFetch; k++) {
            const int i = tid + 512 * k;
            if (i < kPatchChunks) *(u32x4*)(lds + buf * kPatchBytes + i * 16) = regs[k];
        }
    };

    const int G = gridDim.x;
    int tile = blockIdx.x;
    u32x4 pre[kFetch];
    if (tile < P.tiles) {
        fetch(tile, pre);
        stash(0, pre);
    }
    __syncthreads();

    const float4 bia[4] = {*(const float4*)(P.bias + cf * 32 + 0 + 4 * half), *(const float4*)(P.bias + cf * 32 + 8 + 4 * half),
                           *(const float4*)(P.bias + cf * 32 + 16 + 4 * half), *(const float4*)(P.bias + cf * 32 + 24 + 4 * half)};

    // CONV1: wave = (cout half cf) x (pooled row fl of the tile: pixels fl * 32 + frow); its four k16 weight fragments
    bf16x8 a1[4];
    float4 bia1[4];
    if constexpr (CONV1) {
#pragma unroll
        for (int kc = 0; kc < 4; kc++) a1[kc] = *(const bf16x8*)(P.w1 + ((size_t)(cf * 32 + frow) * 64 + 16 * kc + 8 * half) * 2);
#pragma unroll
        for (int g = 0; g < 4; g++) bia1[g] = *(const float4*)(P.b1 + cf * 32 + 8 * g + 4 * half);
    }

    for (int it = 0; tile < P.tiles; tile += G, it++) {
        const int buf = it & 1;
        const int nxt = tile + G;
        const bool more = nxt < P.tiles;
        if (more) fetch(nxt, pre);                      // in flight under the MFMA phase

        const int img = tile / (P.tiles_x * P.tiles_y);
        const int tt = tile - img * (P.tiles_x * P.tiles_y);
        const int ty = tt / P.tiles_x, tx = tt - ty * P.tiles_x;

        const char* pb = lds + buf * kPatchBytes;
        f32x16 acc[5];
#pragma unroll
        for (int j = 0; j < 5; j++)
#pragma unroll
            for (int k = 0; k < 16; k++) acc[j][k] = 0.f;
        const int nfr = fl + 16 < kFrags ? 5 : 4;
#pragma unroll
        for (int s = 0; s < 14; s++) {
            const int koff = ((s >> 1) * kPC + (s & 1) * 4) * 8;
#pragma unroll
            for (int j = 0; j < 5; j++) {
                if (j < nfr) {
                    const u32x2 lo = *(const u32x2*)(pb + boff[j] + koff);
                    const u32x2 hi = *(const u32x2*)(pb + boff[j] + koff + 8);
                    const u32x4 v = {lo.x, lo.y, hi.x, hi.y};
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], __builtin_bit_cast(bf16x8, v), acc[j], 0, 0, 0);
                }
            }
        }

        // ---- bias, ReLU, zero the conv row / column -1 (the pool's padding), bf16 -> staging
        const bool top = ty == 0, left = tx == 0;
#pragma unroll
        for (int j = 0; j < 5; j++) {
            if (j < nfr) {
                const int idx = (fl + 4 * j) * 32 + frow;
                if (idx < kCPx) {
                    const bool ok = (pvalid_r0[j] || !top) && (pvalid_c0[j] || !left);
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        float v0 = fmaxf(acc[j][4 * g] + bia[g].x, 0.f), v1 = fmaxf(acc[j][4 * g + 1] + bia[g].y, 0.f);
                        float v2 = fmaxf(acc[j][4 * g + 2] + bia[g].z, 0.f), v3 = fmaxf(acc[j][4 * g + 3] + bia[g].w, 0.f);
                        u32x2 pk;
                        pk.x = ok ? pack_bf16(v0, v1) : 0u;
                        pk.y = ok ? pack_bf16(v2, v3) : 0u;
                        *(u32x2*)(lds + kOffStg + idx * kStgRow + (cf * 32 + 8 * g + 4 * half) * 2) = pk;
                    }
                }
            }
        }
        __syncthreads();

        // ---- pool: item = (pooled pixel, 8-channel chunk); 128 px x 8 chunks over 512 threads
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int item = tid + 512 * k;
            const int q = item >> 3, ch = item & 7;
            const int qy = q >> 5, qx = q & 31;
            u32x4 m = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int dy = 0; dy < 3; dy++)
#pragma unroll
                for (int dx = 0; dx < 3; dx++) {
                    const u32x4 v = *(const u32x4*)(lds + kOffStg + ((2 * qy + dy) * kCC + 2 * qx + dx) * kStgRow + ch * 16);
                    m.x = max_u16x2(m.x, v.x);
                    m.y = max_u16x2(m.y, v.y);
                    m.z = max_u16x2(m.z, v.z);
                    m.w = max_u16x2(m.w, v.w);
                }
            const int py = ty * kTH + qy, px = tx * kTW + qx;
            if (py < Hq && px < Wq)
                *(u32x4*)(P.out + ((((size_t)img * (Hq + 2) + py + 1) * (Wq + 2) + px + 1) * 64 + ch * 8) * 2) = m;
            if constexpr (CONV1) *(u32x4*)(lds + kOffPT + q * 128 + ((ch ^ ((q >> 1) & 7)) * 16)) = m;
        }
        if (more) stash(buf ^ 1, pre);
        __syncthreads();     // next patch complete; everyone is done with the staging tile (CONV1: the pooled tile is complete)
        if constexpr (CONV1) {
            // (the next tile's pool phase rewrites the pooled tile only behind its own first barrier: every wave is past this)
            f32x16 c1;
#pragma unroll
            for (int k = 0; k < 16; k++) c1[k] = 0.f;
            const int q = fl * 32 + frow;
#pragma unroll
            for (int kc = 0; kc < 4; kc++) {
                const bf16x8 bq = *(const bf16x8*)(lds + kOffPT + q * 128 + (((2 * kc + half) ^ ((q >> 1) & 7)) * 16));
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[kc], bq, c1, 0, 0, 0);
            }
            u32x2 pk[4];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const float v0 = fmaxf(c1[4 * g] + bia1[g].x, 0.f), v1 = fmaxf(c1[4 * g + 1] + bia1[g].y, 0.f);
                const float v2 = fmaxf(c1[4 * g + 2] + bia1[g].z, 0.f), v3 = fmaxf(c1[4 * g + 3] + bia1[g].w, 0.f);
                pk[g].x = pack_bf16(v0, v1);
                pk[g].y = pack_bf16(v2, v3);
            }
            const int py = ty * kTH + fl, px = tx * kTW + frow;
            const bool ok = py < Hq && px < Wq;
            char* o1 = P.out1 + ((((size_t)img * (Hq + 2) + py + 1) * (Wq + 2) + px + 1) * 64 + cf * 32 + 8 * half) * 2;
#pragma unroll
            for (int gp = 0; gp < 2; gp++) {
                // lanes 0..31 get (group 2gp: own channels 0..3 | the upper half-wave's 4..7), lanes 32..63 the same of group 2gp+1
                const u32x2 a = pk[2 * gp], c2 = pk[2 * gp + 1];
                const auto r0 = __builtin_amdgcn_permlane32_swap(a.x, c2.x, false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(a.y, c2.y, false, false);
                const u32x4 v = {r0[0], r1[0], r0[1], r1[1]};
                if (ok) *(u32x4*)(o1 + gp * 32) = v;
            }
        }
    }
}

}  // namespace

extern "C" {

static int stem_launch(const void* d_in, const void* d_weight, const float* d_bias, const void* d_w1, const float* d_b1, int n_images,
                       int H, int W, void* d_out, void* d_out1, void* stream, bool conv1) {
    if (!d_in || !d_weight || !d_bias || !d_out || n_images < 1) return dafne::fail(DAFNE_E_INVALID, "stem_pool: null argument");
    if (conv1 && (!d_w1 || !d_b1 || !d_out1)) return dafne::fail(DAFNE_E_INVALID, "stem_pool_conv1: null argument");
    if (H < 4 || W < 4 || (H % 4) || (W % 4)) return dafne::fail(DAFNE_E_INVALID, "stem_pool: image size %dx%d must be a multiple of 4", H, W);
    StemDev D;
    D.in = (const char*)d_in; D.w = (const char*)d_weight; D.bias = d_bias; D.out = (char*)d_out;
    D.w1 = (const char*)d_w1; D.b1 = d_b1; D.out1 = (char*)d_out1;
    D.N = n_images; D.H = H; D.W = W;
    D.tiles_x = (W / 4 + kTW - 1) / kTW;
    D.tiles_y = (H / 4 + kTH - 1) / kTH;
    const long long tiles = (long long)D.tiles_x * D.tiles_y * n_images;
    if (tiles > (1ll << 30)) return dafne::fail(DAFNE_E_UNSUPPORTED, "stem_pool: too many tiles");
    D.tiles = (int)tiles;
    DAFNE_MAX_LDS_ONCE(kSmem1, (const void*)stem_pool_kernel<false>, (const void*)stem_pool_kernel<true>);
    int cus = 0;
    if (int rc = dafne::device_cus(&cus)) return rc;
    const int grid = D.tiles < cus ? D.tiles : cus;
    if (conv1) hipLaunchKernelGGL(stem_pool_kernel<true>, dim3(grid), dim3(512), kSmem1, (hipStream_t)stream, D);
    else hipLaunchKernelGGL(stem_pool_kernel<false>, dim3(grid), dim3(512), kSmem, (hipStream_t)stream, D);
    return dafne::check_launch(conv1 ? "stem_pool_conv1" : "stem_pool");
}

int dafne_stem_pool_hip(const void* d_in, const void* d_weight, const float* d_bias, int n_images, int H, int W,
                        void* d_out, void* stream) {
    return stem_launch(d_in, d_weight, d_bias, nullptr, nullptr, n_images, H, W, d_out, nullptr, stream, false);
}

int dafne_stem_pool_conv1_hip(const void* d_in, const void* d_weight, const float* d_bias, const void* d_w1, const float* d_b1,
                              int n_images, int H, int W, void* d_out, void* d_out1, void* stream) {
    return stem_launch(d_in, d_weight, d_bias, d_w1, d_b1, n_images, H, W, d_out, d_out1, stream, true);
}

}  // extern "C"
