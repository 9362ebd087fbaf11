// 3x3 / stride 1 / pad 1 convolution with 64 input and 64 output channels (+ bias, optional ReLU): conv2 of the res2
// bottlenecks of ResNet-50/101 [detectron2 BottleneckBlock, recalled; built by build_dafne_resnet_fpn_backbone,
// backbone/fpn.py:58-91].  Three launches per image batch on the 256 x 256 maps (38.7 GFLOP, 134 MB each at batch 8).
//
// The generic implicit-GEMM tile re-fetches the pixel operand once per tap (9 x 67 MB through L2 -> LDS) and stages
// the 72 KB of weights for every tile: it sits at the L2 -> CU ingest limit (65 us, 593 TFLOP/s).  Here:
//   * persistent workgroups (one per CU, 8 waves); ALL weights live in registers for the whole kernel: a wave owns one
//     32-channel half (`wave & 1`) -> 9 taps x 4 k16 steps = 36 A fragments (144 VGPRs);
//   * an output tile is 8 rows x 32 columns; its (8+2) x (32+2) input pixels are DMA'd once into LDS (43 KB, double
//     buffered: the next tile's patch is issued piece by piece under the first half of a tile's MFMAs and awaited at the
//     start of the next; the finished tile's rows leave through a double-buffered staging tile under the second half of
//     the NEXT tile; `vmcnt(4)` = everything but those four stores; ONE barrier per tile);
//   * a wave computes two output rows (`wave >> 1`): it walks the four patch lines under them once, loads the B fragments
//     of a line (3 column shifts x 4 k16 steps) and feeds them to the row above (tap row kh = line) and the row below
//     (kh = line - 1): 48 fragment reads for 72 MFMAs, no barrier inside a tile;
//   * accumulation order per output = tap-major (kh, kw), k16 ascending -- the generic kernel's K order -- and the same
//     epilogue expression: results are bit-identical to dafne_conv2d_nhwc_bf16_hip (tests/test_gpu_conv.py).
#include <stdlib.h>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((address_space(1))) void gvoid;
typedef __attribute__((address_space(3))) void lvoid;

constexpr int kTH = 8, kTW = 32;
constexpr int kPC = kTW + 2;                    // patch columns
constexpr int kPPx = (kTH + 2) * kPC;           // 340 patch pixels
constexpr int kPieces = (kPPx + 7) / 8;         // 43 DMA pieces of 8 px x 128 B
constexpr int kPBuf = kPieces * 1024;           // 44 032 B
constexpr int kOffStage = 2 * kPBuf;            // 2 x bf16 [256 px][64 ch] output staging (a tile's rows are stored under the next tile)
constexpr int kStage = kTH * kTW * 128;
constexpr int kOffBias = kOffStage + 2 * kStage;
constexpr int kSmemTotal = kOffBias + 64 * 4;
constexpr int kNT = 512;
static_assert(kSmemTotal <= 160 * 1024, "LDS budget");

struct C64Dev {
    const char* in;      // bf16 [N, H+2, W+2, 64], zero halo
    const char* w;       // bf16 [64, 576]: k = (kh, kw, channel) (engine.pack_conv)
    const float* bias;   // [64]
    char* out;           // bf16 [N, H+2, W+2, 64] (interior written)
    int N, H, W, tiles_x, tiles_per_img, tiles, relu;
};

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    f32x2 v = {a, b};
    bf16x2 r = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(unsigned, r);
}

__global__ void __launch_bounds__(512, 2) conv3x3_c64_kernel(C64Dev P) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, half = lane >> 5;
    const int ct = wave & 1, r0 = (wave >> 1) * 2;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const int G = gridDim.x;
    const int my_tiles = (P.tiles - (int)blockIdx.x + G - 1) / G;
    const int Wp = P.W + 2, Hp = P.H + 2;

    // ---- weights -> registers (once): A fragment (tap, s) = rows ct*32 + (lane & 31), K columns tap*64 + 16 s + 8 (lane >> 5) .. +8
    bf16x8 a[9][4];
    {
        const char* wr = P.w + (size_t)(ct * 32 + frow) * (576 * 2) + half * 16;
#pragma unroll
        for (int t = 0; t < 9; t++)
#pragma unroll
            for (int s = 0; s < 4; s++)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a[t][s]) : "v"(wr + (t * 64 + s * 16) * 2) : "memory");
        if (tid < 64) ((float*)(lds + kOffBias))[tid] = P.bias[tid];
    }

    // ---- patch DMA: 43 pieces of 8 patch pixels; wave w moves pieces w, w + 8, ..  (6 per wave: the surplus ones repeat piece 42).
    // ONE piece per call: the six calls of a tile are spread over its first six fragment groups (issued back to back at the
    // top of the tile they held every wave ~2 k cycles in the vector-memory issue queue before its first MFMA).
    struct TileXY { int img, y0, x0; };
    auto tile_xy = [&](int t) {
        TileXY r;
        r.img = t / P.tiles_per_img;
        const int rem = t - r.img * P.tiles_per_img;
        const int ty = rem / P.tiles_x;
        r.y0 = ty * kTH;
        r.x0 = (rem - ty * P.tiles_x) * kTW;
        return r;
    };
    auto issue_piece = [&](const TileXY& T, int buf, int ii) {
        int piece = wave + 8 * ii;
        piece = piece < kPieces ? piece : kPieces - 1;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        int pp = piece * 8 + (ln >> 3);
        const unsigned q = (unsigned)(((ln & 7) ^ ((pp >> 1) & 7)) * 16);     // swizzle by the LDS pixel slot
        pp = pp < kPPx ? pp : kPPx - 1;
        const int line = (pp * 1928) >> 16;                       // pp / 34 for pp < 344
        const int col = pp - line * kPC;
        int yy = T.y0 + line, xx = T.x0 + col;                    // halo coordinates
        yy = yy < Hp ? yy : Hp - 1;                               // ragged tiles: stay inside the tensor (those outputs are not stored)
        xx = xx < Wp ? xx : Wp - 1;
        __builtin_amdgcn_global_load_lds((gvoid*)(P.in + ((size_t)(T.img * Hp + yy) * Wp + xx) * 128 + q),
                                         (lvoid*)(lds + buf * kPBuf + piece * 1024), 16, 0, 0);
    };
    // rows of a finished tile -> HBM, one of its 4 passes: 8 threads write one pixel's 128 B (pixels outside the image alias
    // the tile's last valid row / column: identical duplicate writes, so every lane issues exactly 4 stores per tile)
    auto store_pass = [&](const TileXY& T, int sbuf, int i) {
        const int ymax = P.H - 1 - T.y0, xmax = P.W - 1 - T.x0;
        int idx = tid + kNT * i;
        asm volatile("" : "+v"(idx));
        int py = idx >> 8, pxx = (idx >> 3) & 31;
        py = py < ymax ? py : ymax;
        pxx = pxx < xmax ? pxx : xmax;
        const int px = py * kTW + pxx, q = idx & 7;
        const u32x4 v = *(const u32x4*)(lds + kOffStage + sbuf * kStage + px * 128 + ((q ^ ((px >> 1) & 7)) * 16));
        *(u32x4*)(P.out + ((size_t)(T.img * Hp + T.y0 + py + 1) * Wp + T.x0 + pxx + 1) * 128 + q * 16) = v;
    };
    auto barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    if (my_tiles > 0) {
        const TileXY T0 = tile_xy((int)blockIdx.x);
#pragma unroll
        for (int ii = 0; ii < 6; ii++) issue_piece(T0, 0, ii);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int s = 0; s < 4; s++) asm volatile("" : "+v"(a[t][s]));

    // Vector-memory program order of a lane in tile k: 6 DMA pieces of tile k+1's patch (fragment groups 0..5), then the
    // 4 row stores of tile k-1 (groups 6..9; none in tile 0).  At the top of tile k+1 the patch must have landed and only
    // those stores are younger: vmcnt(4) (vmcnt(0) at the top of tile 1).
    for (int kk = 0; kk < my_tiles; kk++) {
        const int t = (int)blockIdx.x + kk * G;
        const int buf = kk & 1;
        if (kk == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (kk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        barrier();      // patch k visible; staging (k-1) & 1 complete; every wave is done with patch k-1 and staging k & 1
        const TileXY Tn = tile_xy(kk + 1 < my_tiles ? t + G : t);    // the last tile re-reads its own patch: fixed instruction count
        const TileXY Tp = tile_xy(kk > 0 ? t - G : t);

        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int k = 0; k < 16; k++) acc[j][k] = 0.f;
        // (hand-placed lgkmcnt waits with the next group's fragment reads issued ahead of a group's MFMAs were tried: no change)
        const char* pb = lds + buf * kPBuf;
#pragma unroll
        for (int L = 0; L < 4; L++) {
            int fr = frow;
            asm volatile("" : "+v"(fr));      // fragment addresses recomputed per line: hoisted out of the tile loop the 48 of them spill
#pragma unroll
            for (int kw = 0; kw < 3; kw++) {
                const int grp = L * 3 + kw;
                const int pp = (r0 + L) * kPC + kw + fr;              // patch pixel of this lane's B row
                bf16x8 bf[4];
#pragma unroll
                for (int s = 0; s < 4; s++)
                    bf[s] = *(const bf16x8*)(pb + pp * 128 + (((2 * s + half) ^ ((pp >> 1) & 7)) * 16));
                if (grp < 6) issue_piece(Tn, buf ^ 1, grp);
                else if (grp < 10 && kk > 0) store_pass(Tp, (kk - 1) & 1, grp - 6);
                if (L <= 2) {
#pragma unroll
                    for (int s = 0; s < 4; s++) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[L * 3 + kw][s], bf[s], acc[0], 0, 0, 0);
                }
                if (L >= 1) {
#pragma unroll
                    for (int s = 0; s < 4; s++) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(L - 1) * 3 + kw][s], bf[s], acc[1], 0, 0, 0);
                }
            }
        }
        // ---- epilogue: acc + bias (-> ReLU) -> bf16 -> staging tile kk & 1: [256 px][128 B], 16-byte piece ^ ((px >> 1) & 7)
        const float lo = P.relu ? 0.f : -__builtin_huge_valf();
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int px = (r0 + j) * kTW + frow;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const float* bp = (const float*)(lds + kOffBias) + ct * 32 + 8 * g + 4 * half;
                const float v0 = fmaxf(acc[j][4 * g] + bp[0], lo), v1 = fmaxf(acc[j][4 * g + 1] + bp[1], lo);
                const float v2 = fmaxf(acc[j][4 * g + 2] + bp[2], lo), v3 = fmaxf(acc[j][4 * g + 3] + bp[3], lo);
                u32x2 pk;
                pk.x = pack_bf16(v0, v1);
                pk.y = pack_bf16(v2, v3);
                const unsigned ad = lds_base + (unsigned)(kOffStage + buf * kStage + px * 128 + (((ct * 4 + g) ^ ((px >> 1) & 7)) * 16) + 8 * half);
                asm volatile("ds_write_b64 %0, %1" ::"v"(ad), "v"(pk) : "memory");
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- rows of the last tile
    barrier();
    if (my_tiles > 0) {
        const TileXY Tl = tile_xy((int)blockIdx.x + (my_tiles - 1) * G);
#pragma unroll
        for (int i = 0; i < 4; i++) store_pass(Tl, (my_tiles - 1) & 1, i);
    }
}

}  // namespace

extern "C" {

int dafne_conv3x3_c64_hip(const void* d_in, const void* d_weight, const float* d_bias, int n_images, int H, int W, int relu,
                          void* d_out, void* stream) {
    if (!d_in || !d_weight || !d_bias || !d_out) return dafne::fail(DAFNE_E_INVALID, "conv3x3_c64: null argument");
    if (n_images < 1 || H < 1 || W < 1 || (long long)H * W > (1 << 24)) return dafne::fail(DAFNE_E_INVALID, "conv3x3_c64: bad size");
    C64Dev D;
    D.in = (const char*)d_in; D.w = (const char*)d_weight; D.bias = d_bias; D.out = (char*)d_out;
    D.N = n_images; D.H = H; D.W = W; D.relu = relu ? 1 : 0;
    D.tiles_x = (W + kTW - 1) / kTW;
    D.tiles_per_img = D.tiles_x * ((H + kTH - 1) / kTH);
    const long long tiles = (long long)D.tiles_per_img * n_images;
    if (tiles > (1ll << 24)) return dafne::fail(DAFNE_E_UNSUPPORTED, "conv3x3_c64: too many tiles");
    D.tiles = (int)tiles;
    DAFNE_MAX_LDS_ONCE(kSmemTotal, (const void*)conv3x3_c64_kernel);
    int n_cu = 0;
    if (int rc = dafne::device_cus(&n_cu)) return rc;
    static const int cap = getenv("DAFNE_STREAM_GRID") ? atoi(getenv("DAFNE_STREAM_GRID")) : 0;
    const int lim = cap > 0 && cap < n_cu ? cap : n_cu;
    const int grid = D.tiles < lim ? D.tiles : lim;
    hipLaunchKernelGGL(conv3x3_c64_kernel, dim3(grid), dim3(kNT), kSmemTotal, (hipStream_t)stream, D);
    return dafne::check_launch("conv3x3_c64");
}

}  // extern "C"
