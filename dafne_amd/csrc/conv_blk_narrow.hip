// A WHOLE res2 bottleneck body (+ optionally the head of the next block) in one kernel (gfx950), ResNet-50/101
// [detectron2 BottleneckBlock, recalled; built by build_dafne_resnet_fpn_backbone, backbone/fpn.py:58-91]:
//
//     T = relu(conv2(U) + bias2)                  3x3, 64 -> 64, pad 1     (U = the block's conv1 output)
//     Y = relu(conv3(T) + bias3 + X)              1x1, 64 -> 256           (X = identity shortcut, or PROJ: conv_sc(X0) + bias_sc)
//     Z = relu(conv1'(Y) + bias1)                 1x1, 256 -> 64           (HEAD: the next block's first convolution)
//
// conv_b2b_narrow.hip already keeps Y on chip between conv3 and the next conv1; here T never leaves the CU either.  res2 is
// a pure streaming problem (1.3 flop per activation byte): per block and batch of 8 the 64-channel map T is 67 MB written
// by the 3x3 launch and 67 MB read back by the tail, next to 600 MB that have to move -- and one launch per block and
// stream instead of two.
//
// Structure: conv_b2b_narrow's (persistent workgroups, one per CU, 8 waves; 1x1 weights in registers; loads of tile k+1
// issued during tile k and awaited at its top with a COUNTED vmcnt that never waits for a store) with the T tile COMPUTED:
//   * tiles are 4 x 32 pixels; the (4+2) x (32+2) x 64-channel input patch (26 KB) is DMA'd into LDS (16-byte chunk XOR
//     by patch column: conflict-free ds_read_b128 at every tap offset), single-buffered: the patch of tile k+1 is
//     requested the moment phase A of tile k is done with it and lands under tile k's two 1x1 GEMMs;
//   * the 3x3's weights (72 KB) are RESIDENT IN LDS in fragment order for the whole kernel (144 VGPRs in registers would
//     not fit next to the 1x1 matrices): phase A = 36 k16 steps of (A fragment from LDS, B fragment from the patch, one
//     MFMA); a wave owns 32 output channels x one tile row;
//   * Y is produced in two 128-channel halves through a 32-KB buffer (the LDS budget: weights 72 + patch 26 + T 16 + Y 32 +
//     constants 4 = 150 KB): shortcut rows -> buffer, GEMM1 half, (acc + bias3) + X -> ReLU -> bf16 in place, rows -> HBM,
//     GEMM2 accumulates the half's K range; Z through a staging tile that aliases T;
//   * PROJ (block 0): the shortcut is a 1x1 convolution of the block input X0; its 128-pixel tile is prefetched into two
//     registers per lane, parked in the Y buffer at the top of the tile, and S = bf16(Wsc . X0 + bias_sc) is kept as
//     packed bf16 pairs in the accumulator layout (rounded exactly as the separate launch stores it);
//   * ragged tiles: loads are clamped into the tensor, rows of out-of-image pixels are STORED to a dump area (never
//     predicated: the vmcnt bookkeeping needs an exact instruction count).
// K orders (3x3: tap-major, k16 ascending; 1x1: ascending) and epilogue expressions are those of the separate kernels:
// bit-identical to dafne_conv2d_nhwc_bf16_hip(conv2, RELU) + dafne_bottleneck_[proj_]tail_head_narrow_hip.
#include <stdlib.h>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((address_space(1))) void gvoid;
typedef __attribute__((address_space(3))) void lvoid;

constexpr int kTH = 4, kTW = 32, kPx = kTH * kTW;
constexpr int kPC = kTW + 2, kPR = kTH + 2;
constexpr int kPPieces = (kPR * kPC + 7) / 8;        // 26 DMA pieces of 8 px x 128 B
constexpr int kPatch = kPPieces * 1024;              // 26 624 B
constexpr int kSlab = kPx * 128;                     // [128 px][64 ch]: 16 KB
constexpr int kCM = 64, kCB = 256;
constexpr int kStepsA = 36;                          // 9 taps x 4 k16 steps
constexpr int kW2Bytes = 2 * kStepsA * 1024;         // conv2 weights, fragment-major: 72 KB
constexpr int kOffW2 = 0;
constexpr int kOffPatch = kOffW2 + kW2Bytes;
constexpr int kOffT = kOffPatch + kPatch;            // T tile; later the Z staging tile
constexpr int kOffY = kOffT + kSlab;                 // one 128-channel half of Y: 2 slabs (PROJ: the X0 tile first)
constexpr int kOffBias = kOffY + 2 * kSlab;          // fp32 [64 conv2 | 256 conv3 | 64 conv1 | 256 projection]
constexpr int kSmemTotal = kOffBias + 4096;
static_assert((2 * kCM + 2 * kCB) * 4 <= 4096 && kSmemTotal <= 160 * 1024, "LDS budget");
constexpr int kNW = 8, kNT = 512;
constexpr int kDumpBytes = kPx * kCB * 2;            // one Y row per tile pixel: 64 KB
// d_wfrag sections (bytes)
constexpr int kWfA3 = kW2Bytes;                      // conv3: [2 halves][4 quarters][4 steps][64][8]
constexpr int kWfA1 = kWfA3 + 2 * 4 * 4 * 1024;      // conv1': [2 halves][16 steps][64][8]
constexpr int kWfSc = kWfA1 + 2 * 16 * 1024;         // projection: conv3's layout

struct BlkDev {
    const char* in;      // bf16 [N, H+2, W+2, 64]   U
    const char* res;     // bf16 [N, H+2, W+2, 256]  X;  PROJ: the block input X0 [N, H+2, W+2, 64]
    const char* wf;
    const float* b2;     // [64]
    const float* b3;     // [256]
    const float* b1;     // [64]   (HEAD)
    const float* bsc;    // [256]  (PROJ)
    char* out;           // bf16 [N, H+2, W+2, 256]  Y
    char* next;          // bf16 [N, H+2, W+2, 64]   Z  (HEAD)
    char* dump;          // >= kDumpBytes
    int N, H, W, tiles_x, tiles_per_img, tiles;
    unsigned max_pix;    // N * (H+2) * (W+2) - 1
};

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    f32x2 v = {a, b};
    bf16x2 r = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(unsigned, r);
}

// Vector-memory program order of a lane in tile k (after the top wait):
//   patch(k+1): 4 DMA pieces | [PROJ: 2 X0(k+1) row loads] | [!PROJ: 4 X(k+1) half-0 row loads] | 4 Y half-0 row stores |
//   [!PROJ: 4 X(k+1) half-1 row loads] | 4 Y half-1 row stores | [HEAD: 2 Z row stores]
// At the top of tile k+1 everything up to the last LOAD must have landed; younger than it are only stores:
//   PROJ: 8 + (HEAD ? 2 : 0);  !PROJ: 4 + (HEAD ? 2 : 0).   The wait never waits for a store of its own tile.
template <bool PROJ, bool HEAD>
__global__ void __launch_bounds__(512, 2) conv_blk_narrow_kernel(BlkDev P) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, half = lane >> 5;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const int Wp = P.W + 2;
    const int G = gridDim.x;
    const int my_tiles = (P.tiles - (int)blockIdx.x + G - 1) / G;

    struct TileXY { int img, row0, col0; };
    auto tile_xy = [&](int t) {
        TileXY r;
        r.img = t / P.tiles_per_img;
        const int rem = t - r.img * P.tiles_per_img;
        const int ty = rem / P.tiles_x;
        r.row0 = ty * kTH;
        r.col0 = (rem - ty * P.tiles_x) * kTW;
        return r;
    };
    // tile pixel px = r * 32 + c  <->  image pixel (row0 + r, col0 + c)
    auto pix_index = [&](const TileXY& T, int px) {           // haloed pixel index, clamped into the image (loads)
        int r = T.row0 + (px >> 5), c = T.col0 + (px & 31);
        r = r < P.H ? r : P.H - 1;
        c = c < P.W ? c : P.W - 1;
        return (unsigned)((T.img * (P.H + 2) + r + 1) * Wp + c + 1);
    };
    auto pix_valid = [&](const TileXY& T, int px) { return T.row0 + (px >> 5) < P.H && T.col0 + (px & 31) < P.W; };

    // ---- resident operands: conv2 weights -> LDS (72 pieces of 1 KB, 9 per wave), 1x1 weights -> registers, biases -> LDS
#pragma unroll
    for (int i = 0; i < 9; i++)
        __builtin_amdgcn_global_load_lds((gvoid*)(P.wf + (size_t)(wave * 9 + i) * 1024 + lane * 16), (lvoid*)(lds + kOffW2 + (wave * 9 + i) * 1024), 16, 0, 0);
    const int cq = wave & 3, ph = wave >> 2;                  // GEMM1: 32-channel quarter of a 128-channel half, 64-pixel half
    const int ct = wave & 1, pt = wave >> 1;                  // phase A / GEMM2: 32-channel half, tile row (32 pixels)
    bf16x8 a3[8], a1[HEAD ? 16 : 1], asc[PROJ ? 8 : 1];
    {
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const char* w3 = P.wf + kWfA3 + (size_t)((h * 4 + cq) * 4 + s) * 1024 + lane * 16;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a3[h * 4 + s]) : "v"(w3) : "memory");
                if (PROJ) {
                    const char* ws = P.wf + kWfSc + (size_t)((h * 4 + cq) * 4 + s) * 1024 + lane * 16;
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(asc[h * 4 + s]) : "v"(ws) : "memory");
                }
            }
        if (HEAD) {
#pragma unroll
            for (int s = 0; s < 16; s++) {
                const char* w1 = P.wf + kWfA1 + (size_t)(ct * 16 + s) * 1024 + lane * 16;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a1[s]) : "v"(w1) : "memory");
            }
        }
        float* lb = (float*)(lds + kOffBias);
        if (tid < kCM) lb[tid] = P.b2[tid];
        if (tid < kCB) lb[kCM + tid] = P.b3[tid];
        if (HEAD && tid < kCM) lb[kCM + kCB + tid] = P.b1[tid];
        if (PROJ && tid < kCB) lb[2 * kCM + kCB + tid] = P.bsc[tid];
    }

    // ---- per-tile loads
    auto issue_patch = [&](const TileXY& T) {
        // piece pc = 8 consecutive patch pixels (pp = p * 34 + q <-> haloed input pixel (row0 + p, col0 + q)); wave w moves
        // pieces w, w + 8, w + 16 and min(w + 24, 25): every wave issues the same number of DMAs
#pragma unroll
        for (int ii = 0; ii < 4; ii++) {
            int pc = wave + kNW * ii;
            pc = pc < kPPieces ? pc : kPPieces - 1;
            int ln = lane;
            asm volatile("" : "+v"(ln));                       // addresses recomputed per tile (held across the loop they spill)
            const int pp = pc * 8 + (ln >> 3);
            const int p = (pp * 1928) >> 16;                   // pp / 34 for pp < 344
            const int q = pp - p * kPC;
            unsigned g = (unsigned)((T.img * (P.H + 2) + T.row0 + p) * Wp + T.col0 + q);
            g = g < P.max_pix ? g : P.max_pix;                 // ragged tiles reach past the image (and the buffer)
            __builtin_amdgcn_global_load_lds((gvoid*)(P.in + (size_t)g * (kCM * 2) + (unsigned)(((ln & 7) ^ ((q >> 1) & 7)) * 16)),
                                             (lvoid*)(lds + kOffPatch + pc * 1024), 16, 0, 0);
        }
    };
    // shortcut rows of half h: pass i of 4, 16 threads read one pixel's 256 B
    u32x4 rr[PROJ ? 2 : 8];
#pragma unroll
    for (int i = 0; i < (PROJ ? 2 : 8); i++) rr[i] = u32x4{0u, 0u, 0u, 0u};
    auto issue_x_half = [&](const TileXY& T, int h) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int idx = tid + kNT * i;
            asm volatile("" : "+v"(idx));
            const char* src = P.res + (size_t)pix_index(T, idx >> 4) * (kCB * 2) + h * 256 + (idx & 15) * 16;
            // "+v": the destination stays the register that carries rr[] around the tile loop
            asm volatile("global_load_dwordx4 %0, %1, off nt" : "+v"(rr[h * 4 + i]) : "v"(src) : "memory");
        }
    };
    auto issue_x0 = [&](const TileXY& T) {                     // PROJ: the 128 x 64-channel X0 tile, 2 x 16 B per lane
#pragma unroll
        for (int i = 0; i < 2; i++) {
            int idx = tid + kNT * i;
            asm volatile("" : "+v"(idx));
            const char* src = P.res + (size_t)pix_index(T, idx >> 3) * (kCM * 2) + (idx & 7) * 16;
            asm volatile("global_load_dwordx4 %0, %1, off nt" : "+v"(rr[i]) : "v"(src) : "memory");
        }
    };

    unsigned bs[4];                          // B fragment of k16 step s inside a [128 px][128 B] slab, pixel fragment 0
#pragma unroll
    for (int s = 0; s < 4; s++) bs[s] = (unsigned)(frow * 128 + (((2 * s + half) ^ ((frow >> 1) & 7)) * 16));
    auto barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    const float* lbias = (const float*)(lds + kOffBias);

    if (my_tiles > 0) {
        const TileXY T0 = tile_xy((int)blockIdx.x);
        issue_patch(T0);
        if (PROJ) issue_x0(T0);
        else { issue_x_half(T0, 0); issue_x_half(T0, 1); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < 8; s++) asm volatile("" : "+v"(a3[s]));
    if (HEAD) {
#pragma unroll
        for (int s = 0; s < 16; s++) asm volatile("" : "+v"(a1[s]));
    }
    if (PROJ) {
#pragma unroll
        for (int s = 0; s < 8; s++) asm volatile("" : "+v"(asc[s]));
    }

    for (int kk = 0; kk < my_tiles; kk++) {
        const int t = (int)blockIdx.x + kk * G;
        const TileXY T = tile_xy(t);
        const TileXY Tn = tile_xy(kk + 1 < my_tiles ? t + G : t);      // the last tile re-requests itself: fixed instruction count
        // ---- 1. this tile's loads have landed (only the previous tile's last stores are younger)
        if (kk > 0) {
            if (PROJ) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 + (HEAD ? 2 : 0)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + (HEAD ? 2 : 0)) : "memory");
        }
        // ---- 2. parked operands -> the Y buffer (free since the previous tile's last barrier): PROJ the X0 tile, else the
        // shortcut rows of half 0 ([slab][px][64 ch], 16-byte chunk ^ ((px >> 1) & 7))
        if (PROJ) {
#pragma unroll
            for (int i = 0; i < 2; i++) {
                asm volatile("" : "+v"(rr[i]));
                const int idx = tid + kNT * i;
                const int px = idx >> 3, j = idx & 7;
                const unsigned ad = lds_base + (unsigned)(kOffY + px * 128 + ((j ^ ((px >> 1) & 7)) * 16));
                asm volatile("ds_write_b128 %0, %1" ::"v"(ad), "v"(rr[i]) : "memory");
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                asm volatile("" : "+v"(rr[i]));
                const int idx = tid + kNT * i;
                const int px = idx >> 4, j = idx & 15;
                const unsigned ad = lds_base + (unsigned)(kOffY + (j >> 3) * kSlab + px * 128 + (((j & 7) ^ ((px >> 1) & 7)) * 16));
                asm volatile("ds_write_b128 %0, %1" ::"v"(ad), "v"(rr[i]) : "memory");
            }
        }
        barrier();       // patch k (+ weights on the first tile) visible; every wave is done with the previous tile's LDS
        // ---- 3. phase A: T (32 channels ct x tile row pt) = conv2 over the patch, weights from LDS
        {
            f32x16 acc;
#pragma unroll
            for (int k = 0; k < 16; k++) acc[k] = 0.f;
            int fr = frow;
            asm volatile("" : "+v"(fr));
            const char* wa = lds + kOffW2 + ct * kStepsA * 1024 + lane * 16;
#pragma unroll
            for (int kh = 0; kh < 3; kh++)
#pragma unroll
                for (int kw = 0; kw < 3; kw++) {
                    const int q = kw + fr;                              // patch column of this lane's B row
                    const char* pb = lds + kOffPatch + ((pt + kh) * kPC + q) * 128;
                    const int sw = (q >> 1) & 7;
                    bf16x8 af[4], bf[4];
#pragma unroll
                    for (int s = 0; s < 4; s++) {
                        af[s] = *(const bf16x8*)(wa + ((kh * 3 + kw) * 4 + s) * 1024);
                        bf[s] = *(const bf16x8*)(pb + (((2 * s + half) ^ sw) * 16));
                    }
#pragma unroll
                    for (int s = 0; s < 4; s++) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], bf[s], acc, 0, 0, 0);
                }
            // (acc + bias2) -> ReLU -> bf16 -> T tile
            const int px = pt * 32 + frow;
            const unsigned tb = lds_base + (unsigned)(kOffT + px * 128 + 8 * half);
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const float* bp = lbias + ct * 32 + 8 * g + 4 * half;
                const float v0 = fmaxf(acc[4 * g] + bp[0], 0.f), v1 = fmaxf(acc[4 * g + 1] + bp[1], 0.f);
                const float v2 = fmaxf(acc[4 * g + 2] + bp[2], 0.f), v3 = fmaxf(acc[4 * g + 3] + bp[3], 0.f);
                u32x2 pk;
                pk.x = pack_bf16(v0, v1);
                pk.y = pack_bf16(v2, v3);
                const unsigned ad = tb + (unsigned)((((ct * 4 + g) ^ ((px >> 1) & 7))) * 16);
                asm volatile("ds_write_b64 %0, %1" ::"v"(ad), "v"(pk) : "memory");
            }
        }
        barrier();       // T complete; every wave is done with the patch
        // ---- 4. next tile's loads (the patch buffer is free; rr[0..3] / the X0 registers were parked in step 2)
        issue_patch(Tn);
        if (PROJ) issue_x0(Tn);
        else issue_x_half(Tn, 0);
        __builtin_amdgcn_sched_barrier(0);
        // ---- 4a. PROJ: S = bf16(Wsc . X0 + bias_sc) for both halves, packed bf16 pairs in the accumulator layout
        u32x2 sres[PROJ ? 2 : 1][2][4];
        if (PROJ) {
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int b = 0; b < 2; b++) {            // one pixel fragment at a time: 16 accumulator registers, not 32
                    f32x16 accs;
#pragma unroll
                    for (int k = 0; k < 16; k++) accs[k] = 0.f;
                    bf16x8 bfr[4];
#pragma unroll
                    for (int s = 0; s < 4; s++) bfr[s] = *(const bf16x8*)(lds + kOffY + (2 * ph + b) * 4096 + bs[s]);
#pragma unroll
                    for (int s = 0; s < 4; s++) accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(asc[h * 4 + s], bfr[s], accs, 0, 0, 0);
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const float* bp = lbias + 2 * kCM + kCB + h * 128 + cq * 32 + 8 * g + 4 * half;
                        sres[h][b][g].x = pack_bf16(accs[4 * g] + bp[0], accs[4 * g + 1] + bp[1]);
                        sres[h][b][g].y = pack_bf16(accs[4 * g + 2] + bp[2], accs[4 * g + 3] + bp[3]);
                    }
                }
        }
        f32x16 acc2;
#pragma unroll
        for (int k = 0; k < 16; k++) acc2[k] = 0.f;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            // ---- 5. GEMM1 half h: Y (32 channels cq x 64 pixels ph) = W3 . T
            f32x16 acc1[2];
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int k = 0; k < 16; k++) acc1[b][k] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; s++) {
                bf16x8 bfr[2];
#pragma unroll
                for (int b = 0; b < 2; b++) bfr[b] = *(const bf16x8*)(lds + kOffT + (2 * ph + b) * 4096 + bs[s]);
#pragma unroll
                for (int b = 0; b < 2; b++) acc1[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[h * 4 + s], bfr[b], acc1[b], 0, 0, 0);
            }
            if (PROJ && h == 0) barrier();       // every wave has read the X0 tile: the Y buffer may be overwritten
            if (!PROJ && h == 1) barrier();      // (the barrier behind step 7 of half 0: shortcut rows of half 1 are in the buffer)
            // ---- 6. (acc + bias3) + X -> ReLU -> bf16, in place in the Y buffer (each wave touches only its own 32 channels x 64 px)
            {
                typedef __attribute__((ext_vector_type(4))) float f32x4;
                typedef __attribute__((ext_vector_type(2))) float f32x2;
                const unsigned ebase = lds_base + (unsigned)(kOffY + (cq >> 1) * kSlab + (2 * ph) * 4096 + frow * 128 + 8 * half);
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const unsigned ead = ebase + (unsigned)(((((cq & 1) * 4 + g) ^ ((frow >> 1) & 7))) * 16);
                    const f32x4 bv = *(const f32x4*)(lbias + kCM + h * 128 + cq * 32 + 8 * g + 4 * half);
                    u32x2 rc[2];
                    if (PROJ) {
                        rc[0] = sres[h][0][g];
                        rc[1] = sres[h][1][g];
                    } else {
                        asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:4096\n\ts_waitcnt lgkmcnt(0)"
                                     : "=&v"(rc[0]), "=&v"(rc[1]) : "v"(ead) : "memory");
                    }
                    const f32x2 blo = {bv[0], bv[1]}, bhi = {bv[2], bv[3]};
#pragma unroll
                    for (int b = 0; b < 2; b++) {
                        const u32x2 r = rc[b];
                        const f32x2 rlo = {__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u)};
                        const f32x2 rhi = {__uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
                        const f32x2 alo = {acc1[b][4 * g], acc1[b][4 * g + 1]}, ahi = {acc1[b][4 * g + 2], acc1[b][4 * g + 3]};
                        const f32x2 vlo = alo + blo + rlo, vhi = ahi + bhi + rhi;          // (acc + bias) + residual
                        rc[b].x = pack_bf16(fmaxf(vlo[0], 0.f), fmaxf(vlo[1], 0.f));
                        rc[b].y = pack_bf16(fmaxf(vhi[0], 0.f), fmaxf(vhi[1], 0.f));
                    }
                    asm volatile("ds_write_b64 %2, %0\n\tds_write_b64 %2, %1 offset:4096" ::"v"(rc[0]), "v"(rc[1]), "v"(ead) : "memory");
                }
            }
            barrier();       // the Y half is complete
            // ---- 7. Y half rows -> HBM: pass i of 4, 16 threads write one pixel's 256 B (exactly 4 stores per lane)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                int idx = tid + kNT * i;
                asm volatile("" : "+v"(idx));
                const int px = idx >> 4, j = idx & 15;
                const u32x4 v = *(const u32x4*)(lds + kOffY + (j >> 3) * kSlab + px * 128 + (((j & 7) ^ ((px >> 1) & 7)) * 16));
                char* a = P.out + (size_t)pix_index(T, px) * (kCB * 2);
                char* d = P.dump + (size_t)px * (kCB * 2);
                a = pix_valid(T, px) ? a : d;
                __builtin_nontemporal_store(v, (u32x4*)(a + h * 256 + j * 16));
            }
            // ---- 8. GEMM2 over this half's K range: Z (channel half ct x tile row pt) += W1[:, half h] . Y half
            if (HEAD) {
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    bf16x8 bfr[4];
#pragma unroll
                    for (int s = 0; s < 4; s++) bfr[s] = *(const bf16x8*)(lds + kOffY + q * kSlab + pt * 4096 + bs[s]);
#pragma unroll
                    for (int s = 0; s < 4; s++) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[(2 * h + q) * 4 + s], bfr[s], acc2, 0, 0, 0);
                }
            }
            if (h == 0) {
                barrier();   // every wave is done with the Y half (row stores and GEMM2 have read it)
                if (!PROJ) {
                    // shortcut rows of half 1 -> the Y buffer, then the next tile's half-1 rows into the freed registers
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        asm volatile("" : "+v"(rr[4 + i]));
                        const int idx = tid + kNT * i;
                        const int px = idx >> 4, j = idx & 15;
                        const unsigned ad = lds_base + (unsigned)(kOffY + (j >> 3) * kSlab + px * 128 + (((j & 7) ^ ((px >> 1) & 7)) * 16));
                        asm volatile("ds_write_b128 %0, %1" ::"v"(ad), "v"(rr[4 + i]) : "memory");
                    }
                    issue_x_half(Tn, 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // ---- 9. (acc + bias1) -> ReLU -> bf16 -> Z staging tile (the T tile's LDS: GEMM1 is done) -> Z rows
        if (HEAD) {
            const unsigned zb = lds_base + (unsigned)(kOffT + (pt * 32 + frow) * 128 + 8 * half);
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const float* bp = lbias + kCM + kCB + ct * 32 + 8 * g + 4 * half;
                const float v0 = fmaxf(acc2[4 * g] + bp[0], 0.f), v1 = fmaxf(acc2[4 * g + 1] + bp[1], 0.f);
                const float v2 = fmaxf(acc2[4 * g + 2] + bp[2], 0.f), v3 = fmaxf(acc2[4 * g + 3] + bp[3], 0.f);
                u32x2 pk;
                pk.x = pack_bf16(v0, v1);
                pk.y = pack_bf16(v2, v3);
                const unsigned ad = zb + (unsigned)((((ct * 4 + g) ^ ((frow >> 1) & 7))) * 16);
                asm volatile("ds_write_b64 %0, %1" ::"v"(ad), "v"(pk) : "memory");
            }
            barrier();
#pragma unroll
            for (int i = 0; i < 2; i++) {           // 8 threads write one pixel's 128 B (exactly 2 stores per lane)
                int idx = tid + kNT * i;
                asm volatile("" : "+v"(idx));
                const int px = idx >> 3, q = idx & 7;
                const u32x4 v = *(const u32x4*)(lds + kOffT + px * 128 + ((q ^ ((px >> 1) & 7)) * 16));
                char* a = P.next + (size_t)pix_index(T, px) * (kCM * 2);
                char* d = P.dump + (size_t)px * (kCM * 2);
                a = pix_valid(T, px) ? a : d;
                *(u32x4*)(a + q * 16) = v;
            }
        }
        // without the Z epilogue's barrier nothing separates a slow wave's LDS reads for the last row stores from a fast wave
        // parking the next tile's rows in the same buffer
        if (!HEAD) barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    // the last tile re-requested its own patch: nothing may still be on its way into this workgroup's LDS when it is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

extern "C" {

size_t dafne_bottleneck_block_narrow_scratch_bytes(void) { return (size_t)kDumpBytes; }

int dafne_bottleneck_block_narrow_hip(const void* d_in, const void* d_res, const void* d_wfrag, const float* d_bias2,
                                      const float* d_bias3, const float* d_bias_sc, const float* d_bias1, int n_images, int H, int W,
                                      void* d_out, void* d_next, void* d_scratch, size_t scratch_bytes, void* stream) {
    const bool proj = d_bias_sc != nullptr, head = d_next != nullptr;
    if (!d_in || !d_res || !d_wfrag || !d_bias2 || !d_bias3 || !d_out || !d_scratch || (head && !d_bias1))
        return dafne::fail(DAFNE_E_INVALID, "bottleneck_block_narrow: null argument");
    if (n_images < 1 || H < 1 || W < 1 || (long long)H * W > (1 << 20)) return dafne::fail(DAFNE_E_INVALID, "bottleneck_block_narrow: bad size");
    if (scratch_bytes < (size_t)kDumpBytes) return dafne::fail(DAFNE_E_WORKSPACE, "bottleneck_block_narrow: scratch %zu < %d", scratch_bytes, kDumpBytes);
    BlkDev D;
    D.in = (const char*)d_in; D.res = (const char*)d_res; D.wf = (const char*)d_wfrag;
    D.b2 = d_bias2; D.b3 = d_bias3; D.b1 = d_bias1; D.bsc = d_bias_sc;
    D.out = (char*)d_out; D.next = (char*)d_next; D.dump = (char*)d_scratch;
    D.N = n_images; D.H = H; D.W = W;
    D.tiles_x = (W + kTW - 1) / kTW;
    D.tiles_per_img = D.tiles_x * ((H + kTH - 1) / kTH);
    const long long tiles = (long long)D.tiles_per_img * n_images;
    const long long pix = (long long)n_images * (H + 2) * (W + 2);
    if (tiles > (1ll << 24) || pix * (kCB * 2) > 0xffffffffll) return dafne::fail(DAFNE_E_UNSUPPORTED, "bottleneck_block_narrow: too large");
    D.tiles = (int)tiles;
    D.max_pix = (unsigned)(pix - 1);
    DAFNE_MAX_LDS_ONCE(kSmemTotal, (const void*)conv_blk_narrow_kernel<false, false>, (const void*)conv_blk_narrow_kernel<false, true>,
                       (const void*)conv_blk_narrow_kernel<true, false>, (const void*)conv_blk_narrow_kernel<true, true>);
    int n_cu = 0;
    if (int rc = dafne::device_cus(&n_cu)) return rc;
    const int grid = D.tiles < n_cu ? D.tiles : n_cu;
    hipStream_t st = (hipStream_t)stream;
    if (proj && head) hipLaunchKernelGGL((conv_blk_narrow_kernel<true, true>), dim3(grid), dim3(kNT), kSmemTotal, st, D);
    else if (proj) hipLaunchKernelGGL((conv_blk_narrow_kernel<true, false>), dim3(grid), dim3(kNT), kSmemTotal, st, D);
    else if (head) hipLaunchKernelGGL((conv_blk_narrow_kernel<false, true>), dim3(grid), dim3(kNT), kSmemTotal, st, D);
    else hipLaunchKernelGGL((conv_blk_narrow_kernel<false, false>), dim3(grid), dim3(kNT), kSmemTotal, st, D);
    return dafne::check_launch("conv_blk_narrow");
}

}  // extern "C"
