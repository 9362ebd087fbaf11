// A WHOLE res3 bottleneck body (+ the head of the next block) in one kernel (gfx950), ResNet-50/101
// [detectron2 BottleneckBlock, recalled; built by build_dafne_resnet_fpn_backbone, backbone/fpn.py:58-91]:
//
//     T = relu(conv2(U) + bias2)                  3x3, 128 -> 128, pad 1   (U = the block's conv1 output)
//     Y = relu(conv3(T) + bias3 + X)              1x1, 128 -> 512          (X = the block's shortcut: identity or projection output)
//     Z = relu(conv1'(Y) + bias1)                 1x1, 512 -> 128          (HEAD: the next block's first convolution)
//
// Before: conv_igemm<2,2,2,2> for the 3x3 (50 us at batch 8, T written and read back: 67 MB) + conv_b2b_mid for the pair
// (92 us: its 64-pixel tiles stream the 256 KB of 1x1 weights once per 64 pixels -- 4 KB of weights per pixel against 2.6 KB
// of activations; it is bound by the L2 -> CU ingest of its own weight stream).  Here one persistent workgroup (8 waves, one
// per CU) owns a 4 x 32 pixel tile and ALL channels, so every weight byte is fetched once per 128 pixels, and T never
// leaves the CU:
//   * phase A (3x3): the (4+2) x (32+2) x 128-channel patch (two 26-KB slabs, chunk XOR by patch column) is resident; the
//     288 KB of conv2 weights stream L2 -> registers (fragment-major, ring of 8 k16 steps, counted vmcnt); a wave owns 32
//     output channels x two tile rows; K order = conv_igemm's (64-channel slab, kh, kw, k16): T is bit-identical;
//   * phase B, per 256-channel half of Y: GEMM1 (wave = 32 output channels x 128 px, its 8 weight fragments were requested
//     a phase earlier), shortcut rows parked in the Y buffer from registers (prefetched a half ahead, conv_b2b_narrow's
//     scheme), (acc + bias3) + X -> ReLU -> bf16 in place, rows -> HBM, GEMM2 over the half's K range with conv1' fragments
//     through a ring of 8 (wave = 32 output channels x 64 px);
//   * the patch of tile k+1 is requested the moment phase A of tile k is done with it.
// Every s_waitcnt vmcnt is a compile-time count (the sequence is written out in front of the kernel); none waits for an HBM
// load that was issued less than a GEMM earlier, and the next tile's loads are never drained behind this tile's stores
// except at the last step of each GEMM2 (vmcnt(0): its 8 row stores are ~16 k16 steps old by then).
// Ragged tiles: loads clamped into the tensor, rows of out-of-image pixels STORED to a dump area (exact instruction counts).
// Bit-identical to dafne_conv2d_nhwc_bf16_hip(conv2, RELU) + dafne_bottleneck_tail_head_mid_hip (or + (conv3, RELU|RESIDUAL)).
#include <type_traits>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((address_space(1))) void gvoid;
typedef __attribute__((address_space(3))) void lvoid;

constexpr int kTH = 4, kTW = 32, kPx = kTH * kTW;
constexpr int kPC = kTW + 2, kPR = kTH + 2;
constexpr int kPPieces = (kPR * kPC + 7) / 8;        // 26 DMA pieces of 8 px x 128 B per 64-channel slab
constexpr int kPSlab = kPPieces * 1024;              // 26 624 B
constexpr int kSlab = kPx * 128;                     // [128 px][64 ch]: 16 KB
constexpr int kCM = 128, kCB = 512;
constexpr int kStepsA = 9 * (kCM / 16);              // 72 k16 steps of the 3x3
constexpr int kOffPatch = 0;                         // 2 slabs
constexpr int kOffT = kOffPatch + 2 * kPSlab;        // T tile (2 slabs); later the Z staging tile
constexpr int kOffY = kOffT + 2 * kSlab;             // one 256-channel half of Y: 4 slabs
constexpr int kOffBias = kOffY + 4 * kSlab;          // fp32 [128 conv2 | 512 conv3 | 128 conv1]
constexpr int kSmemTotal = kOffBias + 4096;
static_assert((2 * kCM + kCB) * 4 <= 4096 && kSmemTotal <= 160 * 1024, "LDS budget");
constexpr int kNW = 8, kNT = 512;
constexpr int kRing = 8;
constexpr int kDumpBytes = kPx * kCB * 2;            // one Y row per tile pixel: 128 KB
// d_wfrag sections (bytes)
constexpr int kWfA2 = 0;                             // conv2: [4 channel groups][72 steps][64][8]
constexpr int kWfA3 = kWfA2 + 4 * kStepsA * 1024;    // conv3: [2 halves][8 groups][8 steps][64][8]
constexpr int kWfA1 = kWfA3 + 2 * 8 * 8 * 1024;      // conv1': [4 groups][32 steps][64][8]

struct MidBlkDev {
    const char* in;      // bf16 [N, H+2, W+2, 128]  U
    const char* res;     // bf16 [N, H+2, W+2, 512]  X
    const char* wf;
    const float* b2;     // [128]
    const float* b3;     // [512]
    const float* b1;     // [128]
    char* out;           // bf16 [N, H+2, W+2, 512]  Y
    char* next;          // bf16 [N, H+2, W+2, 128]  Z  (HEAD)
    char* dump;          // >= kDumpBytes
    int N, H, W, tiles_x, tiles_per_img, tiles;
    unsigned max_pix;
};

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    f32x2 v = {a, b};
    bf16x2 r = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(unsigned, r);
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Vector-memory program order of a lane in tile k (loads / DMA: L, stores: S):
//   A0  conv3 fragments of half 0        8 L   (registers a3; free since GEMM1(half 1) of tile k-1)
//   A1  conv2 fragments                 72 L   ring of 8: W(j) is awaited with vmcnt(min(7, 71 - j)) -- only later W are younger
//   P   patch of tile k+1                7 L   behind the barrier that retires the patch
//   -- half 0 (round 5: ONE operation per wave behind every GEMM step instead of bursts -- a wave needs ~250 cycles to issue a
//      load once the CU's address path is backed up, and nothing else of that wave runs meanwhile):
//       GEMM1 (a3 is older than every W: landed), step g = 0..15:  g < 7: P(g) (patch piece of tile k+1) | g < 8: R0(g) (conv1'
//       fragment, ring slot g) | g >= 8: X1(g - 8) (shortcut rows of half 1 -> rr; half 0's were parked before GEMM1)
//       A3 (conv3 fragments of half 1): two behind every channel group of GEMM1's last epilogue
//       GEMM2 step j: [wait a1(j)] MFMAs | j < 8: refill a1(j + 8) | j even: S0 row pass j / 2.
//       a1(j) awaited with vmcnt(23 + ceil(j / 2) + max(0, 6 - j)) for j < 8 and vmcnt(19 - j) for j >= 8.  P, X1 and A3 are
//       OLDER than every refill: complete behind the waits of j >= 8.
//   -- half 1:  rr -> Y buffer;  GEMM1 with R1 / X2 (rows of tile k+1, half 0);  GEMM2 with S1: vmcnt(15 + ceil(j / 2)) for j < 8,
//       vmcnt(19 - j) for j >= 8; X2 is older than the refills, i.e. complete for tile k+1 -- its top needs a barrier only
//   S2  Z rows                           4 S   (HEAD)
template <bool HEAD>
__global__ void __launch_bounds__(512, 2) conv_blk_mid_kernel(MidBlkDev P) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, half = lane >> 5;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const int Wp = P.W + 2;
    const int G = gridDim.x;
    const int my_tiles = (P.tiles - (int)blockIdx.x + G - 1) / G;
    const int cg = wave & 3, rp = wave >> 2;                  // phase A: channel group x row pair; GEMM2: channel group x pixel half

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
    auto pix_index = [&](const TileXY& T, int px) {           // haloed pixel index, clamped into the image (loads)
        int r = T.row0 + (px >> 5), c = T.col0 + (px & 31);
        r = r < P.H ? r : P.H - 1;
        c = c < P.W ? c : P.W - 1;
        return (unsigned)((T.img * (P.H + 2) + r + 1) * Wp + c + 1);
    };
    auto pix_valid = [&](const TileXY& T, int px) { return T.row0 + (px >> 5) < P.H && T.col0 + (px & 31) < P.W; };

    {
        float* lb = (float*)(lds + kOffBias);
        if (tid < kCM) lb[tid] = P.b2[tid];
        lb[kCM + tid] = P.b3[tid];
        if (HEAD && tid < kCM) lb[kCM + kCB + tid] = P.b1[tid];
    }
    const float* lbias = (const float*)(lds + kOffBias);

    // ---- patch DMA: 52 pieces (2 slabs x 26) of 8 patch pixels x 128 B; wave w moves pieces w, w + 8, .. (7 per wave, the
    // surplus ones repeat the last piece: every wave issues the same number of DMAs)
    auto issue_patch1 = [&](const TileXY& T, int ii) {
        {
            int pi = wave + kNW * ii;
            pi = pi < 2 * kPPieces ? pi : 2 * kPPieces - 1;
            const int sl = pi >= kPPieces ? 1 : 0;
            const int pc = pi - sl * kPPieces;
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int pp = pc * 8 + (ln >> 3);
            const int p = (pp * 1928) >> 16;                   // pp / 34 for pp < 344
            const int q = pp - p * kPC;
            unsigned g = (unsigned)((T.img * (P.H + 2) + T.row0 + p) * Wp + T.col0 + q);
            g = g < P.max_pix ? g : P.max_pix;
            __builtin_amdgcn_global_load_lds((gvoid*)(P.in + (size_t)g * (kCM * 2) + sl * 128 + (unsigned)(((ln & 7) ^ ((q >> 1) & 7)) * 16)),
                                             (lvoid*)(lds + kOffPatch + sl * kPSlab + pc * 1024), 16, 0, 0);
        }
    };
    auto issue_patch = [&](const TileXY& T) {
#pragma unroll
        for (int ii = 0; ii < 7; ii++) issue_patch1(T, ii);
    };
    // shortcut rows of half h: pass i of 8, 32 threads read one pixel's 512 B
    u32x4 rr[8];
#pragma unroll
    for (int i = 0; i < 8; i++) rr[i] = u32x4{0u, 0u, 0u, 0u};
    auto issue_x1 = [&](const TileXY& T, int h, auto I) {
        constexpr int i = decltype(I)::value;
        u32x4(&rq)[8] = rr;                  // (a non-dependent use: a generic lambda captures rr only through one)
        int idx = tid + kNT * i;
        asm volatile("" : "+v"(idx));
        const char* src = P.res + (size_t)pix_index(T, idx >> 5) * (kCB * 2) + h * 512 + (idx & 31) * 16;
        asm volatile("global_load_dwordx4 %0, %1, off nt" : "+v"(rq[i]) : "v"(src) : "memory");
    };
    auto issue_x = [&](const TileXY& T, int h) {
        static_for<0, 8>([&](auto I) { issue_x1(T, h, I); });
    };
    auto park_x = [&]() {                                      // rr -> Y buffer ([slab][px][64 ch], chunk ^ ((px >> 1) & 7))
#pragma unroll
        for (int i = 0; i < 8; i++) {
            asm volatile("" : "+v"(rr[i]));
            const int idx = tid + kNT * i;
            const int px = idx >> 5, j = idx & 31;
            const unsigned ad = lds_base + (unsigned)(kOffY + (j >> 3) * kSlab + px * 128 + (((j & 7) ^ ((px >> 1) & 7)) * 16));
            asm volatile("ds_write_b128 %0, %1" ::"v"(ad), "v"(rr[i]) : "memory");
        }
    };

    // ---- weight fragment loads, L2 -> registers (inline asm: readiness is tracked by hand).  ONE scalar base (d_wfrag) and
    // a per-lane byte offset: with a scalar base per fragment the compiler precomputes all 120 of them ahead of the tile loop
    // and spills the SGPR pairs into VGPR lanes (and the conv1' bases came back corrupted: memory faults at garbage + the
    // section offset); the per-tile opaque copies below keep the offsets from being hoisted the same way.
    bf16x8 wr[kRing], a3[8];
#pragma unroll
    for (int k = 0; k < kRing; k++) wr[k] = bf16x8{};
#pragma unroll
    for (int k = 0; k < 8; k++) a3[k] = bf16x8{};
    auto load_frag = [&](bf16x8& dst, unsigned vofs) {
        asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(dst) : "v"(vofs), "s"(P.wf) : "memory");
    };
    unsigned vo2 = 0, vo3 = 0, vo1 = 0;                        // set at the top of every tile
    auto load_a3 = [&](int h) {
#pragma unroll
        for (int s = 0; s < 8; s++) load_frag(a3[s], vo3 + (unsigned)((h * 64 + s) * 1024));
    };

    unsigned bs[4];                          // B fragment of k16 step s inside a [128 px][128 B] slab, pixel fragment 0
#pragma unroll
    for (int s = 0; s < 4; s++) bs[s] = (unsigned)(frow * 128 + (((2 * s + half) ^ ((frow >> 1) & 7)) * 16));
    // B fragments of two neighbouring pixel fragments: a per-(k16 step) base register + an immediate (region bases and the
    // wave's pixel half live in the registers: the 16-bit offset field holds slab + fragment)
    unsigned tbase[4], ybase[4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        tbase[s] = lds_base + (unsigned)kOffT + bs[s];
        ybase[s] = lds_base + (unsigned)(kOffY + (2 * rp) * 4096) + bs[s];
    }
    auto bread2 = [&](bf16x8& b0, bf16x8& b1, unsigned base, int off) {
        switch (off >> 12) {          // off = multiple of 4096 in 0 .. 15 * 4096: immediates need compile-time constants
#define BREAD2_CASE(K) case K: asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4\n\ts_waitcnt lgkmcnt(0)" \
                                            : "=&v"(b0), "=&v"(b1) : "v"(base), "n"(K * 4096), "n"(K * 4096 + 4096) : "memory"); break;
            BREAD2_CASE(0) BREAD2_CASE(2) BREAD2_CASE(4) BREAD2_CASE(6) BREAD2_CASE(8) BREAD2_CASE(10) BREAD2_CASE(12) BREAD2_CASE(14)
#undef BREAD2_CASE
            default: __builtin_unreachable();
        }
    };
    auto barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    if (my_tiles > 0) {
        const TileXY T0 = tile_xy((int)blockIdx.x);
        issue_patch(T0);
        issue_x(T0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

#ifdef DAFNE_MID_TIMING
    unsigned long long mstamp[24];
    int nms = 0;
#define MID_STAMP() do { if (nms < 24) mstamp[nms++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MID_STAMP()
#endif
    for (int kk = 0; kk < my_tiles; kk++) {
#ifdef DAFNE_MID_TIMING
        nms = 0;
#endif
        MID_STAMP();                                                   // 0: tile start
        const int t = (int)blockIdx.x + kk * G;
        const TileXY T = tile_xy(t);
        const TileXY Tn = tile_xy(kk + 1 < my_tiles ? t + G : t);      // the last tile re-requests itself: fixed instruction count
        barrier();       // patch k (every wave's pieces: each drained its queue at the end of the previous tile) visible; LDS of tile k-1 retired
        MID_STAMP();                                                   // 1: behind the top barrier
        // ================================================================ phase A: T = relu(conv2(U) + bias2)
        {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            vo2 = (unsigned)(kWfA2 + cg * kStepsA * 1024 + ln * 16);
            vo3 = (unsigned)(kWfA3 + wave * 8 * 1024 + ln * 16);
            vo1 = (unsigned)(kWfA1 + cg * 32 * 1024 + ln * 16);
        }
        load_a3(0);                                                    // A0
        static_for<0, kRing>([&](auto J) { load_frag(wr[decltype(J)::value], vo2 + (unsigned)(decltype(J)::value * 1024)); });
        {
            f32x16 acc[2];
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int k = 0; k < 16; k++) acc[r][k] = 0.f;
            // patch row p (0..5) at tap column kw: pixel p * 34 + q, q = kw + frow; chunk (2 kc + half) ^ ((q >> 1) & 7).  The
            // wave's first row (2 rp) is folded into the base, slab / tap row / second row are immediates: three address
            // registers per tile (recomputed per tile from an opaque copy of frow: hoisted out of the tile loop the 72
            // addresses of the unrolled steps spill, and scratch traffic counts in vmcnt).
            int fr = frow;
            asm volatile("" : "+v"(fr));
            unsigned pb[3];
#pragma unroll
            for (int kw = 0; kw < 3; kw++) {
                const int q = kw + fr;
                const int sw = (q >> 1) & 7;
                pb[kw] = lds_base + (unsigned)(kOffPatch + (2 * rp * kPC + q) * 128 + ((half ^ (sw & 1)) << 4) + ((sw >> 1) << 5));
            }
            static_for<0, kStepsA>([&](auto J) {
                constexpr int j = decltype(J)::value;
                constexpr int sl = j / 36, tt = j % 36, kh = tt / 12, kw = (tt >> 2) % 3, kc = tt & 3;
                constexpr int wn = (kStepsA - 1 - j) < (kRing - 1) ? (kStepsA - 1 - j) : (kRing - 1);
                constexpr int o0 = sl * kPSlab + kh * kPC * 128, o1 = o0 + kPC * 128;
                asm volatile("s_waitcnt vmcnt(%1)" : "+v"(wr[j % kRing]) : "n"(wn) : "memory");
                const unsigned ad = pb[kw] ^ (unsigned)(kc << 5);       // (bits 5..6 of everything but the swizzle term are zero)
                bf16x8 b0, b1;
                asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(b0), "=&v"(b1) : "v"(ad), "n"(o0), "n"(o1) : "memory");
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[j % kRing], b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[j % kRing], b1, acc[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (j + kRing < kStepsA) load_frag(wr[j % kRing], vo2 + (unsigned)((j + kRing) * 1024));
            });
            MID_STAMP();                                               // 2: phase A MFMA loop done (this wave)
            // (acc + bias2) -> ReLU -> bf16 -> T tile: channels cg * 32 .. = half (cg & 1) of slab (cg >> 1)
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const int px = (2 * rp + r) * 32 + frow;
                const unsigned tb = lds_base + (unsigned)(kOffT + (cg >> 1) * kSlab + px * 128 + 8 * half);
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const float* bp = lbias + cg * 32 + 8 * g + 4 * half;
                    const float v0 = fmaxf(acc[r][4 * g] + bp[0], 0.f), v1 = fmaxf(acc[r][4 * g + 1] + bp[1], 0.f);
                    const float v2 = fmaxf(acc[r][4 * g + 2] + bp[2], 0.f), v3 = fmaxf(acc[r][4 * g + 3] + bp[3], 0.f);
                    u32x2 pk;
                    pk.x = pack_bf16(v0, v1);
                    pk.y = pack_bf16(v2, v3);
                    const unsigned ad = tb + (unsigned)(((((cg & 1) * 4 + g) ^ ((px >> 1) & 7))) * 16);
                    asm volatile("ds_write_b64 %0, %1" ::"v"(ad), "v"(pk) : "memory");
                }
            }
        }
        barrier();       // T complete; every wave is done with the patch (its next tile's pieces: P, inside GEMM1 of half 0)
        park_x();                                                      // shortcut rows of half 0 (loaded during the previous tile)
        barrier();
        MID_STAMP();                                                   // 3: T complete, half 0 rows parked

        // ================================================================ phase B: two 256-channel halves
        f32x16 acc2[2];
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int k = 0; k < 16; k++) acc2[b][k] = 0.f;
        static_for<0, 2>([&](auto HH) {
            constexpr int h = decltype(HH)::value;
            if constexpr (h == 1) {
                park_x();                                              // rows of half 1 (X1: complete behind GEMM2's vmcnt(0))
                barrier();
            }
            // ---- round 5: ONE vector-memory operation per wave behind every GEMM1 step -- the next tile's patch pieces (P, half 0), the
            // conv1' fragments (R: the ring is free since phase A / the previous GEMM2) and the next shortcut rows (X: rr is free
            // since park_x) -- instead of bursts of 16-24 per wave with the matrix pipe idle: a wave needs ~250 cycles to ISSUE
            // a load once the CU's address path is backed up (phase stamps, NOTES_r05)
            // ---- GEMM1: Y half (32 channels of this wave x 128 px, two 64-pixel halves) = W3 . T, then in place in the Y buffer
            // (acc + bias3) + X -> ReLU -> bf16
#pragma unroll
            for (int ph2 = 0; ph2 < 2; ph2++) {
                f32x16 acc1[2];
#pragma unroll
                for (int b = 0; b < 2; b++)
#pragma unroll
                    for (int k = 0; k < 16; k++) acc1[b][k] = 0.f;
                static_for<0, 8>([&](auto SS) {
                    constexpr int s = decltype(SS)::value;
                    bf16x8 b0, b1;
                    bread2(b0, b1, tbase[s & 3], (s >> 2) * kSlab + (2 * ph2) * 4096);
                    acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[s], b0, acc1[0], 0, 0, 0);
                    acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[s], b1, acc1[1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (ph2 == 0) {
                        if constexpr (h == 0 && s < 7) issue_patch1(Tn, s);                                     // P
                        load_frag(wr[s], vo1 + (unsigned)((h * 16 + s) * 1024));                                // R0 / R1
                    } else {
                        if constexpr (h == 0) issue_x1(T, 1, SS);                                               // X1
                        else issue_x1(Tn, 0, SS);                                                               // X2
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                MID_STAMP();                                           // GEMM1 quarter done (wave 0)
                const unsigned ebase = lds_base + (unsigned)(kOffY + (wave >> 1) * kSlab + (2 * ph2) * 4096 + frow * 128 + 8 * half);
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const unsigned ead = ebase + (unsigned)(((((wave & 1) * 4 + g) ^ ((frow >> 1) & 7))) * 16);
                    const f32x4 bv = *(const f32x4*)(lbias + kCM + h * 256 + wave * 32 + 8 * g + 4 * half);
                    u32x2 rc[2];
                    asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:4096\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(rc[0]), "=&v"(rc[1]) : "v"(ead) : "memory");
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
                    // A3: the conv3 fragments of half 1, two behind every group of the LAST epilogue of half 0 (a3 was GEMM1's operand
                    // until the step loop above ended)
                    if (h == 0 && ph2 == 1) {
                        switch (g) {
#define A3_CASE(G) case G: load_frag(a3[2 * G], vo3 + (unsigned)((64 + 2 * G) * 1024)); load_frag(a3[2 * G + 1], vo3 + (unsigned)((64 + 2 * G + 1) * 1024)); break;
                            A3_CASE(0) A3_CASE(1) A3_CASE(2) A3_CASE(3)
#undef A3_CASE
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            MID_STAMP();                                               // both epilogues done (wave 0)
            MID_STAMP();                                               // loads issued (wave 0)
            barrier();       // the Y half is complete
            MID_STAMP();                                               // 4 / 7: GEMM1 + epilogue of the half done (all waves)
            // ---- Y half rows -> HBM: pass i of 8, 32 threads write one pixel's 512 B (exactly 8 stores per lane)          S0 / S1
            // round 5: one pass behind every second GEMM2 step instead of eight in a row in front of it
            auto store_rows = [&](int i) {
                int idx = tid + kNT * i;
                asm volatile("" : "+v"(idx));
                const int px = idx >> 5, j = idx & 31;
                u32x4 v;
                const unsigned ad = lds_base + (unsigned)(kOffY + (j >> 3) * kSlab + px * 128 + (((j & 7) ^ ((px >> 1) & 7)) * 16));
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ad) : "memory");
                char* a = P.out + (size_t)pix_index(T, px) * (kCB * 2);
                char* d = P.dump + (size_t)px * (kCB * 2);
                a = pix_valid(T, px) ? a : d;
                __builtin_nontemporal_store(v, (u32x4*)(a + h * 512 + j * 16));
            };
            MID_STAMP();                                               // 5 / 8: (rows are issued inside GEMM2 now)
            // ---- GEMM2 over this half's K range: Z (32 channels cg x 64 px rp) += W1[:, half h] . Y half.  Step s: [wait a1(s)] MFMAs |
            // refill a1(s + 8) (s < 8) | row pass s / 2 (s even).  Younger than a1(s), s < 8: the patch pieces behind it (half 0: 6 - s),
            // the rest of R (7 - s), X (8), A3 (8, half 0), s refills, ceil(s / 2) row passes; s >= 8: 15 - s refills and 4 row passes
            static_for<0, 16>([&](auto SS) {
                constexpr int s = decltype(SS)::value;
                constexpr int wn = s < 8 ? (h == 0 ? 23 + (s < 6 ? 6 - s : 0) : 15) + (s + 1) / 2 : 19 - s;
                asm volatile("s_waitcnt vmcnt(%1)" : "+v"(wr[s % kRing]) : "n"(wn) : "memory");
                bf16x8 b0, b1;
                bread2(b0, b1, ybase[s & 3], (s >> 2) * kSlab);
                if constexpr (HEAD) {
                    acc2[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[s % kRing], b0, acc2[0], 0, 0, 0);
                    acc2[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[s % kRing], b1, acc2[1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (s < 8) load_frag(wr[s % kRing], vo1 + (unsigned)((h * 16 + s + 8) * 1024));
                if constexpr ((s & 1) == 0) store_rows(s >> 1);
                __builtin_amdgcn_sched_barrier(0);
            });
            barrier();       // every wave is done with the Y half (row stores and GEMM2 have read it)
            MID_STAMP();                                               // 6 / 9: GEMM2 of the half done (all waves)
        });
        // ================================================================ Z = relu(acc2 + bias1) -> staging (the T tile's LDS) -> rows
        if constexpr (HEAD) {
#pragma unroll
            for (int b = 0; b < 2; b++) {
                const int px = (2 * rp + b) * 32 + frow;
                const unsigned zb = lds_base + (unsigned)(kOffT + (cg >> 1) * kSlab + px * 128 + 8 * half);
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const float* bp = lbias + kCM + kCB + cg * 32 + 8 * g + 4 * half;
                    const float v0 = fmaxf(acc2[b][4 * g] + bp[0], 0.f), v1 = fmaxf(acc2[b][4 * g + 1] + bp[1], 0.f);
                    const float v2 = fmaxf(acc2[b][4 * g + 2] + bp[2], 0.f), v3 = fmaxf(acc2[b][4 * g + 3] + bp[3], 0.f);
                    u32x2 pk;
                    pk.x = pack_bf16(v0, v1);
                    pk.y = pack_bf16(v2, v3);
                    const unsigned ad = zb + (unsigned)(((((cg & 1) * 4 + g) ^ ((px >> 1) & 7))) * 16);
                    asm volatile("ds_write_b64 %0, %1" ::"v"(ad), "v"(pk) : "memory");
                }
            }
            barrier();
#pragma unroll
            for (int i = 0; i < 4; i++) {           // 16 threads write one pixel's 256 B (exactly 4 stores per lane)       S2
                int idx = tid + kNT * i;
                asm volatile("" : "+v"(idx));
                const int px = idx >> 4, j = idx & 15;
                u32x4 v;
                const unsigned ad = lds_base + (unsigned)(kOffT + (j >> 3) * kSlab + px * 128 + (((j & 7) ^ ((px >> 1) & 7)) * 16));
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ad) : "memory");
                char* a = P.next + (size_t)pix_index(T, px) * (kCM * 2);
                char* d = P.dump + (size_t)px * (kCM * 2);
                a = pix_valid(T, px) ? a : d;
                *(u32x4*)(a + j * 16) = v;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        MID_STAMP();                                                   // 10: Z rows issued
#ifdef DAFNE_MID_TIMING
        if (tid == 0 && kk == 1 && blockIdx.x < 64) {                  // the second tile of a workgroup: steady state
            unsigned long long* o = (unsigned long long*)P.dump + blockIdx.x * 24;
            for (int q = 0; q < 23; q++) o[q] = mstamp[q] - mstamp[0];
        }
#endif
    }
    // nothing may still be on its way into this workgroup's LDS (the last tile re-requested its own patch) or registers
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(rr[0]), "+v"(rr[1]), "+v"(rr[2]), "+v"(rr[3]), "+v"(rr[4]), "+v"(rr[5]), "+v"(rr[6]), "+v"(rr[7]) :: "memory");
}

}  // namespace

extern "C" {

size_t dafne_bottleneck_block_mid_scratch_bytes(void) { return (size_t)kDumpBytes; }

int dafne_bottleneck_block_mid_hip(const void* d_in, const void* d_res, const void* d_wfrag, const float* d_bias2,
                                   const float* d_bias3, const float* d_bias1, int n_images, int H, int W, void* d_out,
                                   void* d_next, void* d_scratch, size_t scratch_bytes, void* stream) {
    const bool head = d_next != nullptr;
    if (!d_in || !d_res || !d_wfrag || !d_bias2 || !d_bias3 || !d_out || !d_scratch || (head && !d_bias1))
        return dafne::fail(DAFNE_E_INVALID, "bottleneck_block_mid: null argument");
    if (n_images < 1 || H < 1 || W < 1 || (long long)H * W > (1 << 20)) return dafne::fail(DAFNE_E_INVALID, "bottleneck_block_mid: bad size");
    if (scratch_bytes < (size_t)kDumpBytes) return dafne::fail(DAFNE_E_WORKSPACE, "bottleneck_block_mid: scratch %zu < %d", scratch_bytes, kDumpBytes);
    MidBlkDev D;
    D.in = (const char*)d_in; D.res = (const char*)d_res; D.wf = (const char*)d_wfrag;
    D.b2 = d_bias2; D.b3 = d_bias3; D.b1 = d_bias1;
    D.out = (char*)d_out; D.next = (char*)d_next; D.dump = (char*)d_scratch;
    D.N = n_images; D.H = H; D.W = W;
    D.tiles_x = (W + kTW - 1) / kTW;
    D.tiles_per_img = D.tiles_x * ((H + kTH - 1) / kTH);
    const long long tiles = (long long)D.tiles_per_img * n_images;
    const long long pix = (long long)n_images * (H + 2) * (W + 2);
    if (tiles > (1ll << 24) || pix * (kCB * 2) > 0xffffffffll) return dafne::fail(DAFNE_E_UNSUPPORTED, "bottleneck_block_mid: too large");
    D.tiles = (int)tiles;
    D.max_pix = (unsigned)(pix - 1);
    DAFNE_MAX_LDS_ONCE(kSmemTotal, (const void*)conv_blk_mid_kernel<true>, (const void*)conv_blk_mid_kernel<false>);
    int n_cu = 0;
    if (int rc = dafne::device_cus(&n_cu)) return rc;
    const int grid = D.tiles < n_cu ? D.tiles : n_cu;
    if (head) hipLaunchKernelGGL(conv_blk_mid_kernel<true>, dim3(grid), dim3(kNT), kSmemTotal, (hipStream_t)stream, D);
    else hipLaunchKernelGGL(conv_blk_mid_kernel<false>, dim3(grid), dim3(kNT), kSmemTotal, (hipStream_t)stream, D);
    return dafne::check_launch("conv_blk_mid");
}

}  // extern "C"
