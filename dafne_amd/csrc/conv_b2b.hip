// Bottleneck tail + next bottleneck head in one kernel (gfx950), for the res4 stage of ResNet-50/101
// [detectron2 BottleneckBlock, recalled; built by build_dafne_resnet_fpn_backbone, backbone/fpn.py:58-91]:
//
//     Y = relu(conv3_b(T) + bias3 + X)            1x1, 256 -> 1024, X = the block's input (identity shortcut)
//     Z = relu(conv1_{b+1}(Y) + bias1)            1x1, 1024 -> 256
//
// Unfused these are two small launches per block (17 GFLOP each at batch 8, 64x64 maps): one round of tiles whose
// duration is prologue + epilogue + operand latency, and Y (67 MB) is written, then read straight back.  Here a
// workgroup owns 64 pixels and ALL channels: for each 256-channel chunk of Y it runs GEMM1 (K = 256, operand T
// resident in LDS), adds bias + residual in place in LDS, stores the chunk to HBM and immediately uses the LDS copy
// as the K-chunk of GEMM2, whose accumulators (64 px x 256 cout) live in registers across the four chunks.  Y is
// never read back; the second launch, its prologue and its epilogue disappear.
//
//   * A workgroup owns 128 pixels: 8 waves, one workgroup per CU.  A wave owns 32 output channels (ONE A fragment) of
//     every 256-channel chunk and all 128 pixels (4 pixel fragments): every weight fragment a CU fetches feeds 4 MFMAs.
//     The first version (64 pixels, 4 waves x 2 workgroups per CU, 2 A + 2 B fragments per 4 MFMAs) streamed the
//     whole 1 MB of weights through every CU TWICE (once per resident workgroup) and sat at the L2 -> CU ingest limit
//     (~42 B/clk/CU: 16 KB of weights per CU and k16 step = 390 cycles against 256 cycles of MFMA issue); here a CU
//     fetches 8 KB per step and the matrix pipe is the longer pole.
//   * LDS per workgroup: T tile 64 KB + Y chunk 64 KB (four [128 px][128 B] slabs each, 16-byte chunk ^ ((px >> 1) & 7):
//     conflict-free for the 16-lane groups ds_read_b128 is serviced in over a 256-byte bank row) + 5 KB of biases.
//     The residual chunk is DMA'd (global_load_lds) into the Y buffer under GEMM1 and updated in place by the epilogue.
//   * No LDS left for a weight ring, so the A operand streams L2 -> REGISTERS: the host packs both weight matrices
//     fragment-major (engine.pack_b2b: [phase][wave][k16 step][lane][8 bf16]); one coalesced 1-KiB load per wave and
//     k16 step through a ring of 8 steps (1 A + 4 B fragments per 4 MFMAs).
//   * K is walked in ascending order in both GEMMs and the epilogue expressions are those of the separate kernels:
//     results are bit-identical to conv3 (+residual) followed by conv1.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((address_space(1))) void gvoid;
typedef __attribute__((address_space(3))) void lvoid;

constexpr int kPx = 128;                 // pixels per workgroup
constexpr int kSlab = kPx * 128;         // one 64-channel slab of the tile: 16 KB
constexpr int kBuf = 4 * kSlab;          // 256 channels: 64 KB
constexpr int kSmem = 2 * kBuf;          // + 5 KB of biases behind it
constexpr int kSmemTotal = kSmem + 8 * 1024;      // biases: [1024 conv3 | 256 conv1 | 3 x 256 duplicates of conv1 (one DMA piece per wave)]
constexpr int kCM = 256, kCB = 1024, kChunks = kCB / 256;
constexpr int kNW = 8, kNT = 512;
constexpr int kPF = kPx / 32;            // pixel fragments per wave
constexpr int kPhaseBytes = kNW * 16 * 1024;        // one phase of the fragment-major weights: 8 waves x 16 steps x 1 KB

struct B2bDev {
    const char* in;      // bf16 [N, H+2, W+2, 256]
    const char* res;     // bf16 [N, H+2, W+2, 1024]
    const char* wf;      // bf16 [8 phases][8 waves][16 steps][64 lanes][8]
    const float* b3;     // [1024]
    const float* b1;     // [256]
    char* out;           // bf16 [N, H+2, W+2, 1024]
    char* next;          // bf16 [N, H+2, W+2, 256]
    int N, H, W, tiles_per_img, tiles;
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    f32x2 v = {a, b};
    bf16x2 r = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(unsigned, r);
}

// Vector-memory program order of a workgroup (per lane): bias piece (1 DMA) | T tile DMA (8) | A(0) .. A(7) (1 load each) |
// R(0), the first residual chunk (8 DMA) | then for every k16 step j = 0 .. 127 (32 per chunk: 16 of GEMM1, 16 of GEMM2,
// the latter in four slab groups of 4 steps):
//   [wait A(j)] MFMAs | 2 row stores of slab q of the Y chunk at the FIRST step of slab group q | A(j + 8) if j + 8 < 128 |
//   after the LAST step of slab group q of chunk c < 3 (and a barrier): 2 DMA pieces of slab q of R(c + 1) -- the next
//   residual chunk replaces the Y chunk slab by slab as GEMM2 is done with it, so its HBM latency hides under
//   GEMM2(c) and GEMM1(c + 1) (issued after the whole chunk it was a burst of 64 KB per CU, on every CU at once,
//   that the next epilogue then waited for: chunks 1..3 took 19-20k cycles against 14.6k for chunk 0).
// b2b_wait(j) = number of those instructions issued after A(j) and before the wait for it: vmcnt is in-order, so
// `s_waitcnt vmcnt(b2b_wait(j))` is exactly "A(j) and everything older has landed".
// experiment switches (scratch/build_variant.sh): -DDAFNE_B2B_STROW  Y chunk stored as whole 512-B pixel rows behind the first
// four GEMM2 steps (round 1's form) instead of slab by slab; -DDAFNE_B2B_RESW  next residual chunk issued whole after GEMM2
#ifdef DAFNE_B2B_STROW
constexpr bool kStRow = true;
#else
constexpr bool kStRow = false;
#endif
#ifdef DAFNE_B2B_RESW
constexpr bool kResWhole = true;
#else
constexpr bool kResWhole = false;
#endif
constexpr int b2b_st(int i) {
    return kStRow ? (((i & 31) >= 16 && (i & 31) <= 19) ? 2 : 0) : (((i & 31) >= 16 && ((i & 31) & 3) == 0) ? 2 : 0);
}
constexpr int b2b_res(int i) {
    return kResWhole ? (((i & 31) == 31 && i < 127) ? 8 : 0) : (((i & 31) >= 16 && ((i & 31) & 3) == 3 && (i >> 5) < 3) ? 2 : 0);
}
#ifdef DAFNE_B2B_RING
constexpr int kRing = DAFNE_B2B_RING;
#else
constexpr int kRing = 8;                 // k16 steps of A fragments in flight per wave
#endif
constexpr int b2b_wait(int j) {
    int n = 0;
    if (j < kRing) {
        n += (kRing - 1 - j) + 8;                              // A(j+1..kRing-1), R(0)
        for (int i = 0; i < j; i++) n += b2b_st(i) + (i + kRing < 128 ? 1 : 0) + b2b_res(i);
    } else {
        n += b2b_res(j - kRing);                               // R pieces right behind A(j) at the end of step j - kRing
        for (int i = j - kRing + 1; i < j; i++) n += b2b_st(i) + (i + kRing < 128 ? 1 : 0) + b2b_res(i);
    }
    return n;
}
static_assert(kRing != 8 || kStRow || kResWhole || (b2b_wait(0) == 15 && b2b_wait(8) == 7 && b2b_wait(127) == 4 && b2b_wait(39) == 9 && b2b_wait(27) == 15 && b2b_wait(24) == 13), "vmcnt bookkeeping");

__global__ void __launch_bounds__(512, 2) conv_b2b_kernel(B2bDev P) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, half = lane >> 5;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;

    const int tile = xcd_remap(blockIdx.x, P.tiles);
    const int img = tile / P.tiles_per_img;
    const int m0 = (tile - img * P.tiles_per_img) * kPx;
    const int HW = P.H * P.W;
    const int Wp = P.W + 2;
    const float invW = 1.0f / (float)P.W;

    auto halo_index = [&](int px) {          // haloed pixel index of tile pixel px (clamped into the image)
        int m = m0 + px;
        m = m < HW ? m : HW - 1;
        const int ho = (int)(((float)m + 0.5f) * invW), wo = m - ho * P.W;    // exact for H*W <= 2^20 (see conv.hip divmod_small)
        return (unsigned)((img * (P.H + 2) + ho + 1) * Wp + wo + 1);
    };

    // ---- DMA maps: a slab is 16 pieces of 8 px x 128 B; wave w moves pieces w and w + 8 of every slab
    unsigned dpix[2], dq[2];
#pragma unroll
    for (int ii = 0; ii < 2; ii++) {
        const int px = (wave + kNW * ii) * 8 + (lane >> 3);
        dpix[ii] = halo_index(px);
        dq[ii] = (unsigned)(((lane & 7) ^ ((px >> 1) & 7)) * 16);
    }
    auto dma_tile = [&](const char* src, unsigned pix_bytes, unsigned col0, int buf) {
#pragma unroll
        for (int sl = 0; sl < 4; sl++)
#pragma unroll
            for (int ii = 0; ii < 2; ii++)
                __builtin_amdgcn_global_load_lds((gvoid*)(src + (size_t)dpix[ii] * pix_bytes + col0 + sl * 128 + dq[ii]),
                                                 (lvoid*)(lds + buf * kBuf + sl * kSlab + (wave + kNW * ii) * 1024), 16, 0, 0);
    };

    auto dma_slab = [&](const char* src, unsigned pix_bytes, unsigned col0, int buf, int sl) {      // 2 pieces per wave
#pragma unroll
        for (int ii = 0; ii < 2; ii++)
            __builtin_amdgcn_global_load_lds((gvoid*)(src + (size_t)dpix[ii] * pix_bytes + col0 + sl * 128 + dq[ii]),
                                             (lvoid*)(lds + buf * kBuf + sl * kSlab + (wave + kNW * ii) * 1024), 16, 0, 0);
    };

    // ---- fragment offsets
    unsigned bs[4];                          // B fragment of k16 step s inside a slab, pixel fragment 0
#pragma unroll
    for (int s = 0; s < 4; s++) bs[s] = (unsigned)(frow * 128 + (((2 * s + half) ^ ((frow >> 1) & 7)) * 16));
    // epilogue: the wave's 32 channels are half (wave & 1) of slab (wave >> 1) of the Y buffer; this lane's row + half inside it
    const unsigned ebase = lds_base + (unsigned)(kBuf + (wave >> 1) * kSlab + frow * 128 + 8 * half);

    // ---- A operand: L2 -> registers through inline asm (the compiler would sink visible loads to their uses and wait
    // for each).  A ring of 8 k16 steps (2 fragments each): the loads of step j + 8 are issued as soon as step j's MFMAs
    // are, so 7 steps (~1 us of matrix work for the two waves of a SIMD) cover the L2 latency.  Readiness is tracked by
    // hand (b2b_wait); stores of a ragged tile are clamped, not predicated, to keep the instruction count exact.
    const unsigned voff = (unsigned)(wave * 16 * 1024 + lane * 16);
    bf16x8 ar[kRing];             // ("+v" in the loads: one register per ring slot from this zero-initialisation on; conv.hip rp_load)
#pragma unroll
    for (int k = 0; k < kRing; k++) ar[k] = bf16x8{};
#define B2B_LOAD_STEP(j)                                                                                        \
    {                                                                                                           \
        const char* sb = P.wf + (size_t)((j) >> 4) * kPhaseBytes + ((j) & 15) * 1024;                           \
        asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(ar[(j) % kRing]) : "v"(voff), "s"(sb) : "memory");     \
    }
#define B2B_WAIT_STEP(j)                                                                                        \
    {                                                                                                           \
        constexpr int kWaitN = b2b_wait(j);                                                                     \
        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(ar[(j) % kRing]) : "n"(kWaitN) : "memory");                       \
    }

    f32x16 acc1[kPF], acc2[kPF];
#pragma unroll
    for (int b = 0; b < kPF; b++)
#pragma unroll
        for (int k = 0; k < 16; k++) acc2[b][k] = 0.f;

    auto barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // the Y buffer (256 channels of 128 px) -> global rows, one slab (64 channels) of 64 pixels per pass: 8 consecutive
    // threads write one pixel's 128 B.  EXACTLY 8 stores per lane and chunk (vmcnt bookkeeping): rows past the end of a
    // ragged tile re-write the last valid row.
    const int plast = HW - 1 - m0;
    auto store_slab = [&](int sl, int h, char* dst, unsigned pix_bytes, unsigned col0) {
        int idx = tid;
        asm volatile("" : "+v"(idx));                // recompute the row address at every pass: holding 8 of them spills
        int px = h * 64 + (idx >> 3);
        px = px < plast ? px : plast;
        const int q = idx & 7;
        const u32x4 v = *(const u32x4*)(lds + kBuf + sl * kSlab + px * 128 + ((q ^ ((px >> 1) & 7)) * 16));
        *(u32x4*)(dst + (size_t)halo_index(px) * pix_bytes + col0 + sl * 128 + q * 16) = v;
    };
    auto store_row = [&](int i, char* dst, unsigned pix_bytes, unsigned col0) {      // pass i of 8: 32 threads write one pixel's 512 B
        int idx = tid + kNT * i;
        asm volatile("" : "+v"(idx));
        int px = idx >> 5;
        px = px < plast ? px : plast;
        const int j = idx & 31;
        const int sl = j >> 3, q = j & 7;
        const u32x4 v = *(const u32x4*)(lds + kBuf + sl * kSlab + px * 128 + ((q ^ ((px >> 1) & 7)) * 16));
        *(u32x4*)(dst + (size_t)halo_index(px) * pix_bytes + col0 + j * 16) = v;
    };
    // k16 step: acc[b] += A . B[b]  (slab q of buffer buf, step s inside the slab)
    auto consume = [&](const bf16x8& a, int q, int st, int buf, f32x16* acc) {
        bf16x8 bfr[kPF];
#pragma unroll
        for (int b = 0; b < kPF; b++) bfr[b] = *(const bf16x8*)(lds + buf * kBuf + q * kSlab + b * 4096 + bs[st]);
#pragma unroll
        for (int b = 0; b < kPF; b++) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfr[b], acc[b], 0, 0, 0);
    };
    // acc + bias (+ residual already in the Y buffer) -> ReLU -> bf16, in place in the Y buffer
    const unsigned lbias_off = lds_base + (unsigned)kSmem;      // [1024 conv3 | 256 conv1] fp32
    auto epilogue = [&](f32x16* acc, int bias0, bool with_res) {
        typedef __attribute__((ext_vector_type(4))) float f32x4;
        typedef __attribute__((ext_vector_type(2))) float f32x2;
        const unsigned rmask = with_res ? 0xffffffffu : 0u;
#pragma unroll
        for (int gp = 0; gp < 2; gp++) {                 // two 8-channel groups at a time (register budget)
            f32x4 bv[2];
            u32x2 rc[2][kPF];
            unsigned ead[2];
            // LDS reads first, one wait.  Inline asm: a plain LDS read here makes the compiler drain
            // vmcnt (it cannot tell the read from the residual DMA's destination).  Pixel fragments sit 4096 B apart.
#pragma unroll
            for (int gg = 0; gg < 2; gg++) {
                const int g = 2 * gp + gg;
                ead[gg] = ebase + (unsigned)(((((wave & 1) * 4 + g) ^ ((frow >> 1) & 7))) * 16);
                const unsigned bad = lbias_off + (unsigned)((bias0 + wave * 32 + 8 * g + 4 * half) * 4);
                asm volatile("ds_read_b128 %4, %6\n\tds_read_b64 %0, %5\n\tds_read_b64 %1, %5 offset:4096\n\t"
                             "ds_read_b64 %2, %5 offset:8192\n\tds_read_b64 %3, %5 offset:12288"
                             : "=&v"(rc[gg][0]), "=&v"(rc[gg][1]), "=&v"(rc[gg][2]), "=&v"(rc[gg][3]), "=&v"(bv[gg])
                             : "v"(ead[gg]), "v"(bad)
                             : "memory");
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
                for (int b = 0; b < kPF; b++) {
                    u32x2 r = rc[gg][b];
                    r.x &= rmask;
                    r.y &= rmask;
                    const f32x2 rlo = {__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u)};
                    const f32x2 rhi = {__uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
                    const f32x2 alo = {acc[b][4 * g], acc[b][4 * g + 1]}, ahi = {acc[b][4 * g + 2], acc[b][4 * g + 3]};
                    const f32x2 vlo = alo + blo + rlo, vhi = ahi + bhi + rhi;          // (acc + bias) + residual
                    rc[gg][b].x = pack_bf16(fmaxf(vlo[0], 0.f), fmaxf(vlo[1], 0.f));
                    rc[gg][b].y = pack_bf16(fmaxf(vhi[0], 0.f), fmaxf(vhi[1], 0.f));
                }
                asm volatile("ds_write_b64 %4, %0\n\tds_write_b64 %4, %1 offset:4096\n\t"
                             "ds_write_b64 %4, %2 offset:8192\n\tds_write_b64 %4, %3 offset:12288"
                             ::"v"(rc[gg][0]), "v"(rc[gg][1]), "v"(rc[gg][2]), "v"(rc[gg][3]), "v"(ead[gg]) : "memory");
            }
        }
    };

#ifdef DAFNE_B2B_TIMING
    unsigned long long stamp[12];
    int nstamp = 0;
#define B2B_STAMP() stamp[nstamp++] = __builtin_amdgcn_s_memtime()
#else
#define B2B_STAMP()
#endif
    B2B_STAMP();
    // ---- prologue: biases -> LDS by DMA, one 1-KiB piece per wave (waves 0..3: conv3, 4..7: conv1, the last three
    // land in spare LDS); the oldest vector-memory operation of every wave, so every later wait covers it
    {
        const float* bsrc = wave < 4 ? P.b3 + wave * 256 : P.b1;
        __builtin_amdgcn_global_load_lds((gvoid*)(bsrc + lane * 4), (lvoid*)(lds + kSmem + wave * 1024), 16, 0, 0);
    }
    dma_tile(P.in, kCM * 2, 0, 0);                                     // T
    B2B_LOAD_STEP(0) B2B_LOAD_STEP(1) B2B_LOAD_STEP(2) B2B_LOAD_STEP(3)
    B2B_LOAD_STEP(4) B2B_LOAD_STEP(5) B2B_LOAD_STEP(6) B2B_LOAD_STEP(7)
    if (kRing > 8) { B2B_LOAD_STEP(8) B2B_LOAD_STEP(9) B2B_LOAD_STEP(10) B2B_LOAD_STEP(11) }
    if (kRing > 12) { B2B_LOAD_STEP(12) B2B_LOAD_STEP(13) B2B_LOAD_STEP(14) B2B_LOAD_STEP(15) }
    dma_tile(P.res, kCB * 2, 0, 1);                                    // R(0): not awaited here
    B2B_WAIT_STEP(0);                                                  // T and A(0)
    barrier();
    B2B_STAMP();

#define B2B_STEP(j, ACC)                                                                                        \
    {                                                                                                           \
        B2B_WAIT_STEP(j);                                                                                       \
        consume(ar[(j) % kRing], ((j) & 15) >> 2, (j) & 3, ((j) >> 4) & 1, ACC);                                    \
        if (!kStRow && ((j) & 31) >= 16 && (((j) & 31) & 3) == 0) {                            \
            store_slab((((j) & 31) - 16) >> 2, 0, P.out, kCB * 2, (unsigned)((j) >> 5) * 512u);                 \
            store_slab((((j) & 31) - 16) >> 2, 1, P.out, kCB * 2, (unsigned)((j) >> 5) * 512u);                 \
        }                                                                                                       \
        if (kStRow && ((j) & 31) >= 16 && ((j) & 31) <= 19) {                                                   \
            store_row(2 * ((j) & 3), P.out, kCB * 2, (unsigned)((j) >> 5) * 512u);                              \
            store_row(2 * ((j) & 3) + 1, P.out, kCB * 2, (unsigned)((j) >> 5) * 512u);                          \
        }                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        if ((j) + kRing < 128) B2B_LOAD_STEP((j) + kRing)                                                               \
    }
#define B2B_STEP4(j, ACC) B2B_STEP((j), ACC) B2B_STEP((j) + 1, ACC) B2B_STEP((j) + 2, ACC) B2B_STEP((j) + 3, ACC)
    /* GEMM2 over slab Q of the Y chunk; then (every wave done with the slab, its row stores have read it) the same slab
       of the NEXT residual chunk starts to land in its place */
#define B2B_G2SLAB(C, Q)                                                                                        \
    {                                                                                                           \
        B2B_STEP4(32 * (C) + 16 + 4 * (Q), acc2)                                                                \
        if (!kResWhole || (Q) == 3) barrier();                                                 \
        if (!kResWhole && (C) + 1 < kChunks) dma_slab(P.res, kCB * 2, (unsigned)((C) + 1) * 512u, 1, (Q));      \
        if (kResWhole && (Q) == 3 && (C) + 1 < kChunks) dma_tile(P.res, kCB * 2, (unsigned)((C) + 1) * 512u, 1);               \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
    }
#define B2B_CHUNK(C)                                                                                            \
    {                                                                                                           \
        /* GEMM1: Y chunk C = W3[C] . T  (K = 256 over the four slabs of the T tile) */                        \
        _Pragma("unroll") for (int b = 0; b < kPF; b++)                                                         \
        _Pragma("unroll") for (int k = 0; k < 16; k++) acc1[b][k] = 0.f;                                        \
        B2B_STEP4(32 * (C) + 0, acc1) B2B_STEP4(32 * (C) + 4, acc1)                                             \
        B2B_STEP4(32 * (C) + 8, acc1) B2B_STEP4(32 * (C) + 12, acc1)                                            \
        barrier(); /* every wave's residual pieces are in the Y buffer (older than A(32C+15)) */                \
        if ((C) == 0) B2B_STAMP();                                                                              \
        epilogue(acc1, (C) * 256, true);                                                                        \
        barrier();                                                                                              \
        if ((C) == 0) B2B_STAMP();                                                                              \
        /* GEMM2: Z += W1[:, chunk C] . Y chunk, slab by slab */                                                \
        B2B_G2SLAB(C, 0) B2B_G2SLAB(C, 1) B2B_G2SLAB(C, 2) B2B_G2SLAB(C, 3)                                     \
        B2B_STAMP();                                                                                            \
    }
    B2B_CHUNK(0)
    B2B_CHUNK(1)
    B2B_CHUNK(2)
    B2B_CHUNK(3)
#undef B2B_CHUNK
#undef B2B_G2SLAB
#undef B2B_STEP4
#undef B2B_STEP
    epilogue(acc2, kCB, false);
    barrier();
    B2B_STAMP();
#pragma unroll
    for (int i = 0; i < 8; i++) store_slab(i >> 1, i & 1, P.next, kCM * 2, 0);
#ifdef DAFNE_B2B_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    B2B_STAMP();
    if (tid == 0 && blockIdx.x < 64) {       // into the top halo row of image 0 of d_next (zeros otherwise)
        unsigned long long* o = (unsigned long long*)P.next + blockIdx.x * 12;
        for (int k = 0; k < nstamp; k++) o[k] = stamp[k] - stamp[0];
    }
#endif
#undef B2B_LOAD_STEP
#undef B2B_WAIT_STEP
}

}  // namespace

extern "C" {

int dafne_bottleneck_tail_head_hip(const void* d_in, const void* d_res, const void* d_wfrag, const float* d_bias3,
                                   const float* d_bias1, int n_images, int H, int W, void* d_out, void* d_next,
                                   void* stream) {
    if (!d_in || !d_res || !d_wfrag || !d_bias3 || !d_bias1 || !d_out || !d_next) return dafne::fail(DAFNE_E_INVALID, "bottleneck_tail_head: null argument");
    if (n_images < 1 || H < 1 || W < 1 || (long long)H * W > (1 << 20)) return dafne::fail(DAFNE_E_INVALID, "bottleneck_tail_head: bad size");
    B2bDev D;
    D.in = (const char*)d_in; D.res = (const char*)d_res; D.wf = (const char*)d_wfrag; D.b3 = d_bias3; D.b1 = d_bias1;
    D.out = (char*)d_out; D.next = (char*)d_next;
    D.N = n_images; D.H = H; D.W = W;
    D.tiles_per_img = (H * W + kPx - 1) / kPx;
    const long long tiles = (long long)D.tiles_per_img * n_images;
    if (tiles > (1ll << 24)) return dafne::fail(DAFNE_E_UNSUPPORTED, "bottleneck_tail_head: too many tiles");
    D.tiles = (int)tiles;
    DAFNE_MAX_LDS_ONCE(kSmemTotal, (const void*)conv_b2b_kernel);
    hipLaunchKernelGGL(conv_b2b_kernel, dim3(D.tiles), dim3(kNT), kSmemTotal, (hipStream_t)stream, D);
    return dafne::check_launch("conv_b2b");
}

}  // extern "C"
