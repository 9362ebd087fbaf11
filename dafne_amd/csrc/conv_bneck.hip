// The body of a res4 bottleneck and the head of the next one in ONE kernel (gfx950), ResNet-50/101
// [detectron2 BottleneckBlock, recalled; built by build_dafne_resnet_fpn_backbone, backbone/fpn.py:58-91]:
//
//     T = relu(conv2_b(U) + bias2)                3x3, 256 -> 256, pad 1   (U = relu(conv1_b(.)), the block's first 1x1)
//     Y = relu(conv3_b(T) + bias3 + X)            1x1, 256 -> 1024, X = the block's shortcut (identity or projection)
//     Z = relu(conv1_{b+1}(Y) + bias1)            1x1, 1024 -> 256
//
// conv_b2b.hip already keeps Y on chip between conv3 and the next conv1; here T never leaves the CU either: the 3x3's
// output tile (128 px x 256 ch bf16 = 64 KB) IS the LDS operand of conv3's GEMM.  Per block that removes a 16.8-MB write
// and read (batch 8), one launch boundary, the 3x3 kernel's epilogue and conv_b2b's exposed prologue (its T tile and first
// residual chunk were requested by all CUs at once with nothing to compute meanwhile).
//
// Structure (8 waves, one workgroup per CU, a workgroup owns a 4 x 32 pixel tile and ALL channels):
//   phase A (3x3): the (4+2) x (32+2) input patch is DMA'd into LDS for all 256 channels (four 64-channel slabs of
//     26 KB, [pixel][128 B] with the 16-byte chunk XOR (patch column >> 1) & 7: conflict-free ds_read_b128 at any tap offset) and
//     STAYS there -- the nine taps read their B fragments from it at tap-dependent pixel offsets (LDS-staged im2col), so
//     the 144 k16 steps of the layer run without a single barrier between them except one after the first slab.  The
//     weights stream L2 -> REGISTERS exactly as in conv_b2b: a wave owns 32 output channels and all 128 pixels, so every
//     1-KiB weight fragment feeds 4 MFMAs and is fetched by exactly one wave (fragment-major packing, ring of 8 steps,
//     hand-counted vmcnt).  K order = (64-channel slab, kh, kw, k16 step): the order of conv_igemm_kernel, so T is
//     bit-identical to the separate launch.  Slab 0 is awaited, slabs 1..3 trickle in one piece per wave and step behind
//     the weight loads (in-order vmcnt: a burst would stall every younger weight wait).
//   phase B: conv_b2b.hip's four chunks (GEMM1 on the resident T, in-place bias + residual + ReLU in LDS, Y chunk stored
//     and reused as the K chunk of GEMM2, next residual chunk landing slab by slab), on the same weight stream.
// Ragged tiles (H % 4 or W % 32 != 0): loads are clamped into the buffer, the rows of out-of-image pixels are STORED to a
// caller-provided dump area -- never predicated, the vmcnt bookkeeping needs an exact instruction count.
#include <type_traits>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((address_space(1))) void gvoid;
typedef __attribute__((address_space(3))) void lvoid;

constexpr int kTH = 4, kTW = 32, kPx = kTH * kTW;      // output tile
constexpr int kPC = kTW + 2, kPR = kTH + 2;            // input patch
constexpr int kPPieces = (kPR * kPC + 7) / 8;          // 26 DMA pieces of 8 px x 128 B per slab
constexpr int kPSlab = kPPieces * 1024;                // 26 624 B
constexpr int kSlab = kPx * 128;                       // one 64-channel slab of a 128-px tile: 16 KB
constexpr int kBuf = 4 * kSlab;                        // 256 channels: 64 KB
// LDS map.  The patch is dead when T and the last residual slab are written; the first three residual slabs could be
// prefetched under phase A (Y slabs 0..2 do not overlap the patch).
constexpr int kOffY = 0;                               // Y chunk / residual chunk / Z staging
constexpr int kOffT = kBuf;                            // T tile
constexpr int kOffPatch = 3 * kSlab;                   // phase A only: [4 slabs][208 px][128 B]
constexpr int kOffSide = kOffT + kBuf;                 // phase B: slab 3 of the NEXT residual chunk (16 KB; inside the dead patch)
constexpr int kOffBias = kOffPatch + 4 * kPSlab;       // fp32 [256 conv2 | 1024 conv3 | 256 conv1 | 2 x 256 spare]
constexpr int kSmemTotal = kOffBias + 8 * 1024;
static_assert(kOffPatch + 4 * kPSlab >= kOffSide + kSlab && kOffPatch <= kOffSide && kSmemTotal <= 160 * 1024, "LDS budget");
constexpr int kCM = 256, kCB = 1024, kChunks = kCB / 256;
constexpr int kNW = 8, kNT = 512;
constexpr int kPF = kPx / 32;                          // pixel fragments per wave
constexpr int kStepsA = 9 * (kCM / 16);                // 144 k16 steps of the 3x3
constexpr int kStepsB = 2 * kChunks * 16;              // 128 k16 steps of conv3 + conv1'
constexpr int kSteps = kStepsA + kStepsB;
constexpr int kRing = 8;                               // k16 steps of A fragments in flight per wave
constexpr int kWABytes = kNW * kStepsA * 1024;         // phase A weights: [8 waves][144 steps][64 lanes][8]
constexpr int kPhaseBytes = kNW * 16 * 1024;           // one phase of conv_b2b's weights: [8 waves][16 steps][64 lanes][8]
constexpr int kTrickle = 12;                           // patch slabs 1..3: 12 pieces per wave, one per step 0..11
constexpr int kRes0 = 60, kResStride = 4;              // slabs 0..2 of the first residual chunk: 6 pieces per wave at steps 60, 64, .., 80
constexpr int kDumpBytes = kPx * kCB * 2;              // 256 KB: one Y row per tile pixel

struct BneckDev {
    const char* in;      // bf16 [N, H+2, W+2, 256]   U
    const char* res;     // bf16 [N, H+2, W+2, 1024]  X
    const char* wf;      // bf16 phase A [8][144][64][8] | phase B [8 phases][8 waves][16 steps][64][8]
    const float* b2;     // [256]
    const float* b3;     // [1024]
    const float* b1;     // [256]
    char* out;           // bf16 [N, H+2, W+2, 1024]  Y
    char* next;          // bf16 [N, H+2, W+2, 256]   Z
    char* dump;          // >= kDumpBytes
    int N, H, W, tiles_x, tiles_per_img, tiles;
    unsigned max_pix;    // N * (H+2) * (W+2) - 1
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

// Vector-memory program order of a wave: bias piece (1 DMA) | patch slab 0 (4 DMA) | A(0) .. A(7) | then for every k16
// step s = 0 .. 271 (0..143 phase A; 144 + j = step j of conv_b2b's schedule):
//   [wait A(s)] MFMAs | bn_st(s) row stores | A(s + 8) if it exists | bn_post(s) DMA pieces:
//     phase A: one patch piece of slabs 1..3 at steps 0..11; slabs 0..2 of the first residual chunk (their Y-buffer space
//              does not overlap the patch) one piece at steps 60, 64, .., 80 -- HBM is idle while every CU is in phase A --,
//              its slab 3 (2 pieces, into the side buffer) after step 143, behind the barrier that retires the patch;
//     phase B: the pieces of slabs 0 + 1 of the next residual chunk after the last step of slab group 1 of GEMM2, those of slab 2
//              after group 2 (into the Y buffer; round 4: slab q behind group q, a barrier behind every group); its slab 3 goes to
//              a 16-KB SIDE buffer right after GEMM1 + epilogue of the
//              current chunk, i.e. a whole GEMM2 + GEMM1 ahead of its use (in conv_b2b it is issued last and has one GEMM1,
//              2.6 us, to arrive from HBM: chunks 1..2 took 18-25k cycles against 15k for chunk 0).
// bn_wait(j) = number of those instructions issued after A(j) and before the wait for it: vmcnt retires in order, so
// `s_waitcnt vmcnt(bn_wait(j))` is exactly "A(j) and everything older has landed".
constexpr int bn_st(int s) {
    if (s < kStepsA) return 0;
    const int i = (s - kStepsA) & 31;
    return (i >= 16 && (i & 3) == 0) ? 2 : 0;
}
constexpr int bn_post(int s) {
    if (s < kStepsA)
        return (s < kTrickle ? 1 : 0) + ((s >= kRes0 && s < kRes0 + 6 * kResStride && (s - kRes0) % kResStride == 0) ? 1 : 0) +
               (s == kStepsA - 1 ? 2 : 0);
    const int j = s - kStepsA, i = j & 31;
    if ((j >> 5) >= kChunks - 1) return 0;
    return i == 23 ? 4 : ((i == 27 || i == 15) ? 2 : 0);      // slabs 0 + 1 behind slab group 1, slab 2 behind group 2, slab 3 (side) behind GEMM1
}
constexpr int bn_wait(int j) {
    int n = 0;
    if (j < kRing) {
        n += kRing - 1 - j;                                    // A(j+1 .. 7)
        for (int s = 0; s < j; s++) n += bn_st(s) + 1 + bn_post(s);
    } else {
        n += bn_post(j - kRing);                               // the DMA pieces right behind A(j) at the end of step j - 8
        for (int s = j - kRing + 1; s < j; s++) n += bn_st(s) + (s + kRing < kSteps ? 1 : 0) + bn_post(s);
    }
    return n;
}
// spot checks (hand-counted): steady state 7; the trickle adds one per step; phase B around the side-buffer and slab pieces
static_assert(bn_wait(0) == 7 && bn_wait(1) == 8 && bn_wait(7) == 14 && bn_wait(8) == 15 && bn_wait(12) == 15 && bn_wait(13) == 14 &&
              bn_wait(20) == 7 && bn_wait(61) == 8 && bn_wait(68) == 9 && bn_wait(100) == 7 && bn_wait(kStepsA + 7) == 9 &&
              bn_wait(kStepsA + 8) == 7 && bn_wait(kStepsA + 16) == 9 && bn_wait(kStepsA + 23) == 13 && bn_wait(kStepsA + 24) == 13 &&
              bn_wait(kStepsA + 28) == 15 && bn_wait(kStepsA + 31) == 17 && bn_wait(kStepsA + 39) == 7 && bn_wait(kStepsA + 127) == 4,
              "vmcnt bookkeeping");

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// A fragment of k16 step J: one coalesced 1-KiB load per wave, L2 -> registers.  Inline asm: the compiler would sink visible
// loads to their uses and wait for each; readiness is tracked by hand (bn_wait).
template <int J>
__device__ __forceinline__ void bn_load(bf16x8 (&ar)[kRing], const char* wf, unsigned voffA, unsigned voffB) {
    if constexpr (J < kStepsA) {
        const char* sb = wf + (size_t)J * 1024;
        // "+v": ONE register per ring slot from the zero-initialisation on (an output-only operand lets the compiler move the
        // value between registers -- e.g. inside the plain-C++ epilogues -- while the load is in flight: conv.hip rp_load)
        asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(ar[J % kRing]) : "v"(voffA), "s"(sb) : "memory");
    } else if constexpr (J < kSteps) {
        constexpr int jb = J - kStepsA;
        const char* sb = wf + (size_t)kWABytes + (size_t)(jb >> 4) * kPhaseBytes + (jb & 15) * 1024;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(ar[J % kRing]) : "v"(voffB), "s"(sb) : "memory");
    }
}
// the four B fragments of phase-A step J (patch rows kh .. kh + 3 at tap column kw, chunk kc of slab sl): inline asm,
// completion is awaited by the caller (lgkmcnt)
template <int J>
__device__ __forceinline__ void bn_bread(bf16x8 (&b)[kPF], const unsigned (&pb)[3], unsigned lds_base) {
    constexpr int sl = J / 36, t = J % 36, kh = t / 12, kw = (t >> 2) % 3, kc = t & 3;
    const unsigned ad = lds_base + ((pb[kw] ^ (unsigned)(kc << 5)) + (unsigned)(sl * kPSlab));
    constexpr int o0 = (kh + 0) * kPC * 128, o1 = (kh + 1) * kPC * 128, o2 = (kh + 2) * kPC * 128, o3 = (kh + 3) * kPC * 128;
    asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"
                 : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3])
                 : "v"(ad), "n"(o0), "n"(o1), "n"(o2), "n"(o3)
                 : "memory");
}

// phase B: the four B fragments (pixel fragments 4096 B apart) of k16 step ST of slab Q of the buffer at byte offset BUF
template <int BUF, int Q, int ST>
__device__ __forceinline__ void bn_bread_b(bf16x8 (&b)[kPF], const unsigned (&bs)[4], unsigned lds_base) {
    const unsigned ad = lds_base + (unsigned)BUF + bs[ST];
    constexpr int o = Q * kSlab;
    asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"
                 : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3])
                 : "v"(ad), "n"(o), "n"(o + 4096), "n"(o + 8192), "n"(o + 12288)
                 : "memory");
}

template <int J>
__device__ __forceinline__ void bn_wait_for(bf16x8 (&ar)[kRing]) {
    constexpr int kWaitN = bn_wait(J);
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(ar[J % kRing]) : "n"(kWaitN) : "memory");
}

// HEAD = false (the stage's last block: no next conv1): GEMM2, its epilogue and the Z rows are skipped; the weight stream still
// walks the (zero) conv1' fragments -- the vmcnt bookkeeping is one schedule for both forms.
template <bool HEAD>
__global__ void __launch_bounds__(512, 2) conv_bneck_kernel(BneckDev P) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, half = lane >> 5;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;

    const int tile = xcd_remap(blockIdx.x, P.tiles);
    const int img = tile / P.tiles_per_img;
    const int tt = tile - img * P.tiles_per_img;
    const int ty = tt / P.tiles_x, tx = tt - ty * P.tiles_x;
    const int row0 = ty * kTH, col0 = tx * kTW;
    const int Wp = P.W + 2;

    // tile pixel px = r * 32 + c  <->  image pixel (row0 + r, col0 + c)
    auto pix_index = [&](int px) {           // haloed pixel index, clamped into the image (loads)
        int r = row0 + (px >> 5), c = col0 + (px & 31);
        r = r < P.H ? r : P.H - 1;
        c = c < P.W ? c : P.W - 1;
        return (unsigned)((img * (P.H + 2) + r + 1) * Wp + c + 1);
    };
    auto pix_valid = [&](int px) { return row0 + (px >> 5) < P.H && col0 + (px & 31) < P.W; };

    // ---- patch DMA map: piece pc = 8 consecutive patch pixels (patch pixel pp = p * 34 + q <-> input pixel
    // (row0 - 1 + p, col0 - 1 + q)); wave w moves pieces w, w + 8, w + 16 and min(w + 24, 25) of every slab (the six waves
    // without a fourth piece re-load piece 25: every wave issues the same number of DMAs)
    size_t pofs[4];
    unsigned pdst[4];
#pragma unroll
    for (int ii = 0; ii < 4; ii++) {
        int pc = wave + kNW * ii;
        pc = pc < kPPieces ? pc : kPPieces - 1;
        const int pp = pc * 8 + (lane >> 3);
        const int p = pp / kPC, q = pp - p * kPC;
        unsigned g = (unsigned)((img * (P.H + 2) + row0 + p) * Wp + col0 + q);
        g = g < P.max_pix ? g : P.max_pix;                       // ragged tiles reach past the image (and the buffer)
        pofs[ii] = (size_t)g * (kCM * 2) + (unsigned)(((lane & 7) ^ ((q >> 1) & 7)) * 16);      // swizzle by patch COLUMN
        pdst[ii] = (unsigned)(kOffPatch + pc * 1024);
    }
    auto patch_piece = [&](int sl, int ii) {
        __builtin_amdgcn_global_load_lds((gvoid*)(P.in + pofs[ii] + sl * 128), (lvoid*)(lds + pdst[ii] + sl * kPSlab), 16, 0, 0);
    };

    // ---- residual DMA map (conv_b2b.hip): a slab is 16 pieces of 8 px x 128 B; wave w moves pieces w and w + 8
    unsigned dpix[2], dq[2];
#pragma unroll
    for (int ii = 0; ii < 2; ii++) {
        const int px = (wave + kNW * ii) * 8 + (lane >> 3);
        dpix[ii] = pix_index(px);
        dq[ii] = (unsigned)(((lane & 7) ^ ((px >> 1) & 7)) * 16);
    }
    auto dma_piece = [&](unsigned col0b, int sl, int ii) {       // slabs 0..2 land in the Y buffer, slab 3 in the side buffer
        __builtin_amdgcn_global_load_lds((gvoid*)(P.res + (size_t)dpix[ii] * (kCB * 2) + col0b + sl * 128 + dq[ii]),
                                         (lvoid*)(lds + (sl == 3 ? kOffSide : kOffY + sl * kSlab) + (wave + kNW * ii) * 1024), 16, 0, 2);      // aux 2 = nt: X is read once per block (the Y rows, not-nt, are the next block's X)
    };
    auto dma_slab = [&](unsigned col0b, int sl) {                // 2 pieces per wave into slab sl of the Y buffer
        dma_piece(col0b, sl, 0);
        dma_piece(col0b, sl, 1);
    };

    // ---- B fragment offsets
    // phase B: k16 step s inside a slab, pixel fragment 0 of a [128 px][128 B] slab
    unsigned bs[4];
#pragma unroll
    for (int s = 0; s < 4; s++) bs[s] = (unsigned)(frow * 128 + (((2 * s + half) ^ ((frow >> 1) & 7)) * 16));
    // phase A: patch row p (0..5) at tap column kw (0..2): pixel p * 34 + q, q = kw + frow; k16 step kc reads the 16-byte
    // chunk (2 kc + half) ^ sw, sw = (q >> 1) & 7 (the row pitch is even: a swizzle by COLUMN is conflict-free like one by
    // pixel index, and independent of the patch row).  That is  (pb[kw] ^ (kc << 5)) + p * 34 * 128  with
    // pb[kw] = q * 128 | ((half ^ (sw & 1)) << 4) | ((sw >> 1) << 5)   (bits 5..6 of q * 128 are zero)
    unsigned pb[3];
#pragma unroll
    for (int kw = 0; kw < 3; kw++) {
        const int q = kw + frow;
        const int sw = (q >> 1) & 7;
        pb[kw] = (unsigned)(kOffPatch + q * 128 + ((half ^ (sw & 1)) << 4) + ((sw >> 1) << 5));
    }

    // ---- A operand: L2 -> registers through inline asm, ring of 8 k16 steps (conv_b2b.hip)
    const unsigned voffA = (unsigned)(wave * kStepsA * 1024 + lane * 16);
    const unsigned voffB = (unsigned)(wave * 16 * 1024 + lane * 16);
    bf16x8 ar[kRing];
#pragma unroll
    for (int k = 0; k < kRing; k++) ar[k] = bf16x8{};
    auto load_step = [&](auto J) { bn_load<decltype(J)::value>(ar, P.wf, voffA, voffB); };
    auto wait_step = [&](auto J) { bn_wait_for<decltype(J)::value>(ar); };

    f32x16 acc1[kPF], acc2[kPF];

    auto barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // the Y buffer (256 channels of 128 px) -> global rows, one slab (64 channels) of 64 pixels per pass: 8 consecutive
    // threads write one pixel's 128 B.  EXACTLY 8 stores per lane and chunk (vmcnt bookkeeping): out-of-image pixels of a
    // ragged tile go to the dump area.
    auto store_slab = [&](int sl, int h, char* dst, unsigned pix_bytes, unsigned col0b) {
        int idx = tid;
        asm volatile("" : "+v"(idx));                // recompute the row address at every pass: holding 8 of them spills
        const int px = h * 64 + (idx >> 3);
        const int q = idx & 7;
        const u32x4 v = *(const u32x4*)(lds + kOffY + sl * kSlab + px * 128 + ((q ^ ((px >> 1) & 7)) * 16));
        char* a = dst + (size_t)pix_index(px) * pix_bytes;
        char* d = P.dump + (size_t)px * (kCB * 2);
        a = pix_valid(px) ? a : d;
        *(u32x4*)(a + col0b + sl * 128 + q * 16) = v;
    };
    // round 5: the Y row stores with everything per-lane precomputed -- a lane stores the same two tile pixels (px0 = tid >> 3 and
    // px0 + 64: same swizzle, LDS address + 8192) in every pass, so the two row addresses are computed ONCE (4 VGPRs); slab and
    // chunk are immediates of the LDS read / the store.  The generic form above costs ~30 vector instructions per pass (clamps,
    // 64-bit multiplies, the dump select): 8 passes per chunk made GEMM2 take 425 cycles per step against 285 in GEMM1.
    char* yrow[2];
    {
        const int px0 = tid >> 3, q = tid & 7;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int px = px0 + 64 * h;
            char* a = P.out + (size_t)pix_index(px) * (kCB * 2);
            char* d = P.dump + (size_t)px * (kCB * 2);
            yrow[h] = (pix_valid(px) ? a : d) + q * 16;
        }
    }
    const unsigned ylds = lds_base + (unsigned)(kOffY + (tid >> 3) * 128 + (((tid & 7) ^ (((tid >> 3) >> 1) & 7)) * 16));
    auto store_y = [&](auto SL, auto H, auto C) {
        constexpr int sl = decltype(SL)::value, h = decltype(H)::value, c = decltype(C)::value;
        const unsigned la = ylds;                // (non-dependent uses: a generic lambda captures the two only through them)
        char* const* yr = yrow;
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(la), "n"(sl * kSlab + h * 8192) : "memory");
        asm volatile("global_store_dwordx4 %0, %1, off offset:%2" :: "v"(yr[h]), "v"(v), "n"(c * 512 + sl * 128) : "memory");
    };
    // acc + bias (+ residual already in the buffer) -> ReLU -> bf16, in place in the buffer at byte offset buf
    // (conv_b2b.hip's epilogue; the expressions are those of the separate kernels)
    const unsigned lbias_off = lds_base + (unsigned)kOffBias;      // fp32 [256 conv2 | 1024 conv3 | 256 conv1]
    auto epilogue = [&](f32x16* acc, int bias0, bool with_res, int buf) {
        typedef __attribute__((ext_vector_type(4))) float f32x4;
        typedef __attribute__((ext_vector_type(2))) float f32x2;
        const unsigned rmask = with_res ? 0xffffffffu : 0u;
        const unsigned ebase = lds_base + (unsigned)(buf + (wave >> 1) * kSlab + frow * 128 + 8 * half);
        // the residual of slab 3 (waves 6, 7) waits in the side buffer
        const unsigned rbase = (with_res && (wave >> 1) == 3) ? lds_base + (unsigned)(kOffSide + frow * 128 + 8 * half) : ebase;
#pragma unroll
        for (int gp = 0; gp < 2; gp++) {                 // two 8-channel groups at a time (register budget)
            f32x4 bv[2];
            u32x2 rc[2][kPF];
            unsigned ead[2], rad[2];
            // LDS reads first, one wait.  Inline asm: a plain LDS read here makes the compiler drain
            // vmcnt (it cannot tell the read from the residual DMA's destination).  Pixel fragments sit 4096 B apart.
#pragma unroll
            for (int gg = 0; gg < 2; gg++) {
                const int g = 2 * gp + gg;
                ead[gg] = ebase + (unsigned)(((((wave & 1) * 4 + g) ^ ((frow >> 1) & 7))) * 16);
                rad[gg] = rbase + (unsigned)(((((wave & 1) * 4 + g) ^ ((frow >> 1) & 7))) * 16);
                const unsigned bad = lbias_off + (unsigned)((bias0 + wave * 32 + 8 * g + 4 * half) * 4);
                asm volatile("ds_read_b128 %4, %6\n\tds_read_b64 %0, %5\n\tds_read_b64 %1, %5 offset:4096\n\t"
                             "ds_read_b64 %2, %5 offset:8192\n\tds_read_b64 %3, %5 offset:12288"
                             : "=&v"(rc[gg][0]), "=&v"(rc[gg][1]), "=&v"(rc[gg][2]), "=&v"(rc[gg][3]), "=&v"(bv[gg])
                             : "v"(rad[gg]), "v"(bad)
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

#ifdef DAFNE_BNECK_TIMING
    unsigned long long stamp[12];
    int nstamp = 0;
#define BN_STAMP() stamp[nstamp++] = __builtin_amdgcn_s_memtime()
#else
#define BN_STAMP()
#endif
    BN_STAMP();
    // ---- prologue: biases -> LDS by DMA, one 1-KiB piece per wave (wave 0: conv2, 1..4: conv3, 5: conv1, 6..7 re-load
    // conv1 into spare LDS); the oldest vector-memory operation of every wave, so every later wait covers it
    {
        const float* bsrc = wave == 0 ? P.b2 : (wave < 5 ? P.b3 + (wave - 1) * 256 : P.b1);
        __builtin_amdgcn_global_load_lds((gvoid*)(bsrc + lane * 4), (lvoid*)(lds + kOffBias + wave * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int ii = 0; ii < 4; ii++) patch_piece(0, ii);
    static_for<0, kRing>(load_step);

    // ================================================================ phase A: T = relu(conv2(U) + bias2)
#pragma unroll
    for (int b = 0; b < kPF; b++)
#pragma unroll
        for (int k = 0; k < 16; k++) acc1[b][k] = 0.f;
    bf16x8 bfr[2][kPF];
    static_for<0, kStepsA>([&](auto J) {
        constexpr int j = decltype(J)::value;
        wait_step(J);
        if constexpr (j == 0) barrier();             // slab 0 and the biases of every wave have landed
        if constexpr (j == 35) barrier();            // slabs 1..3: every wave passed a wait covering its last piece at step 20
        // the B fragments of step j + 1 are requested before the MFMAs of step j (two register sets; the counted lgkmcnt leaves
        // exactly those four reads in flight)
        if constexpr (j == 0) bn_bread<0>(bfr[0], pb, lds_base);
        if constexpr (j + 1 < kStepsA) {
            bn_bread<j + 1>(bfr[(j + 1) & 1], pb, lds_base);
            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(bfr[j & 1][0]), "+v"(bfr[j & 1][1]), "+v"(bfr[j & 1][2]), "+v"(bfr[j & 1][3]) :: "memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bfr[j & 1][0]), "+v"(bfr[j & 1][1]), "+v"(bfr[j & 1][2]), "+v"(bfr[j & 1][3]) :: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < kPF; r++) acc1[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[j % kRing], bfr[j & 1][r], acc1[r], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        load_step(std::integral_constant<int, j + kRing>{});
        if constexpr (j < kTrickle) patch_piece(1 + j / 4, j & 3);
        if constexpr (j >= kRes0 && j < kRes0 + 6 * kResStride && (j - kRes0) % kResStride == 0)
            dma_piece(0u, ((j - kRes0) / kResStride) >> 1, ((j - kRes0) / kResStride) & 1);      // R(0), slabs 0..2
    });
    BN_STAMP();
    barrier();                                       // every wave is done with the patch: T and the residual may land on it
    dma_slab(0u, 3);                                 // R(0), slab 3: awaited through the weight waits of GEMM1(0)
    epilogue(acc1, 0, false, kOffT);
    barrier();
    BN_STAMP();

    // ================================================================ phase B: conv_b2b.hip's schedule
#pragma unroll
    for (int b = 0; b < kPF; b++)
#pragma unroll
        for (int k = 0; k < 16; k++) acc2[b][k] = 0.f;
    static_for<0, kChunks>([&](auto C) {
        constexpr int c = decltype(C)::value;
        constexpr int j0 = kStepsA + 32 * c;
        // GEMM1: Y chunk c = W3[c] . T  (K = 256 over the four slabs of the T tile)
#pragma unroll
        for (int b = 0; b < kPF; b++)
#pragma unroll
            for (int k = 0; k < 16; k++) acc1[b][k] = 0.f;
        static_for<0, 16>([&](auto I) {
            constexpr int i = decltype(I)::value;
            constexpr int j = j0 + i;
            wait_step(std::integral_constant<int, j>{});
            if constexpr (i == 0) bn_bread_b<kOffT, 0, 0>(bfr[j & 1], bs, lds_base);
            if constexpr (i + 1 < 16) {
                bn_bread_b<kOffT, ((i + 1) >> 2), ((i + 1) & 3)>(bfr[(j + 1) & 1], bs, lds_base);
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(bfr[j & 1][0]), "+v"(bfr[j & 1][1]), "+v"(bfr[j & 1][2]), "+v"(bfr[j & 1][3]) :: "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bfr[j & 1][0]), "+v"(bfr[j & 1][1]), "+v"(bfr[j & 1][2]), "+v"(bfr[j & 1][3]) :: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < kPF; b++) acc1[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[j % kRing], bfr[j & 1][b], acc1[b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_step(std::integral_constant<int, j + kRing>{});
        });
        barrier();                                   // every wave's residual pieces are in the Y buffer (older than A(j0+15))
        if constexpr (c == 0) BN_STAMP();
        epilogue(acc1, 256 + c * 256, true, kOffY);
        barrier();
        if constexpr (c + 1 < kChunks) dma_slab((unsigned)(c + 1) * 512u, 3);       // side buffer: free since the epilogue read it
        if constexpr (c == 0) BN_STAMP();
        // GEMM2: Z += W1[:, chunk c] . Y chunk, slab by slab; then (every wave done with the slab, its row stores have
        // read it) the same slab of the NEXT residual chunk starts to land in its place
        static_for<0, 4>([&](auto Q) {
            constexpr int q = decltype(Q)::value;
            static_for<0, 4>([&](auto I) {
                constexpr int i = decltype(I)::value;
                constexpr int j = j0 + 16 + 4 * q + i;
                wait_step(std::integral_constant<int, j>{});
                if constexpr (i == 0) bn_bread_b<kOffY, q, 0>(bfr[j & 1], bs, lds_base);
                if constexpr (i + 1 < 4) {
                    bn_bread_b<kOffY, q, i + 1>(bfr[(j + 1) & 1], bs, lds_base);
                    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(bfr[j & 1][0]), "+v"(bfr[j & 1][1]), "+v"(bfr[j & 1][2]), "+v"(bfr[j & 1][3]) :: "memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bfr[j & 1][0]), "+v"(bfr[j & 1][1]), "+v"(bfr[j & 1][2]), "+v"(bfr[j & 1][3]) :: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (HEAD) {
#pragma unroll
                    for (int b = 0; b < kPF; b++) acc2[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[j % kRing], bfr[j & 1][b], acc2[b], 0, 0, 0);
                }
                if constexpr (i == 0) {
                    store_y(Q, std::integral_constant<int, 0>{}, C);
                    store_y(Q, std::integral_constant<int, 1>{}, C);
                }
                __builtin_amdgcn_sched_barrier(0);
                load_step(std::integral_constant<int, j + kRing>{});
            });
            // round 5: a barrier only where a residual slab lands behind it -- behind slab group 1 (slabs 0 and 1 of the next chunk,
            // 4 pieces) and behind group 2 (slab 2); none behind groups 0 and 3 and none in the last chunk (16 -> 6 barriers per
            // block in GEMM2).  The next epilogue's in-place writes are ordered behind every wave's GEMM2 reads by the barrier
            // that follows GEMM1 of the next chunk / precedes the Z epilogue.
            if constexpr (c + 1 < kChunks && q == 1) {
                barrier();
                dma_slab((unsigned)(c + 1) * 512u, 0);
                dma_slab((unsigned)(c + 1) * 512u, 1);
            }
            if constexpr (c + 1 < kChunks && q == 2) {
                barrier();
                dma_slab((unsigned)(c + 1) * 512u, 2);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        BN_STAMP();
    });
    if constexpr (HEAD) {
        barrier();                                   // every wave is done reading the last Y chunk (no barrier behind slab group 3 any more)
        epilogue(acc2, 256 + kCB, false, kOffY);
        barrier();
        BN_STAMP();
#pragma unroll
        for (int i = 0; i < 8; i++) store_slab(i >> 1, i & 1, P.next, kCM * 2, 0);
    }
#ifdef DAFNE_BNECK_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BN_STAMP();
    if (tid == 0 && blockIdx.x < 64) {       // into the dump area
        unsigned long long* o = (unsigned long long*)P.dump + blockIdx.x * 12;
        for (int k = 0; k < nstamp; k++) o[k] = stamp[k] - stamp[0];
    }
#endif
}

}  // namespace

extern "C" {

size_t dafne_bottleneck_body_scratch_bytes(void) { return (size_t)kDumpBytes; }

int dafne_bottleneck_body_hip(const void* d_in, const void* d_res, const void* d_wfrag, const float* d_bias2,
                              const float* d_bias3, const float* d_bias1, int n_images, int H, int W, void* d_out,
                              void* d_next, void* d_scratch, size_t scratch_bytes, void* stream) {
    const bool head = d_next != nullptr;
    if (!d_in || !d_res || !d_wfrag || !d_bias2 || !d_bias3 || (head && !d_bias1) || !d_out || !d_scratch)
        return dafne::fail(DAFNE_E_INVALID, "bottleneck_body: null argument");
    if (n_images < 1 || H < 1 || W < 1 || (long long)H * W > (1 << 20)) return dafne::fail(DAFNE_E_INVALID, "bottleneck_body: bad size");
    if (scratch_bytes < (size_t)kDumpBytes) return dafne::fail(DAFNE_E_WORKSPACE, "bottleneck_body: scratch %zu < %d", scratch_bytes, kDumpBytes);
    BneckDev D;
    D.in = (const char*)d_in; D.res = (const char*)d_res; D.wf = (const char*)d_wfrag;
    D.b2 = d_bias2; D.b3 = d_bias3; D.b1 = head ? d_bias1 : d_bias2;      // (a valid 1-KB source for the bias DMA; unused without the head)
    D.out = (char*)d_out; D.next = (char*)d_next; D.dump = (char*)d_scratch;
    D.N = n_images; D.H = H; D.W = W;
    D.tiles_x = (W + kTW - 1) / kTW;
    D.tiles_per_img = D.tiles_x * ((H + kTH - 1) / kTH);
    const long long tiles = (long long)D.tiles_per_img * n_images;
    const long long pix = (long long)n_images * (H + 2) * (W + 2);
    if (tiles > (1ll << 24) || pix * (kCB * 2) > 0xffffffffll) return dafne::fail(DAFNE_E_UNSUPPORTED, "bottleneck_body: too large");
    D.tiles = (int)tiles;
    D.max_pix = (unsigned)(pix - 1);
    DAFNE_MAX_LDS_ONCE(kSmemTotal, (const void*)conv_bneck_kernel<true>, (const void*)conv_bneck_kernel<false>);
    if (head) hipLaunchKernelGGL(conv_bneck_kernel<true>, dim3(D.tiles), dim3(kNT), kSmemTotal, (hipStream_t)stream, D);
    else hipLaunchKernelGGL(conv_bneck_kernel<false>, dim3(D.tiles), dim3(kNT), kSmemTotal, (hipStream_t)stream, D);
    return dafne::check_launch("conv_bneck");
}

}  // extern "C"
