// Implicit-GEMM convolution for gfx950: NHWC bf16 activations with a 1-pixel zero
// halo, bf16 weights [Cout][Cin/64][KH][KW][64], fp32 accumulation on the matrix cores.
//
// Replaces every torch conv2d the reference's inference path issues through cuDNN:
// detectron2 ResNet bottlenecks / FPN laterals and outputs (dafne/modeling/backbone/
// fpn.py:58-91 [d2 ResNet/FPN recalled, SURVEY appendix B]), LastLevelP6P7
// (fpn.py:16-37), and the DAFNeHead tower / prediction convs
// (dafne/modeling/dafne/dafne.py:209-229,318-344).  Fused epilogues: folded-FrozenBN
// bias, residual add, nearest-2x top-down add (FPN), ReLU, per-tile GroupNorm partial
// sums, fp32 prediction outputs.
//
// GEMM view: D[cout][pixel] = sum_k W[cout][k] * X[pixel][k], k = (cin/64, kh, kw, cin%64).
// The halo makes every tap of a 3x3 a plain in-bounds 128-byte row segment, so the
// pixel operand is staged exactly like the weight operand: `global_load_lds` 16-byte
// pieces, 8 lanes per 64-channel row slab, XOR-swizzled on the SOURCE address
// (LDS image stays lane-linear) so the MFMA fragment reads (`ds_read_b128`, rows =
// lane&31) are bank-conflict free.  v_mfma_f32_32x32x16_bf16, weights as the A
// operand: each lane ends up with 4 consecutive output channels of one pixel per
// accumulator quad -> 8-byte NHWC stores.
//
// Pipeline: 2 LDS stages, one barrier per 64-deep K step; the next step's loads are
// issued right after the barrier and land under the current step's 16 MFMAs/wave.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(1))) const void gvoid;
typedef __attribute__((address_space(3))) void lvoid;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int kMaxSegs = 5;
constexpr int kBK = 64;           // K step (bf16 elements) = one 128-byte row slab
constexpr int kRowBytes = kBK * 2;

struct SegDev {
    const char* in;
    char* out;
    const char* res;
    int Hin, Win, Hout, Wout;
    int tiles_per_img;   // ceil(Hout*Wout / BM); patch kernel: tiles_x * ceil(Hout/8)
    int tile0;           // first M tile of this segment
    int tiles_x;         // patch kernel: ceil(Wout/32)
};

struct ConvDev {
    SegDev seg[kMaxSegs];
    int n_segs, N;
    int Cin, Cout, Cout_pad, KH, KW, stride, pad;
    unsigned flags;
    const char* w;
    const float* bias;
    float* gn_partial;
    int ksteps;        // KH*KW*Cin/64 (stem: 4)
    int kbytes;        // bytes per weight row
    int mtiles, ntiles;
    int bn, bm;        // chosen tile
    int stem;          // 7x7 s2 stem on the 4-channel padded image
    int patch;         // 3x3 patch kernel (2-D tiles)
    int slab;          // 3x3 narrow-Cout fp32 kernel (2-D tiles, whole-slab operands)
    int rp;            // 3x3 resident-patch kernel (4 x 32 tiles, Cin = 256, fragment-major weights)
    const float* in_stats;   // GN_INPUT: [n_segs][N][Cin/8][2] mean, rstd of the input
    const float* in_gamma;   //           [Cin]
    const float* in_beta;    //           [Cin]
    const float* oscale;     // fp8 path: [Cout] output scale (weight scale / in_qscale)
    float in_qscale;         // fp8 path: activations are multiplied by this before the e4m3 rounding
    float* gn_stats_out;     // GN_FINALIZE: [n_segs][N][Cout/8][2] mean, rstd (written by the last tile of every image)
    int* gn_counters;        // GN_FINALIZE: [n_segs][N] arrival tickets, zero between launches
    float gn_eps;
};

__device__ __forceinline__ unsigned short f2bf(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);   // round to nearest even (inputs are finite)
    return (unsigned short)(u >> 16);
}

// two fp32 -> packed bf16x2 with the hardware converter (v_cvt_pk_bf16_f32: round to nearest even,
// the same rounding as f2bf for finite inputs); one instruction instead of eight
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(3))) float f32x3;
typedef __attribute__((ext_vector_type(2))) short i16x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    const f32x2 v = {lo, hi};
    const bf16x2_t w = __builtin_convertvector(v, bf16x2_t);
    return __builtin_bit_cast(unsigned, w);
}

__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// Bijective XCD-aware remap: consecutive logical tiles (which share the pixel
// operand across N tiles and the halo rows across M tiles) land on one XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}

// m / W without an integer division (a ~40-instruction sequence on this ISA), for 0 <= m < H*W <= 2^20
// (checked on the host): (m + 0.5) / W is at least 0.5/W away from an integer, the fp32 product is off by
// at most H * 2^-23, so truncation is exact while H*W < 2^22.
__device__ __forceinline__ void divmod_small(int m, int W, float invW, int& q, int& r) {
    q = (int)(((float)m + 0.5f) * invW);
    r = m - q * W;
}

// NS: operand stages in LDS of the 4-wave loop.  2 = double buffer with a drain per K step (two workgroups per CU hide each
// other's load latency when the launch has the tiles for it); 4 = ring with counted waits for launches with at most one
// tile per CU (res5, the top FPN levels, every small layer of a sub-batch plan), whose K loop was one L2 round trip per
// step: res5 conv2 (K = 4608, 72 steps) 0.9 -> 0.3 us per step.  Same K order, same results.
template <int WC, int WP, int TC, int TP, int NS = 2>
__global__ void __launch_bounds__(WC * WP * 64) conv_igemm_kernel(ConvDev P) {
    constexpr int NW = WC * WP;            // waves per block
    constexpr int NT = NW * 64;
    constexpr int BN = WC * TC * 32;       // output channels per block
    constexpr int BM = WP * TP * 32;       // output pixels per block
    constexpr int ROWS = BN + BM;
    constexpr int NL = ROWS / (8 * NW);    // 1-KiB load instructions per wave per stage
    constexpr int STAGE = ROWS * kRowBytes;
    // epilogue staging: fp32 [BM][EN] (+16 B row pad), EN channels per pass
    constexpr int EN = BN > 128 ? 128 : BN;
    constexpr int EPASS = BN / EN;
    constexpr int ROWF = EN * 4 + 16;
    constexpr int CH = EN / 8;             // 16-byte bf16 output chunks per pixel row
    static_assert(ROWS % (8 * NW) == 0 && NT % CH == 0 && BM % (NT / CH) == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave / WP, wp = wave % WP;
#ifdef DAFNE_CONV_TIMING
    unsigned long long ts[6];
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz, chip-wide
    ts[0] = __builtin_readcyclecounter();
#define TSTAMP(i) ts[i] = __builtin_readcyclecounter()
#else
#define TSTAMP(i)
#endif

    const int bid = xcd_remap(blockIdx.x, P.mtiles * P.ntiles);
    const int nt = bid % P.ntiles;
    const int mt = bid / P.ntiles;
    int si = 0;
#pragma unroll
    for (int k = 1; k < kMaxSegs; k++)
        if (k < P.n_segs && mt >= P.seg[k].tile0) si = k;
    const SegDev& S = P.seg[si];
    const int img = (mt - S.tile0) / S.tiles_per_img;
    const int m0 = ((mt - S.tile0) % S.tiles_per_img) * BM;
    const int HW = S.Hout * S.Wout;
    const float invW = 1.0f / (float)S.Wout;
    const int Wp = P.stem ? S.Win : S.Win + 2;   // input row pitch in pixels (stem: already padded)
    const int Hp = P.stem ? S.Hin : S.Hin + 2;
    const int cpx = P.stem ? 8 : P.Cin * 2;      // bytes per input pixel

    // ---- per-lane source offsets of the rows this lane stages -----------------
    unsigned gofs[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) {
        const int r = (j * NW + wave) * 8 + (lane >> 3);
        const int q = (lane & 7) ^ ((r >> 1) & 7);     // logical 16-byte chunk this lane fetches
        if ((j * NW + wave) * 8 < BN) {
            gofs[j] = (unsigned)(nt * BN + r) * (unsigned)P.kbytes + (unsigned)q * 16u;
        } else {
            int pix = m0 + (r - BN);
            pix = pix < HW ? pix : HW - 1;             // ragged last tile: stay in bounds
            int ho, wo;
            divmod_small(pix, S.Wout, invW, ho, wo);
            const unsigned row = (unsigned)(img * Hp + ho * P.stride + (P.stem ? 0 : 1 - P.pad));
            const unsigned col = (unsigned)(wo * P.stride + (P.stem ? 0 : 1 - P.pad));
            const unsigned lanepart = P.stem ? (unsigned)(q >> 2) * (unsigned)(Wp * 8) + (unsigned)(q & 3) * 16u
                                             : (unsigned)q * 16u;
            gofs[j] = (row * (unsigned)Wp + col) * (unsigned)cpx + lanepart;
        }
    }

    // K-step bookkeeping (scalar): tap (kh,kw) and channel slab c0
    int kh = 0, kw = 0, c0 = 0;
    auto issue = [&](int stage, int step) {
        const unsigned koffW = (unsigned)step * (unsigned)kRowBytes;
        const unsigned koffX = P.stem ? (unsigned)(2 * step) * (unsigned)(Wp * 8)
                                      : (unsigned)((kh * Wp + kw) * P.Cin + c0) * 2u;
#pragma unroll
        for (int j = 0; j < NL; j++) {
            const bool isW = (j * NW + wave) * 8 < BN;
            const char* base = isW ? P.w : S.in;
            const unsigned off = gofs[j] + (isW ? koffW : koffX);
            char* dst = lds + stage * STAGE + (j * NW + wave) * 8 * kRowBytes;
            __builtin_amdgcn_global_load_lds((gvoid*)(base + off), (lvoid*)dst, 16, 0, 0);
        }
        // K order = (64-channel slab, kh, kw): the 9 taps of one slab touch the same input
        // rows back to back, so 8 of the 9 re-reads hit L2/L1 instead of MALL/HBM
        if (++kw == P.KW) {
            kw = 0;
            if (++kh == P.KH) { kh = 0; c0 += kBK; }
        }
    };

    // Residual / top-down rows of single-pass tiles are fetched NOW (16 B per lane, coalesced):
    // the HBM read stream then runs under the K loop instead of colliding with the output
    // write burst of the epilogue.
    constexpr int PPT0 = BM / (NT / CH);
    constexpr bool kEarlyRes = (EPASS == 1);
    uint4 rr0[kEarlyRes ? PPT0 : 1];
    if (kEarlyRes && (P.flags & (DAFNE_CONV_RESIDUAL | DAFNE_CONV_UPSAMPLE_ADD))) {
        const bool up = P.flags & DAFNE_CONV_UPSAMPLE_ADD;
        const int cob = nt * BN + (tid % CH) * 8;
#pragma unroll
        for (int i = 0; i < PPT0; i++) {
            const int p = tid / CH + i * (NT / CH);
            int m = m0 + p;
            m = m < HW ? m : HW - 1;
            int ho, wo;
            divmod_small(m, S.Wout, invW, ho, wo);
            const size_t rpix = up
                ? ((size_t)(img * (S.Hout / 2 + 2) + ho / 2 + 1) * (S.Wout / 2 + 2) + wo / 2 + 1)
                : ((size_t)(img * (S.Hout + 2) + ho + 1) * (S.Wout + 2) + wo + 1);
            rr0[kEarlyRes ? i : 0] = *(const uint4*)(S.res + (rpix * P.Cout + cob) * 2);
        }
    }

    f32x16 acc[TC][TP];
#pragma unroll
    for (int a = 0; a < TC; a++)
#pragma unroll
        for (int b = 0; b < TP; b++)
#pragma unroll
            for (int k = 0; k < 16; k++) acc[a][b][k] = 0.f;

    // fragment read offsets: row = lane&31 inside a 32-row tile, chunk (2ks + lane>>5) ^ f
    const int frow = lane & 31;
    const int fsw = (frow >> 1) & 7;
    unsigned roff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) roff[ks] = (unsigned)frow * kRowBytes + (unsigned)(((2 * ks + (lane >> 5)) ^ fsw) * 16);
    const int arow0 = wc * TC * 32;
    const int brow0 = BN + wp * TP * 32;

    if constexpr (NW == 8) {
        // ---- streaming two-group schedule (8 waves = 2 per SIMD) -----------------------
        // A K step (64 channels of one tap) is 4 phases: L0 (fragment reads of K-half 0),
        // M0 (16 MFMAs), L1, M1.  Waves 4-7 run one phase behind waves 0-3, so on every
        // SIMD one wave is in an MFMA phase while its partner reads LDS.
        // Operand staging is a continuous stream of HALF-K pieces (16 rows x 64 B per
        // global_load_lds): every wave issues 2 pieces per phase -- uniform load on the
        // vector-memory path instead of a burst -- into the half-stage that became free
        // two phases earlier; a piece is first read >= 4 phases after it was issued.
        // Waits are counted (vmcnt(8): the 8 youngest pieces stay in flight across the
        // raw s_barrier), never a drain, except in the last ~1.5 steps.
        //   piece m (per wave):  step = m/8, K-half = (m/4)&1, instruction i = m&3
        //   prologue issues m = 0..11; global phase t issues m = 12+2t, 13+2t
        const int grp = wave >> 2;
        const int K = P.ksteps;
        constexpr int HSTAGE = ROWS * 64;          // bytes of one K-half of a stage
        // half-K staging map: lane -> (row = 16*inst + lane/4, chunk = lane&3), 64-byte rows,
        // physical chunk = logical ^ ((row>>2)&3)  (conflict-free ds_read_b128, 64 B pitch)
        unsigned hofs[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = (i * NW + wave) * 16 + (lane >> 2);
            const int q = (lane & 3) ^ ((r >> 2) & 3);
            if ((i * NW + wave) * 16 < BN) {
                hofs[i] = (unsigned)(nt * BN + r) * (unsigned)P.kbytes + (unsigned)q * 16u;
            } else {
                int pix = m0 + (r - BN);
                pix = pix < HW ? pix : HW - 1;
                int ho, wo;
            divmod_small(pix, S.Wout, invW, ho, wo);
                const unsigned row = (unsigned)(img * Hp + ho * P.stride + 1 - P.pad);
                const unsigned colp = (unsigned)(wo * P.stride + 1 - P.pad);
                hofs[i] = (row * (unsigned)Wp + colp) * (unsigned)cpx + (unsigned)q * 16u;
            }
        }
        // running tap offsets of the two steps pieces are currently issued for
        int tkh = 0, tkw = 0, tc0 = 0, tstep = 0;     // (kh,kw,c0) of step `tstep`
        unsigned offX[2];                             // pixel-side K offset of step s at offX[s&1]
        offX[0] = 0u;
        offX[1] = 0u;
        auto advance_to = [&](int sidx) {             // make offX[sidx&1] valid for step sidx
            while (tstep < sidx) {
                if (++tkw == P.KW) { tkw = 0; if (++tkh == P.KH) { tkh = 0; tc0 += kBK; } }
                ++tstep;
            }
            offX[sidx & 1] = (unsigned)((tkh * Wp + tkw) * P.Cin + tc0) * 2u;
        };
        auto piece = [&](int sidx, int h, int i) {    // i is a compile-time constant at every call site
            if (sidx >= K) return;
            const bool isW = (i * NW + wave) * 16 < BN;
            const char* base = isW ? P.w : S.in;
            const unsigned koff = isW ? (unsigned)sidx * (unsigned)kRowBytes : offX[sidx & 1];
            const unsigned off = hofs[i] + koff + (unsigned)h * 64u;
            char* dst = lds + (sidx & 1) * STAGE + h * HSTAGE + (i * NW + wave) * 16 * 64;
            __builtin_amdgcn_global_load_lds((gvoid*)(base + off), (lvoid*)dst, 16, 0, 0);
        };
        // fragment offsets for 64-byte rows
        const int fsw4 = (frow >> 2) & 3;
        unsigned hroff[2];
#pragma unroll
        for (int k2 = 0; k2 < 2; k2++) hroff[k2] = (unsigned)frow * 64u + (unsigned)(((2 * k2 + (lane >> 5)) ^ fsw4) * 16);
        bf16x8 af[2][TC], bfr[2][TP];
        auto read_half = [&](int stage, int h) {
            const char* sb = lds + stage * STAGE + h * HSTAGE;
#pragma unroll
            for (int k2 = 0; k2 < 2; k2++) {
#pragma unroll
                for (int a = 0; a < TC; a++) af[k2][a] = *(const bf16x8*)(sb + (arow0 + a * 32) * 64 + hroff[k2]);
#pragma unroll
                for (int b = 0; b < TP; b++) bfr[k2][b] = *(const bf16x8*)(sb + (brow0 + b * 32) * 64 + hroff[k2]);
            }
        };
        // 16 MFMAs; the phase's two load pieces are issued in the shadow of the matrix pipe
        // (after the 4th and the 8th MFMA) instead of in front of it
        auto mma_half_ld = [&](int s0, int h0, int i0) {
#pragma unroll
            for (int k2 = 0; k2 < 2; k2++)
#pragma unroll
                for (int a = 0; a < TC; a++) {
#pragma unroll
                    for (int b = 0; b < TP; b++)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[k2][a], bfr[k2][b], acc[a][b], 0, 0, 0);
                    if (k2 == 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        piece(s0, h0, i0 + a);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
        };
        auto phase_end = [&](bool odd, int t) {       // t = global phase index
            if (odd) {
                if (14 + 2 * t <= 8 * K) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        // prologue: (step 0, both halves), (step 1, half 0)
        advance_to(0);
#pragma unroll
        for (int i = 0; i < 4; i++) piece(0, 0, i);
#pragma unroll
        for (int i = 0; i < 4; i++) piece(0, 1, i);
        advance_to(1);
#pragma unroll
        for (int i = 0; i < 4; i++) piece(1, 0, i);
        if (K > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        phase_end(false, -1);
        TSTAMP(1);
        if (grp == 0) {
            for (int j = 0; j < K; j++) {
                const int cur = j & 1;
                piece(j + 1, 1, 0); piece(j + 1, 1, 1);        // t = 4j    : L0
                read_half(cur, 0);
                phase_end(false, 4 * j);
                mma_half_ld(j + 1, 1, 2);                      // t = 4j+1  : M0
                phase_end(true, 4 * j + 1);
                advance_to(j + 2);
                piece(j + 2, 0, 0); piece(j + 2, 0, 1);        // t = 4j+2  : L1
                read_half(cur, 1);
                phase_end(false, 4 * j + 2);
                mma_half_ld(j + 2, 0, 2);                      // t = 4j+3  : M1
                phase_end(true, 4 * j + 3);
            }
        } else {
            piece(1, 1, 0); piece(1, 1, 1);                    // t = 0 (this group idles one phase)
            phase_end(false, 0);
            for (int j = 0; j < K; j++) {
                const int cur = j & 1;
                piece(j + 1, 1, 2); piece(j + 1, 1, 3);        // t = 4j+1  : L0
                read_half(cur, 0);
                phase_end(true, 4 * j + 1);
                advance_to(j + 2);
                mma_half_ld(j + 2, 0, 0);                      // t = 4j+2  : M0
                phase_end(false, 4 * j + 2);
                piece(j + 2, 0, 2); piece(j + 2, 0, 3);        // t = 4j+3  : L1
                read_half(cur, 1);
                phase_end(true, 4 * j + 3);
                mma_half_ld(j + 2, 1, 0);                      // t = 4j+4  : M1
                if (j + 1 < K) phase_end(false, 4 * j + 4);
            }
        }
    } else if constexpr (NS > 2) {
        // ---- ring of NS stages: stage s + NS - 1 is requested when stage s is about to be multiplied; the wait for stage s
        // leaves the NS - 2 younger stages (NL loads each per wave) in flight; one raw barrier per step (everybody's pieces
        // of stage s have landed, everybody is done with stage s - 1, whose buffer the new request overwrites)
        static_assert(NS == 3 || NS == 4, "ring depth");
        const int K = P.ksteps;
        for (int s0 = 0; s0 < NS - 1 && s0 < K; s0++) issue(s0, s0);
        int cur = 0, nxt = NS - 1;
        for (int step = 0; step < K; step++) {
            const int ahead = K - 1 - step;
            if (ahead >= NS - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * NL) : "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (step == 0) { TSTAMP(1); }
            if (step + NS - 1 < K) issue(nxt, step + NS - 1);
            const char* sb = lds + cur * STAGE;
            // one wave per SIMD here: the fragments of k16 sub-step ks + 1 are requested before the MFMAs of sub-step ks
            // (pinned: left alone the compiler waits for every read group right in front of its MFMAs)
            bf16x8 af[2][TC], bfr[2][TP];
            auto fetch = [&](int ks, int set) {
#pragma unroll
                for (int a = 0; a < TC; a++) af[set][a] = *(const bf16x8*)(sb + (arow0 + a * 32) * kRowBytes + roff[ks]);
#pragma unroll
                for (int b = 0; b < TP; b++) bfr[set][b] = *(const bf16x8*)(sb + (brow0 + b * 32) * kRowBytes + roff[ks]);
            };
            fetch(0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, TC + TP, 0);
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                if (ks < 3) {
                    fetch(ks + 1, (ks + 1) & 1);
                    __builtin_amdgcn_sched_group_barrier(0x100, TC + TP, 0);
                }
#pragma unroll
                for (int a = 0; a < TC; a++)
#pragma unroll
                    for (int b = 0; b < TP; b++)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][a], bfr[ks & 1][b], acc[a][b], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, TC * TP, 0);
            }
            cur = cur + 1 == NS ? 0 : cur + 1;
            nxt = nxt + 1 == NS ? 0 : nxt + 1;
        }
    } else {
    issue(0, 0);
    for (int step = 0; step < P.ksteps; step++) {
        const int cur = step & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (step == 0) { TSTAMP(1); }
        if (step + 1 < P.ksteps) issue(cur ^ 1, step + 1);
        const char* sb = lds + cur * STAGE;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            bf16x8 af[TC], bfr[TP];
#pragma unroll
            for (int a = 0; a < TC; a++) af[a] = *(const bf16x8*)(sb + (arow0 + a * 32) * kRowBytes + roff[ks]);
#pragma unroll
            for (int b = 0; b < TP; b++) bfr[b] = *(const bf16x8*)(sb + (brow0 + b * 32) * kRowBytes + roff[ks]);
#pragma unroll
            for (int a = 0; a < TC; a++)
#pragma unroll
                for (int b = 0; b < TP; b++)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
        }
    }
    }

    TSTAMP(2);
    // ------------------------------------------------------------ epilogue
    // Accumulators go through LDS as fp32 [pixel][channel] so that global traffic is
    // coalesced: every lane then owns 8 consecutive channels (16 B bf16) of one pixel,
    // 16-lane groups cover 256 contiguous bytes; residual reads are coalesced the same
    // way, and 8 channels == one GroupNorm group.
    const bool relu = P.flags & DAFNE_CONV_RELU;
    const float relu_lo = relu ? 0.f : -__builtin_inff();
    const bool has_res = P.flags & DAFNE_CONV_RESIDUAL;
    const bool has_up = P.flags & DAFNE_CONV_UPSAMPLE_ADD;
    const bool out_f32 = P.flags & DAFNE_CONV_OUT_F32;
    const bool gn = P.flags & DAFNE_CONV_GN_STATS;
    const int half = lane >> 5;
    const int col = tid % CH;
    char* stg = lds;
    float* red = (float*)(lds + BM * ROWF);   // [NW][CH][2]

    if (!has_res && !has_up && !out_f32) {
        // ---- plain bf16 output: bias / ReLU / GN sums in the accumulator layout, bf16 tile
        // staged ONCE through LDS ([px][cout], 16-B row pad), then 16-byte coalesced stores.
        constexpr int ROWB = BN * 2 + 16;
        constexpr int CHB = BN / 8;
        float4 bia4[TC][4];
#pragma unroll
        for (int a = 0; a < TC; a++)
#pragma unroll
            for (int g = 0; g < 4; g++)
                bia4[a][g] = P.bias ? *(const float4*)(P.bias + nt * BN + (wc * TC + a) * 32 + 8 * g + 4 * half)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
        float gsum[TC][4], gsq[TC][4];
#pragma unroll
        for (int a = 0; a < TC; a++)
#pragma unroll
            for (int g = 0; g < 4; g++) gsum[a][g] = gsq[a][g] = 0.f;
        __syncthreads();   // every wave is done reading the stage buffers
#pragma unroll
        for (int b = 0; b < TP; b++) {
            const int px = (wp * TP + b) * 32 + frow;
            const bool valid = m0 + px < HW;
#pragma unroll
            for (int a = 0; a < TC; a++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    // (branch-free: ReLU as a max with 0 / -inf, the GroupNorm sums added under a select -- x + 0 is exact; round 5)
                    const float v0 = fmaxf(acc[a][b][4 * g] + bia4[a][g].x, relu_lo), v1 = fmaxf(acc[a][b][4 * g + 1] + bia4[a][g].y, relu_lo);
                    const float v2 = fmaxf(acc[a][b][4 * g + 2] + bia4[a][g].z, relu_lo), v3 = fmaxf(acc[a][b][4 * g + 3] + bia4[a][g].w, relu_lo);
                    gsum[a][g] += valid ? (v0 + v1) + (v2 + v3) : 0.f;
                    gsq[a][g] += valid ? (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3) : 0.f;
                    uint2 pk;
                    pk.x = pack_bf16(v0, v1);
                    pk.y = pack_bf16(v2, v3);
                    const int co = (wc * TC + a) * 32 + 8 * g + 4 * half;
                    *(uint2*)(stg + px * ROWB + co * 2) = pk;
                }
        }
        float* redb = (float*)(lds + BM * ROWB);   // [NW][TC*4][2]
        if (gn) {
            // deterministic: butterfly over the wave (32 pixels x 2 channel halves), then a
            // fixed-order sum over the WP pixel-waves through LDS
#pragma unroll
            for (int a = 0; a < TC; a++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    float sv = gsum[a][g], qv = gsq[a][g];
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) {
                        sv += __shfl_xor(sv, o, 64);
                        qv += __shfl_xor(qv, o, 64);
                    }
                    if (lane == 0) {
                        redb[(wave * TC * 4 + a * 4 + g) * 2 + 0] = sv;
                        redb[(wave * TC * 4 + a * 4 + g) * 2 + 1] = qv;
                    }
                }
        }
        __syncthreads();
        if (gn && tid < BN / 8) {
            const int wcc = tid / (TC * 4), ag = tid % (TC * 4);
            float sv = 0.f, qv = 0.f;
#pragma unroll
            for (int p2 = 0; p2 < WP; p2++) {
                sv += redb[((wcc * WP + p2) * TC * 4 + ag) * 2 + 0];
                qv += redb[((wcc * WP + p2) * TC * 4 + ag) * 2 + 1];
            }
            const int group = (nt * BN) / 8 + tid;
            if (group < P.Cout / 8) {
                float* o = P.gn_partial + ((size_t)mt * (P.Cout / 8) + group) * 2;
                o[0] = sv;
                o[1] = qv;
            }
        }
        constexpr int CPT = BM * CHB / NT;       // 16-byte chunks per thread
#pragma unroll
        for (int i = 0; i < CPT; i++) {
            const int idx = tid + i * NT;
            const int p = idx / CHB, cc = idx - p * CHB;
            const int m = m0 + p;
            if (m < HW) {
                int ho, wo;
            divmod_small(m, S.Wout, invW, ho, wo);
                const size_t opix = ((size_t)(img * (S.Hout + 2) + ho + 1) * (S.Wout + 2) + wo + 1);
                const uint4 v = *(const uint4*)(stg + p * ROWB + cc * 16);
                *(uint4*)(S.out + (opix * P.Cout + nt * BN + cc * 8) * 2) = v;
            }
        }
        TSTAMP(3);
        TSTAMP(4);
    } else
#pragma unroll
    for (int ep = 0; ep < EPASS; ep++) {
        __syncthreads();   // previous readers of this LDS region are done
        // waves whose channel range falls into this pass deposit their accumulators
        const int cw0 = wc * TC * 32 - ep * EN;          // wave's first channel relative to the pass
        if (cw0 >= 0 && cw0 < EN) {
#pragma unroll
            for (int b = 0; b < TP; b++) {
                const int px = (wp * TP + b) * 32 + frow;
#pragma unroll
                for (int a = 0; a < TC; a++)
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const int co = cw0 + a * 32 + 8 * g + 4 * half;
                        float4 v = make_float4(acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2],
                                               acc[a][b][4 * g + 3]);
                        *(float4*)(stg + px * ROWF + co * 4) = v;
                    }
            }
        }
        __syncthreads();
        const int cobase = nt * BN + ep * EN + col * 8;   // first of this lane's 8 channels
        float bia[8];
#pragma unroll
        for (int k = 0; k < 8; k++) bia[k] = 0.f;
        if (P.bias) {
            const float4 b0 = *(const float4*)(P.bias + cobase), b1 = *(const float4*)(P.bias + cobase + 4);
            bia[0] = b0.x; bia[1] = b0.y; bia[2] = b0.z; bia[3] = b0.w;
            bia[4] = b1.x; bia[5] = b1.y; bia[6] = b1.z; bia[7] = b1.w;
        }
        float gs = 0.f, gq = 0.f;
        constexpr int PPT = BM / (NT / CH);     // pixels per thread per pass
        // residual / top-down rows first: all PPT 16-byte loads in flight together
        uint4 rr[PPT];
        if (kEarlyRes) {
#pragma unroll
            for (int i = 0; i < PPT; i++) rr[i] = rr0[kEarlyRes ? i : 0];
        } else if (has_res || has_up) {
#pragma unroll
            for (int i = 0; i < PPT; i++) {
                const int p = tid / CH + i * (NT / CH);
                int m = m0 + p;
                m = m < HW ? m : HW - 1;
                int ho, wo;
            divmod_small(m, S.Wout, invW, ho, wo);
                const size_t rpix = has_up
                    ? ((size_t)(img * (S.Hout / 2 + 2) + ho / 2 + 1) * (S.Wout / 2 + 2) + wo / 2 + 1)
                    : ((size_t)(img * (S.Hout + 2) + ho + 1) * (S.Wout + 2) + wo + 1);
                rr[i] = *(const uint4*)(S.res + (rpix * P.Cout + cobase) * 2);
            }
        }
#pragma unroll
        for (int i = 0; i < PPT; i++) {
            const int p = tid / CH + i * (NT / CH);
            const int m = m0 + p;
            if (m >= HW) continue;
            int ho, wo;
            divmod_small(m, S.Wout, invW, ho, wo);
            const float4 x0 = *(const float4*)(stg + p * ROWF + col * 32);
            const float4 x1 = *(const float4*)(stg + p * ROWF + col * 32 + 16);
            float v[8] = {x0.x + bia[0], x0.y + bia[1], x0.z + bia[2], x0.w + bia[3],
                          x1.x + bia[4], x1.y + bia[5], x1.z + bia[6], x1.w + bia[7]};
            if (has_res || has_up) {
                const unsigned u[4] = {rr[i].x, rr[i].y, rr[i].z, rr[i].w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    v[2 * k] += bf2f((unsigned short)(u[k] & 0xffff));
                    v[2 * k + 1] += bf2f((unsigned short)(u[k] >> 16));
                }
            }
            if (relu) {
#pragma unroll
                for (int k = 0; k < 8; k++) v[k] = fmaxf(v[k], 0.f);
            }
            if (gn) {
#pragma unroll
                for (int k = 0; k < 8; k++) { gs += v[k]; gq += v[k] * v[k]; }
            }
            if (out_f32) {
                float* o = (float*)S.out + ((size_t)(img * S.Hout + ho) * S.Wout + wo) * P.Cout + cobase;
#pragma unroll
                for (int k = 0; k < 8; k++) if (cobase + k < P.Cout) o[k] = v[k];
            } else {
                const size_t opix = ((size_t)(img * (S.Hout + 2) + ho + 1) * (S.Wout + 2) + wo + 1);
                uint4 pk;
                pk.x = pack_bf16(v[0], v[1]);
                pk.y = pack_bf16(v[2], v[3]);
                pk.z = pack_bf16(v[4], v[5]);
                pk.w = pack_bf16(v[6], v[7]);
                *(uint4*)(S.out + (opix * P.Cout + cobase) * 2) = pk;
            }
        }
        TSTAMP(3 + ep);
        if (gn) {
            // deterministic: butterfly over the lanes that share a channel group, then a
            // fixed-order sum over the waves through LDS
#pragma unroll
            for (int o = 32; o >= CH; o >>= 1) {
                gs += __shfl_xor(gs, o, 64);
                gq += __shfl_xor(gq, o, 64);
            }
            if (lane < CH) {
                red[(wave * CH + lane) * 2 + 0] = gs;
                red[(wave * CH + lane) * 2 + 1] = gq;
            }
            __syncthreads();
            if (tid < CH) {
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int w2 = 0; w2 < NW; w2++) {
                    s += red[(w2 * CH + tid) * 2 + 0];
                    q += red[(w2 * CH + tid) * 2 + 1];
                }
                const int group = (nt * BN + ep * EN) / 8 + tid;
                if (group < P.Cout / 8) {
                    float* o = P.gn_partial + ((size_t)mt * (P.Cout / 8) + group) * 2;
                    o[0] = s;
                    o[1] = q;
                }
            }
        }
    }
#ifdef DAFNE_CONV_TIMING
    // debug build only (scratch/conv_timeline.py): per-wave cycle stamps into d_gn_partial
    if (lane == 0 && P.gn_partial && !(P.flags & DAFNE_CONV_GN_STATS)) {
        unsigned long long* o = (unsigned long long*)P.gn_partial + ((size_t)blockIdx.x * NW + wave) * 8;
        for (int k = 0; k < 5; k++) o[k] = ts[k];
        o[5] = rt0;
        o[6] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// ---------------------------------------------------------------------------------------
// Persistent streaming variant (128 cout x 128 px tiles, 4 waves, 2 blocks per CU).
//
// The 1x1 convolutions of the bottlenecks have K = 64..1024: a tile is 2..32 half-K stages
// of MFMA work between a prologue (address set-up, first-load latency) and an epilogue
// (residual read, output write).  Launched one tile per workgroup, every K step waited a
// full memory latency (one stage of prefetch) and prologue/epilogue were never overlapped:
// those layers ran at ~2.8 TB/s and ~15 % of the matrix peak.  Here a workgroup walks over
// many tiles and the operand stream never stops:
//   * ring of kSNS half-K stages (32 channels = 64-byte rows, 16 KiB per stage), filled by
//     `global_load_lds`; the loads of tile i+1's first stages are issued during the last
//     iterations of tile i, so they are in flight under tile i's epilogue;
//   * waits are counted (`s_waitcnt vmcnt(n)`, n = the vector-memory instructions issued
//     after the stage being waited for) -- never a drain except for the very last stage;
//   * the residual tile is DMA'd (`global_load_lds`, source-side XOR swizzle) into a bf16
//     [px][cout] LDS tile at the first iteration of its tile; the epilogue adds it in
//     fp32 in the accumulator layout, rounds once, writes the bf16 result back IN PLACE,
//     and the tile leaves with 16-byte coalesced stores.
// gfx950 counts stores in vmcnt as well: every store of the epilogue is issued
// unconditionally (ragged tiles re-write their last valid pixel) so that the counts hold.
constexpr int kSNS = 3;                 // ring depth in half-K stages
constexpr int kSHS = 256 * 64;          // one half-K stage: 128 weight rows + 128 pixel rows x 64 B
constexpr int kSRes = 128 * 256;        // bf16 [128 px][128 cout] residual / output staging tile

struct TileInfo {
    int nt, si, img, m0;
};

__device__ __forceinline__ TileInfo tile_info(const ConvDev& P, int logical) {
    TileInfo t;
    t.nt = logical % P.ntiles;
    const int mt = logical / P.ntiles;
    int si = 0;
#pragma unroll
    for (int k = 1; k < kMaxSegs; k++)
        if (k < P.n_segs && mt >= P.seg[k].tile0) si = k;
    t.si = si;
    t.img = (mt - P.seg[si].tile0) / P.seg[si].tiles_per_img;
    t.m0 = ((mt - P.seg[si].tile0) % P.seg[si].tiles_per_img) * 128;
    return t;
}


#define DAFNE_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

__global__ void __launch_bounds__(256, 2) conv_stream_kernel(ConvDev P) {
    constexpr int NW = 4, WP = 2, TC = 2, TP = 2;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* ring = lds;
    char* resb = lds + kSNS * kSHS;
    const unsigned resb_off = (unsigned)(size_t)(__attribute__((address_space(3))) char*)resb;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave / WP, wp = wave % WP;
    const int frow = lane & 31, half = lane >> 5;
    const int T = P.mtiles * P.ntiles;
    const int G = gridDim.x;
    const int my_tiles = (T - (int)blockIdx.x + G - 1) / G;
    const int H = 2 * P.ksteps;
    const bool has_res = P.flags & DAFNE_CONV_RESIDUAL;
    const bool relu = P.flags & DAFNE_CONV_RELU;

    // ---- issue cursor: runs kSNS-1 half-K stages ahead of the consumer, across tiles --------
    int ic_tile = 0, ic_h = 0, ic_slot = 0, ic_Wp = 0;
    int kh = 0, kw = 0, c0 = 0;
    unsigned offX = 0;
    unsigned hofs[4];
    const char* xin = nullptr;
    auto setup_issue = [&](int seq) {
        const TileInfo t = tile_info(P, xcd_remap((int)blockIdx.x + seq * G, T));
        const SegDev& S = P.seg[t.si];
        const int HW = S.Hout * S.Wout;
        const int Wp = S.Win + 2, Hp = S.Hin + 2;
        const float invW = 1.0f / (float)S.Wout;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = (i * NW + wave) * 16 + (lane >> 2);
            const int q = (lane & 3) ^ ((r >> 2) & 3);
            if (i < 2) {
                hofs[i] = (unsigned)(t.nt * 128 + r) * (unsigned)P.kbytes + (unsigned)q * 16u;
            } else {
                int pix = t.m0 + (r - 128);
                pix = pix < HW ? pix : HW - 1;
                int ho, wo;
                divmod_small(pix, S.Wout, invW, ho, wo);
                const unsigned row = (unsigned)(t.img * Hp + ho * P.stride + 1 - P.pad);
                const unsigned colp = (unsigned)(wo * P.stride + 1 - P.pad);
                hofs[i] = (row * (unsigned)Wp + colp) * (unsigned)(P.Cin * 2) + (unsigned)q * 16u;
            }
        }
        xin = S.in;
        ic_Wp = Wp;
        kh = kw = c0 = 0;
        offX = 0;
    };
    auto issue_next = [&]() {
        if (ic_tile >= my_tiles) return;
        const unsigned hb = (unsigned)(ic_h & 1) * 64u;
        const unsigned koffW = (unsigned)(ic_h >> 1) * (unsigned)kRowBytes + hb;
        const unsigned koffX = offX + hb;
        char* dst = ring + ic_slot * kSHS + wave * 1024;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const char* base = i < 2 ? P.w : xin;
            const unsigned off = hofs[i] + (i < 2 ? koffW : koffX);
            __builtin_amdgcn_global_load_lds((gvoid*)(base + off), (lvoid*)(dst + i * 4096), 16, 0, 0);
        }
        ic_slot = ic_slot == kSNS - 1 ? 0 : ic_slot + 1;
        if (ic_h & 1) {
            if (++kw == P.KW) { kw = 0; if (++kh == P.KH) { kh = 0; c0 += kBK; } }
            offX = (unsigned)((kh * ic_Wp + kw) * P.Cin + c0) * 2u;
        }
        if (++ic_h == H) {
            ic_h = 0;
            if (++ic_tile < my_tiles) setup_issue(ic_tile);
        }
    };

    // fragment read offsets inside a half-K stage (64-byte rows, chunk ^ ((row>>2)&3))
    const int fsw4 = (frow >> 2) & 3;
    unsigned hroff[2];
#pragma unroll
    for (int k2 = 0; k2 < 2; k2++) hroff[k2] = (unsigned)frow * 64u + (unsigned)(((2 * k2 + half) ^ fsw4) * 16);
    const int arow0 = wc * TC * 32;
    const int brow0 = 128 + wp * TP * 32;

    if (my_tiles > 0) setup_issue(0);
#pragma unroll
    for (int k = 0; k < kSNS - 1; k++) issue_next();

    int slot = 0;
    for (int kk = 0; kk < my_tiles; kk++) {
        const TileInfo t = tile_info(P, xcd_remap((int)blockIdx.x + kk * G, T));
        const SegDev& S = P.seg[t.si];
        const int HW = S.Hout * S.Wout;
        const float invW = 1.0f / (float)S.Wout;
        const bool last_tile = kk == my_tiles - 1;

        f32x16 acc[TC][TP];
#pragma unroll
        for (int a = 0; a < TC; a++)
#pragma unroll
            for (int b = 0; b < TP; b++)
#pragma unroll
                for (int k = 0; k < 16; k++) acc[a][b][k] = 0.f;

        auto lds_barrier = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        auto compute = [&]() {
            const char* sb = ring + slot * kSHS;
            slot = slot == kSNS - 1 ? 0 : slot + 1;
            bf16x8 af[2][TC], bfr[2][TP];
#pragma unroll
            for (int k2 = 0; k2 < 2; k2++) {
#pragma unroll
                for (int a = 0; a < TC; a++) af[k2][a] = *(const bf16x8*)(sb + (arow0 + a * 32) * 64 + hroff[k2]);
#pragma unroll
                for (int b = 0; b < TP; b++) bfr[k2][b] = *(const bf16x8*)(sb + (brow0 + b * 32) * 64 + hroff[k2]);
            }
#pragma unroll
            for (int k2 = 0; k2 < 2; k2++)
#pragma unroll
                for (int a = 0; a < TC; a++)
#pragma unroll
                    for (int b = 0; b < TP; b++)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[k2][a], bfr[k2][b], acc[a][b], 0, 0, 0);
        };

        // ---- h = 0.  Waits: n = vector-memory instructions issued after the stage's 4 pieces.
        // Issued after stage (kk,0): [4 pieces of (kk,1)] [8 stores of tile kk-1]
        if (last_tile && H == 1) DAFNE_VMCNT(0);
        else if (kk) DAFNE_VMCNT(12);
        else DAFNE_VMCNT(4);
        lds_barrier();
        if (has_res) {
            // residual tile -> LDS: piece = 4 pixel rows x 256 B; lane = (row, physical chunk),
            // fetches logical chunk = physical ^ (row & 15)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int p = (i * NW + wave) * 4 + (lane >> 4);
                const int q = (lane & 15) ^ (p & 15);
                int m = t.m0 + p;
                m = m < HW ? m : HW - 1;
                int ho, wo;
                divmod_small(m, S.Wout, invW, ho, wo);
                const unsigned rpix = (unsigned)((t.img * (S.Hout + 2) + ho + 1) * (S.Wout + 2) + wo + 1);
                const unsigned off = (rpix * (unsigned)P.Cout + (unsigned)(t.nt * 128 + q * 8)) * 2u;
                __builtin_amdgcn_global_load_lds((gvoid*)(S.res + off), (lvoid*)(resb + (i * NW + wave) * 1024), 16, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        issue_next();
        __builtin_amdgcn_sched_barrier(0);
        compute();

        // ---- h = 1: issued after stage (kk,1): [8 stores of tile kk-1] [8 residual pieces] [4 pieces]
        if (last_tile && H == 2) DAFNE_VMCNT(0);
        else if (kk && has_res) DAFNE_VMCNT(20);
        else if (kk || has_res) DAFNE_VMCNT(12);
        else DAFNE_VMCNT(4);
        lds_barrier();
        issue_next();
        __builtin_amdgcn_sched_barrier(0);
        compute();

        // ---- h >= 2: [4 pieces]
        for (int h = 2; h < H; h++) {
            if (last_tile && h == H - 1) DAFNE_VMCNT(0); else DAFNE_VMCNT(4);
            lds_barrier();
            issue_next();
            __builtin_amdgcn_sched_barrier(0);
            compute();
        }

        // ------------------------------------------------------------ epilogue
        // this wave's residual pieces have landed: after them came >= 8 ring pieces
        if (last_tile) DAFNE_VMCNT(0); else DAFNE_VMCNT(8);
        lds_barrier();     // every wave's residual pieces are in LDS
        // bias through the scalar cache (s_load does not touch vmcnt, so the operand stream
        // stays in flight); lanes pick their half of each 8-channel group
        const float* bc = P.bias + t.nt * 128 + wc * TC * 32;
        const unsigned rmask = has_res ? 0xffffffffu : 0u;
        const float lo = relu ? 0.f : -__builtin_huge_valf();
#pragma unroll
        for (int a = 0; a < TC; a++) {
            f32x4 bl[4], bh[4];
            asm volatile(
                "s_load_dwordx4 %0, %8, 0x0\n\ts_load_dwordx4 %1, %8, 0x10\n\t"
                "s_load_dwordx4 %2, %8, 0x20\n\ts_load_dwordx4 %3, %8, 0x30\n\t"
                "s_load_dwordx4 %4, %8, 0x40\n\ts_load_dwordx4 %5, %8, 0x50\n\t"
                "s_load_dwordx4 %6, %8, 0x60\n\ts_load_dwordx4 %7, %8, 0x70\n\ts_waitcnt lgkmcnt(0)"
                : "=&s"(bl[0]), "=&s"(bh[0]), "=&s"(bl[1]), "=&s"(bh[1]), "=&s"(bl[2]), "=&s"(bh[2]), "=&s"(bl[3]), "=&s"(bh[3])
                : "s"(bc + a * 32)
                : "memory");
            // LDS accesses of the epilogue are inline asm: the compiler drains vmcnt in front of
            // every LDS access it cannot disambiguate from a pending LDS-DMA (the ring prefetch)
            u32x2 rc[4][TP];
            unsigned caddr[4][TP];
#pragma unroll
            for (int g = 0; g < 4; g++)
#pragma unroll
                for (int b = 0; b < TP; b++) {
                    const int px = (wp * TP + b) * 32 + frow;
                    const int chunk = (wc * TC + a) * 4 + g;            // 16-byte chunk = 8 channels
                    caddr[g][b] = resb_off + (unsigned)(px * 256 + ((chunk ^ (px & 15)) * 16) + 8 * half);
                    asm volatile("ds_read_b64 %0, %1" : "=v"(rc[g][b]) : "v"(caddr[g][b]) : "memory");
                }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(rc[0][0]), "+v"(rc[0][1]), "+v"(rc[1][0]), "+v"(rc[1][1]), "+v"(rc[2][0]), "+v"(rc[2][1]),
                           "+v"(rc[3][0]), "+v"(rc[3][1])
                         :
                         : "memory");
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const f32x4 bv = half ? bh[g] : bl[g];
#pragma unroll
                for (int b = 0; b < TP; b++) {
                    u32x2 r = rc[g][b];
                    r.x &= rmask;
                    r.y &= rmask;
                    const float v0 = fmaxf(acc[a][b][4 * g] + bv[0] + bf2f((unsigned short)(r.x & 0xffff)), lo);
                    const float v1 = fmaxf(acc[a][b][4 * g + 1] + bv[1] + bf2f((unsigned short)(r.x >> 16)), lo);
                    const float v2 = fmaxf(acc[a][b][4 * g + 2] + bv[2] + bf2f((unsigned short)(r.y & 0xffff)), lo);
                    const float v3 = fmaxf(acc[a][b][4 * g + 3] + bv[3] + bf2f((unsigned short)(r.y >> 16)), lo);
                    u32x2 pk;
                    pk.x = pack_bf16(v0, v1);
                    pk.y = pack_bf16(v2, v3);
                    asm volatile("ds_write_b64 %0, %1" ::"v"(caddr[g][b]), "v"(pk) : "memory");
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        lds_barrier();
        {
            const int cc = tid & 15;
            const int plast = HW - 1 - t.m0;      // last valid pixel row of a ragged tile
            u32x4 ov[8];
            int pr[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                int p = (tid >> 4) + 16 * i;
                p = p < plast ? p : plast;
                pr[i] = p;
                const unsigned ad = resb_off + (unsigned)(p * 256 + ((cc ^ (p & 15)) * 16));
                asm volatile("ds_read_b128 %0, %1" : "=v"(ov[i]) : "v"(ad) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(ov[0]), "+v"(ov[1]), "+v"(ov[2]), "+v"(ov[3]), "+v"(ov[4]), "+v"(ov[5]), "+v"(ov[6]), "+v"(ov[7])
                         :
                         : "memory");
#pragma unroll
            for (int i = 0; i < 8; i++) {
                int ho, wo;
                divmod_small(t.m0 + pr[i], S.Wout, invW, ho, wo);
                const size_t opix = (size_t)((t.img * (S.Hout + 2) + ho + 1) * (S.Wout + 2) + wo + 1);
                *(u32x4*)(S.out + (opix * P.Cout + t.nt * 128 + cc * 8) * 2) = ov[i];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---------------------------------------------------------------------------------------
// Weight-stationary streaming kernel for the 1x1 layers with Cin <= 256 (res2/res3/res4 conv3
// and the stage shortcuts: K = 64..256, HBM-bound, residual read + 4x wider output).
//
// A workgroup keeps its 128 output channels' weights in REGISTERS (MFMA A fragments, loaded
// once; the tile order keeps the channel tile fixed per workgroup) and streams pixel tiles:
// the LDS ring then carries only the pixel operand (8 KiB half-K stages, 6 deep = 5 in
// flight, against 2 in the generic streaming kernel), LDS fragment reads halve, and the
// L2->LDS traffic per tile drops from 128 KiB to 64 KiB (Cin = 256).
// vmcnt bookkeeping is exact: K is a template parameter, so the number of vector-memory
// instructions issued after any awaited stage (ring pieces, residual DMA pieces, output stores
// of earlier tiles) is a compile-time function of the position in the tile (ws_after).
// Ring geometry by row bytes RB: 64 -> half-K stages (32 channels, 8 KiB, 6 deep), 128 -> full K steps (64
// channels, 16 KiB, 3 deep).  Same 48 KiB; the half-K ring keeps more loads in flight (HBM-bound K = 64/128
// layers), the full-step ring halves the barriers per MFMA (res4 conv3, K = 256: an iteration is ~950 cycles of
// which only 256-512 are MFMA issue).
constexpr int ws_ns(int RB) { return RB == 128 ? 3 : 6; }
constexpr int ws_pps(int RB) { return RB == 128 ? 4 : 2; }       // DMA pieces per wave and stage
constexpr int kWRing = 6 * 128 * 64;                             // = 3 * 128 * 128

constexpr int ws_after(int h, int H, bool res, int RB) {
    const int NS = ws_ns(RB);
    int n = ws_pps(RB) * (NS - 2);                                // ring pieces of the NS-2 younger stages
    for (int d = 1; d <= NS - 1; d++)
        if (((h - d) % H + H) % H == H - 1) n += res ? 16 : 8;    // a tile end: [8 residual loads of the next tile] 8 output stores
    return n;
}

template <int N>
__device__ __forceinline__ void vmcnt_le() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// residual reads and output stores are streamed once: the non-temporal hint keeps them from evicting
// the pixel tiles that the 2..8 channel-tile siblings of a workgroup re-read from L2 (measured
// 116 vs 125 us on res2 conv3, 44 vs 47 us on res4 conv3)
#define WS_NT " nt"
#define WS_STORE(p, v) __builtin_nontemporal_store(v, p)

template <int WC, int WP, int KS, bool RES, int RB>
__global__ void __launch_bounds__(256, 2) conv_ws_kernel(ConvDev P) {
    constexpr int NW = 4, TC = 4 / WC, TP = 4 / WP, NS = ws_ns(RB), PPS = ws_pps(RB);
    constexpr int H = KS * (128 / RB);            // stages per tile
    constexpr int CPS = RB / 32;                  // 16-deep K chunks per stage
    constexpr int STAGE = 128 * RB;               // bytes
    constexpr int RPP = 1024 / RB;                // pixel rows per DMA piece
    static_assert(WC * WP == NW && ws_after(H - 1, H, true, RB) <= 63 && ws_after(0, H, true, RB) <= 63, "configuration");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* ring = lds;
    char* resb = lds + kWRing;
    const unsigned resb_off = (unsigned)(size_t)(__attribute__((address_space(3))) char*)resb;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave / WP, wp = wave % WP;
    const int frow = lane & 31, half = lane >> 5;
    const int T = P.mtiles * P.ntiles;
    const int G = gridDim.x;
    const int my_tiles = (T - (int)blockIdx.x + G - 1) / G;
    const int total = my_tiles * H;
    const bool relu = P.flags & DAFNE_CONV_RELU;

    // ---- issue cursor: kWNS-1 half-K stages ahead of the consumer, across tiles ---------------
    int ic_tile = 0, ic_h = 0, ic_slot = 0;
    unsigned hofs[PPS];
    const char* xin = nullptr;
    auto setup_issue = [&](int seq) {
        const TileInfo t = tile_info(P, xcd_remap((int)blockIdx.x + seq * G, T));
        const SegDev& S = P.seg[t.si];
        const int HW = S.Hout * S.Wout;
        const int Wp = S.Win + 2, Hp = S.Hin + 2;
        const float invW = 1.0f / (float)S.Wout;
#pragma unroll
        for (int i = 0; i < PPS; i++) {
            const int r = (i * NW + wave) * RPP + (RB == 128 ? lane >> 3 : lane >> 2);
            const int q = RB == 128 ? (lane & 7) ^ ((r >> 1) & 7) : (lane & 3) ^ ((r >> 2) & 3);
            int pix = t.m0 + r;
            pix = pix < HW ? pix : HW - 1;
            int ho, wo;
            divmod_small(pix, S.Wout, invW, ho, wo);
            const unsigned row = (unsigned)(t.img * Hp + ho * P.stride + 1 - P.pad);
            const unsigned colp = (unsigned)(wo * P.stride + 1 - P.pad);
            hofs[i] = (row * (unsigned)Wp + colp) * (unsigned)(P.Cin * 2) + (unsigned)q * 16u;
        }
        xin = S.in;
    };
    auto issue_next = [&]() {
        if (ic_tile >= my_tiles) return;
        char* dst = ring + ic_slot * STAGE + wave * 1024;
#pragma unroll
        for (int i = 0; i < PPS; i++)
            __builtin_amdgcn_global_load_lds((gvoid*)(xin + hofs[i] + (unsigned)ic_h * (unsigned)RB), (lvoid*)(dst + i * 4096), 16, 0, 0);
        ic_slot = ic_slot == NS - 1 ? 0 : ic_slot + 1;
        if (++ic_h == H) {
            ic_h = 0;
            if (++ic_tile < my_tiles) setup_issue(ic_tile);
        }
    };

    const int fsw = RB == 128 ? (frow >> 1) & 7 : (frow >> 2) & 3;
    unsigned hroff[CPS];
#pragma unroll
    for (int k2 = 0; k2 < CPS; k2++) hroff[k2] = (unsigned)frow * (unsigned)RB + (unsigned)(((2 * k2 + half) ^ fsw) * 16);
    const int brow0 = wp * TP * 32;

    // residual rows of a tile live in registers (16 B per lane and row: lane = (pixel row tid/16 + 16 i,
    // chunk tid%16), the mapping of the output stores).  They are fetched a whole tile ahead -- at the
    // start of the previous tile's epilogue -- because vmcnt retires in order: a residual load issued
    // next to ring pieces would make every later ring wait sit out its HBM latency.
    u32x4 rr[RES ? 8 : 1];
    auto fetch_residual = [&](int seq) {
        const TileInfo t = tile_info(P, xcd_remap((int)blockIdx.x + seq * G, T));
        const SegDev& S = P.seg[t.si];
        const int HW = S.Hout * S.Wout;
        const float invW = 1.0f / (float)S.Wout;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            int m = t.m0 + (tid >> 4) + 16 * i;
            m = m < HW ? m : HW - 1;
            int ho, wo;
            divmod_small(m, S.Wout, invW, ho, wo);
            const size_t rpix = (size_t)((t.img * (S.Hout + 2) + ho + 1) * (S.Wout + 2) + wo + 1);
            const char* src = S.res + (rpix * P.Cout + t.nt * 128 + (tid & 15) * 8) * 2;
            asm volatile("global_load_dwordx4 %0, %1, off" WS_NT : "=v"(rr[RES ? i : 0]) : "v"(src) : "memory");
        }
    };
    if (RES && my_tiles > 0) fetch_residual(0);
    if (my_tiles > 0) setup_issue(0);
#pragma unroll
    for (int k = 0; k < NS - 1; k++) issue_next();

#ifdef DAFNE_WS_TIMING
    // debug build only (scratch/ws_timeline.py): cycle stamps of wave 0 into d_gn_partial, 64 slots per workgroup
    unsigned long long* tlog = (unsigned long long*)P.gn_partial + (size_t)blockIdx.x * 64;
    int tl = 2;
    if (tid == 0 && P.gn_partial) { tlog[0] = __builtin_amdgcn_s_memrealtime(); tlog[1] = __builtin_readcyclecounter(); }
#define WS_STAMP() do { if (tid == 0 && P.gn_partial && tl < 64) tlog[tl++] = __builtin_readcyclecounter(); } while (0)
#else
#define WS_STAMP()
#endif
    bf16x8 afr[TC][KS * 4];      // this wave's weights: [channel tile][16-deep K chunk]
    int cur_nt = -1;
    int slot = 0;
    auto lds_barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    for (int kk = 0; kk < my_tiles; kk++) {
        const TileInfo t = tile_info(P, xcd_remap((int)blockIdx.x + kk * G, T));
        const SegDev& S = P.seg[t.si];
        const int HW = S.Hout * S.Wout;
        const float invW = 1.0f / (float)S.Wout;
        if (t.nt != cur_nt) {
            cur_nt = t.nt;
#pragma unroll
            for (int a = 0; a < TC; a++) {
                const char* wr = P.w + (size_t)(t.nt * 128 + (wc * TC + a) * 32 + frow) * (size_t)P.kbytes + half * 16;
                // inline asm: a compiler-visible load would make it drain vmcnt in front of the first
                // MFMA of EVERY tile (it cannot see that the reload is rare)
#pragma unroll
                for (int kc = 0; kc < KS * 4; kc++)
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(afr[a][kc]) : "v"(wr + kc * 32) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int a = 0; a < TC; a++)
#pragma unroll
                for (int kc = 0; kc < KS * 4; kc++) asm volatile("" : "+v"(afr[a][kc]));
        }
        f32x16 acc[TC][TP];
#pragma unroll
        for (int a = 0; a < TC; a++)
#pragma unroll
            for (int b = 0; b < TP; b++)
#pragma unroll
                for (int k = 0; k < 16; k++) acc[a][b][k] = 0.f;
        WS_STAMP();      // tile start (weights resident)

        static_for<0, H>([&](auto hc) {
            constexpr int h = decltype(hc)::value;
            // wait for half-K stage (kk, h): at most n younger vector-memory instructions may be pending
            if (kk * H + h + NS - 2 >= total) vmcnt_le<0>();             // tail: the cursor has run dry
            else if (kk * H + h < NS - 1) vmcnt_le<PPS * (NS - 2)>();    // window reaches into the prologue: ring pieces only
            else vmcnt_le<ws_after(h, H, RES, RB)>();
            lds_barrier();
#ifdef DAFNE_WS_TIMING_ITER
            WS_STAMP();
#endif
            issue_next();
            __builtin_amdgcn_sched_barrier(0);
            const char* sb = ring + slot * STAGE;
            slot = slot == NS - 1 ? 0 : slot + 1;
#pragma unroll
            for (int k2 = 0; k2 < CPS; k2++) {
                bf16x8 bfr[TP];
#pragma unroll
                for (int b = 0; b < TP; b++) bfr[b] = *(const bf16x8*)(sb + (brow0 + b * 32) * RB + hroff[k2]);
#pragma unroll
                for (int a = 0; a < TC; a++)
#pragma unroll
                    for (int b = 0; b < TP; b++)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[a][CPS * h + k2], bfr[b], acc[a][b], 0, 0, 0);
            }
        });

        WS_STAMP();      // K loop done
        // ------------------------------------------------------------ epilogue
        if (RES) {
            // this tile's residual registers have landed: after their loads came the previous tile's 8
            // stores (not for the first tile) and this tile's 2H ring pieces
            if ((kk + 1) * H + NS - 2 >= total) vmcnt_le<0>();
            else if (kk == 0) vmcnt_le<PPS * H>();
            else vmcnt_le<PPS * H + 8>();
#pragma unroll
            for (int i = 0; i < 8; i++) {
                asm volatile("" : "+v"(rr[i]));
                const int p = (tid >> 4) + 16 * i;
                const unsigned ad = resb_off + (unsigned)(p * 256 + (((tid & 15) ^ (p & 15)) * 16));
                asm volatile("ds_write_b128 %0, %1" ::"v"(ad), "v"(rr[i]) : "memory");
            }
            // next tile's residual (the last tile re-reads its own: the instruction count stays fixed)
            fetch_residual(kk + 1 < my_tiles ? kk + 1 : kk);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            lds_barrier();     // the residual tile is complete in LDS
        }
        WS_STAMP();      // residual tile in LDS
        const unsigned rmask = RES ? 0xffffffffu : 0u;
        const float lo = relu ? 0.f : -__builtin_huge_valf();
        const float* bc = P.bias + t.nt * 128 + wc * TC * 32;
#pragma unroll
        for (int a = 0; a < TC; a++) {
            f32x4 bl[4], bh[4];
            asm volatile(
                "s_load_dwordx4 %0, %8, 0x0\n\ts_load_dwordx4 %1, %8, 0x10\n\t"
                "s_load_dwordx4 %2, %8, 0x20\n\ts_load_dwordx4 %3, %8, 0x30\n\t"
                "s_load_dwordx4 %4, %8, 0x40\n\ts_load_dwordx4 %5, %8, 0x50\n\t"
                "s_load_dwordx4 %6, %8, 0x60\n\ts_load_dwordx4 %7, %8, 0x70\n\ts_waitcnt lgkmcnt(0)"
                : "=&s"(bl[0]), "=&s"(bh[0]), "=&s"(bl[1]), "=&s"(bh[1]), "=&s"(bl[2]), "=&s"(bh[2]), "=&s"(bl[3]), "=&s"(bh[3])
                : "s"(bc + a * 32)
                : "memory");
            // LDS accesses of the epilogue are inline asm (see conv_stream_kernel)
            u32x2 rc[4][TP];
            unsigned caddr[4][TP];
#pragma unroll
            for (int g = 0; g < 4; g++)
#pragma unroll
                for (int b = 0; b < TP; b++) {
                    const int px = (wp * TP + b) * 32 + frow;
                    const int chunk = (wc * TC + a) * 4 + g;            // 16-byte chunk = 8 channels
                    caddr[g][b] = resb_off + (unsigned)(px * 256 + ((chunk ^ (px & 15)) * 16) + 8 * half);
                    if (RES) asm volatile("ds_read_b64 %0, %1" : "=v"(rc[g][b]) : "v"(caddr[g][b]) : "memory");
                    else rc[g][b] = u32x2{0u, 0u};
                }
            if (RES) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int g = 0; g < 4; g++)
#pragma unroll
                    for (int b = 0; b < TP; b++) asm volatile("" : "+v"(rc[g][b]));   // uses stay behind the wait
            }
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const f32x4 bv = half ? bh[g] : bl[g];
#pragma unroll
                for (int b = 0; b < TP; b++) {
                    u32x2 r = rc[g][b];
                    r.x &= rmask;
                    r.y &= rmask;
                    const float v0 = fmaxf(acc[a][b][4 * g] + bv[0] + bf2f((unsigned short)(r.x & 0xffff)), lo);
                    const float v1 = fmaxf(acc[a][b][4 * g + 1] + bv[1] + bf2f((unsigned short)(r.x >> 16)), lo);
                    const float v2 = fmaxf(acc[a][b][4 * g + 2] + bv[2] + bf2f((unsigned short)(r.y & 0xffff)), lo);
                    const float v3 = fmaxf(acc[a][b][4 * g + 3] + bv[3] + bf2f((unsigned short)(r.y >> 16)), lo);
                    u32x2 pk;
                    pk.x = pack_bf16(v0, v1);
                    pk.y = pack_bf16(v2, v3);
                    asm volatile("ds_write_b64 %0, %1" ::"v"(caddr[g][b]), "v"(pk) : "memory");
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        lds_barrier();
        WS_STAMP();      // accumulator pass done
        {
            const int cc = tid & 15;
            const int plast = HW - 1 - t.m0;      // last valid pixel row of a ragged tile
            u32x4 ov[8];
            int pr[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                int p = (tid >> 4) + 16 * i;
                p = p < plast ? p : plast;
                pr[i] = p;
                const unsigned ad = resb_off + (unsigned)(p * 256 + ((cc ^ (p & 15)) * 16));
                asm volatile("ds_read_b128 %0, %1" : "=v"(ov[i]) : "v"(ad) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(ov[0]), "+v"(ov[1]), "+v"(ov[2]), "+v"(ov[3]), "+v"(ov[4]), "+v"(ov[5]), "+v"(ov[6]), "+v"(ov[7])
                         :
                         : "memory");
#pragma unroll
            for (int i = 0; i < 8; i++) {
                int ho, wo;
                divmod_small(t.m0 + pr[i], S.Wout, invW, ho, wo);
                const size_t opix = (size_t)((t.img * (S.Hout + 2) + ho + 1) * (S.Wout + 2) + wo + 1);
                WS_STORE((u32x4*)(S.out + (opix * P.Cout + t.nt * 128 + cc * 8) * 2), ov[i]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // the staging tile is re-written (residual DMA) right after the next tile's first barrier:
        // every wave has finished its staging reads by then (they precede its stores above)
    }
}


// ---------------------------------------------------------------------------------------
// GN_FINALIZE (patch kernels): the GroupNorm statistics of the output are finalised by the LAST tile of every
// (segment, image) to finish, instead of by a separate launch (gn_finalize_kernel, dense_ops.hip: 36 launches per
// step of the R101 head, each a serialisation point between two tower convolutions).
//   * every tile publishes its 32 (sum, sum-sq) pairs as 8-byte agent-scope atomic stores (write-through, never left
//     in an L2 another XCD cannot see), wave 0 drains them (`s_waitcnt vmcnt(0)`: the stores are its own) and takes a
//     ticket on the image's counter with a relaxed agent-scope fetch-add;
//   * the tile that draws tiles_per_img - 1 reads all partials of the image back with 8-byte agent-scope atomic
//     loads (both sides of the hand-off are device-scope atomics: valid for any placement of the tiles on XCDs /
//     CUs, cdna_hip_programming.md G16) and reduces them IN gn_finalize_kernel's ORDER -- 32 slices of tiles
//     sl, sl + 32, .., then the slices in order -- so mean / rstd are bit-identical to the separate kernel and do not
//     depend on which tile came last; it then resets the counter for the next launch.
// `scratch`: >= 32 * 32 * 2 floats of LDS nobody else touches any more; `flag`: one LDS int.  All 512 threads call.
__device__ __forceinline__ void gn_fused_finalize(const ConvDev& P, const SegDev& S, int si, int img, int mt,
                                                  const float (&sq)[2], bool writer, int group, float* scratch, int* flag) {
    const int tid = threadIdx.x;
    const int G = P.Cout / 8;
    typedef unsigned long long u64a;
    if (writer) {
        const u64a v = ((u64a)__float_as_uint(sq[1]) << 32) | (u64a)__float_as_uint(sq[0]);
        __hip_atomic_store((u64a*)(P.gn_partial + ((size_t)mt * G + group) * 2), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid < 64) {                                    // wave 0 holds every writer lane
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) {
            const int old = __hip_atomic_fetch_add(P.gn_counters + si * P.N + img, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flag = old == S.tiles_per_img - 1;
        }
    }
    __syncthreads();
    if (!*flag) return;
    const int t0 = S.tile0 + img * S.tiles_per_img;
    const int g = tid & 31;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int sl = (tid >> 5) + 16 * k;            // 512 threads cover gn_finalize_kernel's 32 slices in two rounds
        float sum = 0.f, sqs = 0.f;
        if (g < G)
            for (int t = sl; t < S.tiles_per_img; t += 32) {
                const u64a v = __hip_atomic_load((const u64a*)(P.gn_partial + ((size_t)(t0 + t) * G + g) * 2), __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
                sum += __uint_as_float((unsigned)v);
                sqs += __uint_as_float((unsigned)(v >> 32));
            }
        scratch[(sl * 32 + g) * 2 + 0] = sum;
        scratch[(sl * 32 + g) * 2 + 1] = sqs;
    }
    __syncthreads();
    if (tid < 32 && tid < G) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int k = 0; k < 32; k++) {
            a += scratch[(k * 32 + tid) * 2 + 0];
            b += scratch[(k * 32 + tid) * 2 + 1];
        }
        const float cnt = (float)(S.Hout * S.Wout * 8);
        const float mean = a / cnt;
        float var = b / cnt - mean * mean;
        var = var > 0.f ? var : 0.f;
        float* o = P.gn_stats_out + (((size_t)si * P.N + img) * G + tid) * 2;
        o[0] = mean;
        o[1] = rsqrtf(var + P.gn_eps);
    }
    if (tid == 0) __hip_atomic_store(P.gn_counters + si * P.N + img, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------
// 3x3 stride-1 convolution fed from an LDS-staged input PATCH (im2col inside LDS).
//
// The implicit-GEMM kernel above re-fetches the pixel operand once per tap: 9 x 32 KiB of
// L2->LDS traffic per 64-channel slab and tile.  Here the output tile is a 2-D block of 8 rows x
// 32 columns, and the (8+2) x (32+2) input pixels it needs are DMA'd ONCE per slab into LDS
// (43 KiB, double-buffered across slabs); the nine taps read their B fragments straight out of
// that patch at a tap-dependent row offset.  Per K step the L2->LDS volume drops from 64 KiB to
// 32 KiB (weights) + 4.8 KiB (patch share), and the LDS-DMA write traffic with it -- this tile
// shape sits at the L2->CU ingest and LDS bandwidth limits otherwise.
//   * patch row p = (py, px) holds 64 channels (128 B), XOR-swizzled by (px>>1)&7 on the SOURCE side
//     of the DMA: a B-fragment read (32 consecutive px of one patch line) is bank-conflict free;
//   * weights stream exactly like in the 8-wave loop above (half-K pieces, counted vmcnt, two wave
//     groups one phase apart), one piece per wave and phase; the 6 patch pieces of the NEXT slab
//     ride along in taps 0..5 and are added to the allowed-outstanding count of the waits;
//   * GN_INPUT: the input is the RAW output of the previous tower convolution; its GroupNorm + ReLU
//     (dafne.py:330-344) is applied to the patch in LDS, once per slab, by the wave that loaded the
//     piece (halo / out-of-image pixels stay zero: padding follows the activation).  This removes the
//     separate normalisation pass (read + write of the whole map) between tower convolutions.
constexpr int kPH = 8, kPW = 32;
constexpr int kPCols = kPW + 2;
constexpr int kPRows = (kPH + 2) * kPCols;      // 340 input pixels
constexpr int kPPieces = (kPRows + 7) / 8;      // 43 DMA pieces of 8 pixels x 128 B
constexpr int kPBuf = kPPieces * 1024;
constexpr int kPAHalf = 256 * 64;               // weights of half a K step: 256 rows x 64 B
constexpr int kPAStage = 2 * kPAHalf;
constexpr int kPOffPatch = 2 * kPAStage;
constexpr int kPOffTab = kPOffPatch + 2 * kPBuf;
constexpr int kPTabMaxC = 512;
constexpr int kPSmem = kPOffTab + kPTabMaxC * 9;   // stats [C/8][2] + gamma [C] + beta [C]

template <bool GNIN>
__global__ void __launch_bounds__(512) conv3x3_patch_kernel(ConvDev P) {
    constexpr int NW = 8, WP = 2, TC = 2, TP = 4, BN = 256, BM = 256, NT = 512;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave / WP, wp = wave % WP;
    const int grp = wave >> 2;
    const int frow = lane & 31, half = lane >> 5;

    const int T = P.mtiles * P.ntiles;
    const int bid = xcd_remap(blockIdx.x, T);
    const int nt = bid % P.ntiles;
    const int mt = bid / P.ntiles;
    int si = 0;
#pragma unroll
    for (int k = 1; k < kMaxSegs; k++)
        if (k < P.n_segs && mt >= P.seg[k].tile0) si = k;
    const SegDev& S = P.seg[si];
    const int tloc = mt - S.tile0;
    const int img = tloc / S.tiles_per_img;
    const int tt = tloc - img * S.tiles_per_img;
    const int ty = tt / S.tiles_x, tx = tt - ty * S.tiles_x;
    const int Y0 = ty * kPH, X0 = tx * kPW;
    const int H = S.Hout, W = S.Wout, Hp = H + 2, Wp = W + 2;
    const int K = P.ksteps;                 // 9 * Cin/64
    const int nslab = P.Cin / kBK;

    // ---- GroupNorm table of this image (input side) -> LDS, before any DMA is in flight
    float* tab_stats = (float*)(lds + kPOffTab);
    float* tab_gamma = tab_stats + P.Cin / 4;
    float* tab_beta = tab_gamma + P.Cin;
    if (GNIN) {
        const float* st = P.in_stats + ((size_t)si * P.N + img) * (P.Cin / 8) * 2;
        for (int k = tid; k < P.Cin / 4; k += NT) tab_stats[k] = st[k];
        for (int k = tid; k < P.Cin; k += NT) {
            tab_gamma[k] = P.in_gamma[k];
            tab_beta[k] = P.in_beta[k];
        }
        __syncthreads();
    }
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;

    // ---- per-lane source offsets ------------------------------------------------------------
    unsigned hofs[2];                      // weight rows of this wave's two pieces per K half
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int r = (i * NW + wave) * 16 + (lane >> 2);
        const int q = (lane & 3) ^ ((r >> 2) & 3);
        hofs[i] = (unsigned)(nt * BN + r) * (unsigned)P.kbytes + (unsigned)q * 16u;
    }
    unsigned pofs[6];                      // patch pixels of this wave's six pieces per slab
    int ppi[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        int pi = i * NW + wave;
        if (pi >= kPPieces) pi -= NW;      // pieces 43..47: the wave re-fetches its previous piece
        ppi[i] = pi;
        int r = pi * 8 + (lane >> 3);
        r = r < kPRows ? r : kPRows - 1;
        const int py = r / kPCols, px = r - py * kPCols;
        int gy = Y0 + py, gx = X0 + px;
        gy = gy < Hp ? gy : Hp - 1;
        gx = gx < Wp ? gx : Wp - 1;
        const int q = (lane & 7) ^ ((px >> 1) & 7);
        pofs[i] = ((unsigned)(img * Hp + gy) * (unsigned)Wp + (unsigned)gx) * (unsigned)(P.Cin * 2) + (unsigned)q * 16u;
    }
    // B-fragment offsets inside a patch line: pixel column frow+kw, chunk (2ks+half) ^ swizzle
    unsigned boff[3][4];
#pragma unroll
    for (int kw = 0; kw < 3; kw++)
#pragma unroll
        for (int ks = 0; ks < 4; ks++)
            boff[kw][ks] = (unsigned)(frow + kw) * 128u + (unsigned)(((2 * ks + half) ^ (((frow + kw) >> 1) & 7)) * 16);
    const int fsw4 = (frow >> 2) & 3;
    unsigned hroff[2];
#pragma unroll
    for (int k2 = 0; k2 < 2; k2++) hroff[k2] = (unsigned)frow * 64u + (unsigned)(((2 * k2 + half) ^ fsw4) * 16);
    const int arow0 = wc * TC * 32;

    // Weight half-steps are walked in the order u = ((slab*3 + kw)*2 + half)*3 + kh: the three kh taps of
    // one (kw, K half) are consecutive, so the B fragments of 6 patch lines are read ONCE and serve all
    // three (fragment reads per MFMA: 0.75 -> 0.5; this loop is LDS-read bound).  4 half-step buffers.
    const int U = 2 * K;                    // half-steps in total
    auto piece_a = [&](int u, int i) {
        if (u >= U) return;
        const int kh = u % 3, g = u / 3;
        const int h = g & 1, g2 = g >> 1;
        const int kw = g2 % 3, sl = g2 / 3;
        const unsigned koff = (unsigned)((sl * 9 + kh * 3 + kw) * kRowBytes + h * 64);
        char* dst = lds + (u & 3) * kPAHalf + (i * NW + wave) * 1024;
        __builtin_amdgcn_global_load_lds((gvoid*)(P.w + hofs[i] + koff), (lvoid*)dst, 16, 0, 0);
    };
    auto piece_p = [&](int slab, int i) {
        char* dst = lds + kPOffPatch + (slab & 1) * kPBuf + ppi[i] * 1024;
        __builtin_amdgcn_global_load_lds((gvoid*)(S.in + pofs[i] + (unsigned)slab * 128u), (lvoid*)dst, 16, 0, 0);
    };
    // GroupNorm + ReLU of one landed patch piece, in place (inline-asm LDS ops: see conv_ws_kernel).
    // A lane always handles LOGICAL chunk lane&7 (8 channels = one group) of pixel row lane>>3 of the piece --
    // wherever the swizzle put it -- so mean / rstd / gamma / beta are per-lane constants of the slab
    // of the slab.
    // The constants are re-read from the LDS table in front of every pair of pieces (5 LDS reads): they
    // are live only in the head of a read phase, where the A/B fragment registers are dead.
    auto gn_pieces = [&](int slab, int i0, int n) {
        const int ch = slab * kBK + (lane & 7) * 8;
        const unsigned ts = lds_base + (unsigned)(kPOffTab + (ch >> 3) * 8);
        const unsigned tg = lds_base + (unsigned)(kPOffTab + P.Cin + ch * 4);
        const unsigned tb = tg + (unsigned)P.Cin * 4u;
        u32x2 ms;
        f32x4 g0, g1, b0, b1;
        asm volatile("ds_read_b64 %0, %5\n\tds_read_b128 %1, %6\n\tds_read_b128 %2, %6 offset:16\n\t"
                     "ds_read_b128 %3, %7\n\tds_read_b128 %4, %7 offset:16\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(ms), "=&v"(g0), "=&v"(g1), "=&v"(b0), "=&v"(b1)
                     : "v"(ts), "v"(tg), "v"(tb)
                     : "memory");
        const float gmean = __uint_as_float(ms.x), grstd = __uint_as_float(ms.y);
        const float gam[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
        const float bet[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        for (int i = i0; i < i0 + n; i++) {
            if (i == 5 && 5 * NW + wave >= kPPieces) continue;        // duplicate of piece i = 4
            const int pi = ppi[i];
            const int r = pi * 8 + (lane >> 3);
            const int py = r / kPCols, px = r - py * kPCols;
            const int gy = Y0 + py, gx = X0 + px;
            const bool inside = gy >= 1 && gy <= H && gx >= 1 && gx <= W && r < kPRows;
            const int phys = (lane & 7) ^ ((px >> 1) & 7);
            const unsigned ad = lds_base + (unsigned)(kPOffPatch + (slab & 1) * kPBuf + pi * 1024 + (lane >> 3) * 128 + phys * 16);
            u32x4 v;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(ad) : "memory");
            const unsigned u[4] = {v.x, v.y, v.z, v.w};
            float y[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const float x = bf2f((unsigned short)(k & 1 ? u[k >> 1] >> 16 : u[k >> 1] & 0xffff));
                y[k] = fmaxf((x - gmean) * grstd * gam[k] + bet[k], 0.f);      // expression of gn_apply_kernel
            }
            u32x4 o;
            o.x = inside ? pack_bf16(y[0], y[1]) : 0u;
            o.y = inside ? pack_bf16(y[2], y[3]) : 0u;
            o.z = inside ? pack_bf16(y[4], y[5]) : 0u;
            o.w = inside ? pack_bf16(y[6], y[7]) : 0u;
            asm volatile("ds_write_b128 %0, %1" ::"v"(ad), "v"(o) : "memory");
        }
    };

    // The same conversion for ONE piece with a single LDS round trip: the piece and the constants are read together
    // and waited for once (gn_pieces pays the latency twice: constants, then piece).
    auto gn_skip = [&](int i) { return i == 5 && 5 * NW + wave >= kPPieces; };      // duplicate of piece i = 4
    auto gn_one = [&](int slab, int i) {
        if (gn_skip(i)) return;
        const int ch = slab * kBK + (lane & 7) * 8;
        const unsigned ts = lds_base + (unsigned)(kPOffTab + (ch >> 3) * 8);
        const unsigned tg = lds_base + (unsigned)(kPOffTab + P.Cin + ch * 4);
        const unsigned tb = tg + (unsigned)P.Cin * 4u;
        u32x4 gq_v;
        u32x2 gq_ms;
        f32x4 gq_g0, gq_g1, gq_b0, gq_b1;
        {
            const int pi = ppi[i];
            const int r = pi * 8 + (lane >> 3);
            const int py = r / kPCols, px = r - py * kPCols;
            const int phys = (lane & 7) ^ ((px >> 1) & 7);
            const unsigned ad = lds_base + (unsigned)(kPOffPatch + (slab & 1) * kPBuf + pi * 1024 + (lane >> 3) * 128 + phys * 16);
            asm volatile("ds_read_b128 %0, %6\n\tds_read_b64 %1, %7\n\tds_read_b128 %2, %8\n\tds_read_b128 %3, %8 offset:16\n\t"
                         "ds_read_b128 %4, %9\n\tds_read_b128 %5, %9 offset:16\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(gq_v), "=&v"(gq_ms), "=&v"(gq_g0), "=&v"(gq_g1), "=&v"(gq_b0), "=&v"(gq_b1)
                         : "v"(ad), "v"(ts), "v"(tg), "v"(tb)
                         : "memory");
        }
        const float gmean = __uint_as_float(gq_ms.x), grstd = __uint_as_float(gq_ms.y);
        const float gam[8] = {gq_g0[0], gq_g0[1], gq_g0[2], gq_g0[3], gq_g1[0], gq_g1[1], gq_g1[2], gq_g1[3]};
        const float bet[8] = {gq_b0[0], gq_b0[1], gq_b0[2], gq_b0[3], gq_b1[0], gq_b1[1], gq_b1[2], gq_b1[3]};
        const int pi = ppi[i];
        const int r = pi * 8 + (lane >> 3);
        const int py = r / kPCols, px = r - py * kPCols;
        const int gy = Y0 + py, gx = X0 + px;
        const bool inside = gy >= 1 && gy <= H && gx >= 1 && gx <= W && r < kPRows;
        const int phys = (lane & 7) ^ ((px >> 1) & 7);
        const unsigned ad = lds_base + (unsigned)(kPOffPatch + (slab & 1) * kPBuf + pi * 1024 + (lane >> 3) * 128 + phys * 16);
        const unsigned u[4] = {gq_v.x, gq_v.y, gq_v.z, gq_v.w};
        float y[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float x = bf2f((unsigned short)(k & 1 ? u[k >> 1] >> 16 : u[k >> 1] & 0xffff));
            y[k] = fmaxf((x - gmean) * grstd * gam[k] + bet[k], 0.f);      // expression of gn_apply_kernel
        }
        u32x4 o;
        o.x = inside ? pack_bf16(y[0], y[1]) : 0u;
        o.y = inside ? pack_bf16(y[2], y[3]) : 0u;
        o.z = inside ? pack_bf16(y[4], y[5]) : 0u;
        o.w = inside ? pack_bf16(y[6], y[7]) : 0u;
        asm volatile("ds_write_b128 %0, %1" ::"v"(ad), "v"(o) : "memory");
    };

    f32x16 acc[TC][TP];
#pragma unroll
    for (int a = 0; a < TC; a++)
#pragma unroll
        for (int b = 0; b < TP; b++)
#pragma unroll
            for (int k = 0; k < 16; k++) acc[a][b][k] = 0.f;
    bf16x8 af[2][TC], bfr[2][TP + 2];
    auto read_a = [&](int u) {
        const char* sb = lds + (u & 3) * kPAHalf;
#pragma unroll
        for (int k2 = 0; k2 < 2; k2++)
#pragma unroll
            for (int a = 0; a < TC; a++) af[k2][a] = *(const bf16x8*)(sb + (arow0 + a * 32) * 64 + hroff[k2]);
    };
    auto read_b6 = [&](int slab, int kw, int h) {      // patch lines wp*4 .. wp*4+5 at column offset kw
        const char* pb = lds + kPOffPatch + (slab & 1) * kPBuf;
#pragma unroll
        for (int k2 = 0; k2 < 2; k2++)
#pragma unroll
            for (int rr = 0; rr < TP + 2; rr++)
                bfr[k2][rr] = *(const bf16x8*)(pb + ((wp * TP + rr) * kPCols) * 128 + boff[kw][2 * h + k2]);
    };
    auto mma16 = [&](int kh) {
#pragma unroll
        for (int k2 = 0; k2 < 2; k2++)
#pragma unroll
            for (int a = 0; a < TC; a++)
#pragma unroll
                for (int b = 0; b < TP; b++)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[k2][a], bfr[k2][b + kh], acc[a][b], 0, 0, 0);
    };
    // one weight piece per wave and phase; odd phases wait: the 4 youngest weight pieces plus the patch pieces
    // issued in this and the previous half-step (np = 0, 1, 2) may stay in flight across the barrier
    auto phase_end = [&](bool odd, int t, int np) {
        if (odd) {
            if (t <= 4 * K - 7) {
                if (np == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else if (np == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: patch of slab 0, weights of half-steps 0..2 ------------------------------------------
#pragma unroll
    for (int i = 0; i < 6; i++) piece_p(0, i);
    piece_a(0, 0); piece_a(0, 1); piece_a(1, 0); piece_a(1, 1); piece_a(2, 0); piece_a(2, 1);
    if (GNIN) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // this wave's patch pieces have landed
        gn_pieces(0, 0, 6);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");          // half-step 0 is complete
    phase_end(false, -1, 0);

    // Global phase t = 2 u + p (u = half-step, p = 0: fragment reads of group 0 / MFMAs of group 1,
    // p = 1: the reverse) issues weight piece p of half-step u + 3.  Patch pieces of slab+1: piece i at p = 0 of
    // half-step i of the slab (i = 0..5); a piece has landed three half-steps later (the waits let the pieces of
    // the current and the previous half-step stay in flight).  Their GroupNorm runs ONE piece per read phase, in
    // the read phases that carry no B-fragment reads (kh = 1, 2: half-steps 7, 8, 10, 11, 13, 14): two pieces in
    // one phase made that phase twice as long as the other group's MFMA phase and idled the matrix pipe.
    // G = (kw, half): half-steps 3G .. 3G+2 = kh 0..2; UU = 3G is the half-step index inside the slab.
#define PATCH_NP(UU) (more ? (((UU) <= 5 ? 1 : 0) + (((UU) >= 1 && (UU) <= 6) ? 1 : 0)) : 0)
#define PATCH_GN_ON(UU) (GNIN && more && (UU) >= 7 && (UU) <= 14 && (UU) % 3 != 0)
#define PATCH_GN_PIECE(UU) (((UU) - 7) - ((UU) - 6) / 3)
#define PATCH_HS_G0(UU, KW_, H_, KH_)                                                                        \
    {                                                                                                        \
        const int u = slab * 18 + (UU);                                                                      \
        piece_a(u + 3, 0);                                                                                   \
        if (more && (UU) <= 5) piece_p(slab + 1, (UU));                                                      \
        if (PATCH_GN_ON(UU)) gn_one(slab + 1, PATCH_GN_PIECE(UU));                                           \
        read_a(u);                                                                                           \
        if ((KH_) == 0) read_b6(slab, KW_, H_);                                                              \
        phase_end(false, 2 * u, 0);                                                                          \
        piece_a(u + 3, 1);                                                                                   \
        mma16(KH_);                                                                                          \
        phase_end(true, 2 * u + 1, PATCH_NP(UU));                                                            \
    }
    // group 1 runs one phase behind: reads in p = 1 of half-step u, MFMAs in p = 0 of half-step u + 1
#define PATCH_HS_G1(UU, KW_, H_, KH_)                                                                        \
    {                                                                                                        \
        const int u = slab * 18 + (UU);                                                                      \
        piece_a(u + 3, 1);                                                                                   \
        if (PATCH_GN_ON(UU)) gn_one(slab + 1, PATCH_GN_PIECE(UU));                                           \
        read_a(u);                                                                                           \
        if ((KH_) == 0) read_b6(slab, KW_, H_);                                                              \
        phase_end(true, 2 * u + 1, PATCH_NP(UU));                                                            \
        piece_a(u + 4, 0);                                                                                   \
        if ((UU) < 17) {                                                                                     \
            if (more && (UU) + 1 <= 5) piece_p(slab + 1, (UU) + 1);                                          \
        } else if (slab + 2 < nslab) {                                                                       \
            piece_p(slab + 2, 0);                                                                            \
        }                                                                                                    \
        mma16(KH_);                                                                                          \
        if (u + 1 < U) phase_end(false, 2 * u + 2, 0);                                                       \
    }
#define PATCH_SLAB(M)                                                                                        \
    M(0, 0, 0, 0) M(1, 0, 0, 1) M(2, 0, 0, 2) M(3, 0, 1, 0) M(4, 0, 1, 1) M(5, 0, 1, 2)                      \
    M(6, 1, 0, 0) M(7, 1, 0, 1) M(8, 1, 0, 2) M(9, 1, 1, 0) M(10, 1, 1, 1) M(11, 1, 1, 2)                    \
    M(12, 2, 0, 0) M(13, 2, 0, 1) M(14, 2, 0, 2) M(15, 2, 1, 0) M(16, 2, 1, 1) M(17, 2, 1, 2)
    if (grp == 0) {
        for (int slab = 0; slab < nslab; slab++) {
            const bool more = slab + 1 < nslab;
            PATCH_SLAB(PATCH_HS_G0)
        }
    } else {
        // phase 0 of the whole loop: this group idles one phase (issues its share of the loads only)
        piece_a(3, 0);
        if (nslab > 1) piece_p(1, 0);
        phase_end(false, 0, 0);
        for (int slab = 0; slab < nslab; slab++) {
            const bool more = slab + 1 < nslab;
            PATCH_SLAB(PATCH_HS_G1)
        }
    }
#undef PATCH_SLAB
#undef PATCH_HS_G0
#undef PATCH_HS_G1
#undef PATCH_GN_ON
#undef PATCH_GN_PIECE
#undef PATCH_NP

    // ------------------------------------------------------------ epilogue (bias, ReLU, GN sums, bf16)
    const bool relu = P.flags & DAFNE_CONV_RELU;
    const float relu_lo = relu ? 0.f : -__builtin_inff();
    const bool gn = P.flags & DAFNE_CONV_GN_STATS;
    const bool fin = gn && (P.flags & DAFNE_CONV_GN_FINALIZE);     // wave-uniform
    float fin_sq[2] = {0.f, 0.f};
    bool fin_writer = false;
    constexpr int ROWB = BN * 2 + 16;
    constexpr int CHB = BN / 8;
    char* stg = lds;
    float4 bia4[TC][4];
#pragma unroll
    for (int a = 0; a < TC; a++)
#pragma unroll
        for (int g = 0; g < 4; g++)
            bia4[a][g] = P.bias ? *(const float4*)(P.bias + nt * BN + (wc * TC + a) * 32 + 8 * g + 4 * half)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
    float gsum[TC][4], gsq[TC][4];
#pragma unroll
    for (int a = 0; a < TC; a++)
#pragma unroll
        for (int g = 0; g < 4; g++) gsum[a][g] = gsq[a][g] = 0.f;
    __syncthreads();   // every wave is done with the weight stages and the patch
#pragma unroll
    for (int b = 0; b < TP; b++) {
        const int px = (wp * TP + b) * 32 + frow;
        const bool valid = (Y0 + wp * TP + b) < H && (X0 + frow) < W;
#pragma unroll
        for (int a = 0; a < TC; a++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                // (branch-free: ReLU as a max with 0 / -inf, the GroupNorm sums added under a select -- x + 0 is exact; round 5)
                const float v0 = fmaxf(acc[a][b][4 * g] + bia4[a][g].x, relu_lo), v1 = fmaxf(acc[a][b][4 * g + 1] + bia4[a][g].y, relu_lo);
                const float v2 = fmaxf(acc[a][b][4 * g + 2] + bia4[a][g].z, relu_lo), v3 = fmaxf(acc[a][b][4 * g + 3] + bia4[a][g].w, relu_lo);
                gsum[a][g] += valid ? (v0 + v1) + (v2 + v3) : 0.f;
                gsq[a][g] += valid ? (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3) : 0.f;
                uint2 pk;
                pk.x = pack_bf16(v0, v1);
                pk.y = pack_bf16(v2, v3);
                const int co = (wc * TC + a) * 32 + 8 * g + 4 * half;
                *(uint2*)(stg + px * ROWB + co * 2) = pk;
            }
    }
    float* redb = (float*)(lds + BM * ROWB);   // [NW][TC*4][2]
    if (gn) {
#pragma unroll
        for (int a = 0; a < TC; a++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                float sv = gsum[a][g], qv = gsq[a][g];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    sv += __shfl_xor(sv, o, 64);
                    qv += __shfl_xor(qv, o, 64);
                }
                if (lane == 0) {
                    redb[(wave * TC * 4 + a * 4 + g) * 2 + 0] = sv;
                    redb[(wave * TC * 4 + a * 4 + g) * 2 + 1] = qv;
                }
            }
    }
    __syncthreads();
    if (gn && tid < BN / 8) {
        const int wcc = tid / (TC * 4), ag = tid % (TC * 4);
        float sv = 0.f, qv = 0.f;
#pragma unroll
        for (int p2 = 0; p2 < WP; p2++) {
            sv += redb[((wcc * WP + p2) * TC * 4 + ag) * 2 + 0];
            qv += redb[((wcc * WP + p2) * TC * 4 + ag) * 2 + 1];
        }
        const int group = (nt * BN) / 8 + tid;
        if (group < P.Cout / 8) {
            if (fin) {
                fin_sq[0] = sv;
                fin_sq[1] = qv;
                fin_writer = true;
            } else {
                float* o = P.gn_partial + ((size_t)mt * (P.Cout / 8) + group) * 2;
                o[0] = sv;
                o[1] = qv;
            }
        }
    }
    constexpr int CPT = BM * CHB / NT;       // 16-byte chunks per thread
#pragma unroll
    for (int i = 0; i < CPT; i++) {
        const int idx = tid + i * NT;
        const int p = idx / CHB, cc = idx - p * CHB;
        const int gy = Y0 + (p >> 5), gx = X0 + (p & 31);
        if (gy < H && gx < W) {
            const size_t opix = (size_t)(img * Hp + gy + 1) * Wp + gx + 1;
            const uint4 v = *(const uint4*)(stg + p * ROWB + cc * 16);
            *(uint4*)(S.out + (opix * P.Cout + nt * BN + cc * 8) * 2) = v;
        }
    }
    if (fin) {
        __syncthreads();                     // the staging tile has been read: its LDS is free for the reduction
        gn_fused_finalize(P, S, si, img, mt, fin_sq, fin_writer, tid, (float*)lds, (int*)(lds + 32 * 32 * 2 * 4));
    }
}

// ---------------------------------------------------------------------------------------
// 3x3 stride-1 convolution, 256 input channels, with the WHOLE input patch resident in LDS and the weights streamed
// L2 -> registers (round 3; the structure of conv_bneck.hip's phase A as a stand-alone layer): head towers, FPN outputs.
//
// conv3x3_patch_kernel above stages weights AND pixels through LDS in half-K pieces: 36 K steps x 4 phases, a barrier
// per phase, two wave groups that must stay exactly one phase apart -- measured 935-1000 TFLOP/s, the matrix pipe busy
// ~50 % of the cycles the chip runs.  Here a workgroup (8 waves, one per CU) owns an 8 x 32 pixel tile and 256 output
// channels; the input patch sits in LDS in 64-channel slabs ([pixel][128 B], chunk XOR (patch column >> 1) & 7: conflict-free
// ds_read_b128 at every tap offset, and -- the row pitch being even -- independent of the patch row, so a fragment address is ONE
// per-lane register per tap column plus an immediate); a wave owns 32 output channels and all 256 pixels, its weight fragments
// (fragment-major packing, one coalesced 1-KiB load per k16 step, ring of 8) go straight to registers: every fragment feeds
// EIGHT MFMAs and is fetched by exactly one wave, and the 36 k16 steps of a slab need no barrier for their operands.  K order =
// (64-channel slab, kh, kw, k16 step) = conv_igemm_kernel's: bit-identical outputs.
//
// Round 6: 8 x 32 tiles instead of 4 x 32 (rounds 3-5).  The package runs these layers at its power cap, so a launch's time is its
// energy; per step of the benchmark the 4 x 32 form pulled 26 GB of weight fragments L2 -> registers (1.18 MB per 128 pixels) at
// 16 pJ per byte and re-read 31 % of its input as halo (6 x 34 / 4 x 32) at 132 pJ per HBM byte (scripts/l2_probe.py,
// bench.py energy_ledger: profiles/NOTES_r06.md).  A tile of 256 pixels halves the weight bytes per flop and takes the halo to
// 10 x 34 / 8 x 32 = 1.33.  Its whole patch would be 174 KB: the slabs go through a RING OF THREE 43-KB buffers instead -- slab g
// (counted across tiles) lives in buffer g mod 3, is requested by DMA while slab g - 2 computes (one piece per wave and step, steps
// 1..6 behind the barrier that retires slab g - 3), normalised in place while slab g - 1 computes (GN_INPUT), and published by the
// barrier in front of its own first step.  The accumulators are 128 registers per wave; the B fragments of a step are read in two
// halves of four (two register sets, the second half requested under the first half's MFMAs).
//
// The kernel is PERSISTENT and software-pipelined across tiles (workgroup p of G runs tiles p, p + G, ...):
//   * slabs 0 and 1 of the NEXT tile are requested during slabs 2 and 3 of the current one (their ring buffers follow from the
//     global slab count), slabs 2 and 3 of a tile during its own slabs 0 and 1;
//   * GN_INPUT: a wave normalises the pieces IT loaded (GroupNorm + ReLU in place, out-of-image pixels stay zero) one slab period
//     after issuing them, steps 8..13 (waves 0..3) / 14..19 (waves 4..7: the two waves of a SIMD six steps apart) of the period --
//     under the other waves' MFMAs; statistics of the next tile's image by one 256-B DMA piece at step 56 (double-buffered), its
//     scale / shift table y = max(a x + b, 0), a = rstd gamma, b = beta - mean a at step 70 (256 threads), published by the barrier
//     of step 72, first used at step 116;
//   * the epilogue is short and asynchronous: at the end of a tile bias / ReLU / GroupNorm sums are applied in the
//     accumulator layout (a wave holds all 256 pixels of its 32 channels: no cross-wave reduction), the bf16 values are
//     paired across the two half-waves with v_permlane32_swap (16 contiguous bytes per lane) and stored straight from
//     registers -- 16 stores per lane that drain under the next tile's first steps (no LDS staging, no barrier, nothing
//     held in registers across tiles).  The GroupNorm sums are reduced with DPP in the next tile's steps 1..8 and
//     published by its barriers; with GN_FINALIZE the arrival ticket is taken in step 50 (no drain: the in-order vmcnt
//     waits of the weight ring already cover the partial-sum stores of step 38.. of wave 0), the rare last-tile reduction
//     runs behind the barrier of step 72;
//   * every s_waitcnt vmcnt(n) is exact (rp_wait): stores are never predicated -- rows of out-of-image pixels go to a
//     caller-provided dump area.
constexpr int kRH = 8, kRW = 32, kRPx = kRH * kRW;
constexpr int kRCols = kRW + 2, kRRows = kRH + 2;
constexpr int kRPieces = (kRRows * kRCols + 7) / 8;     // 43 pieces of 8 px x 128 B per slab
constexpr int kRSlab = kRPieces * 1024;                 // 44 032 B
constexpr int kRBufs = 3;                               // ring of slab buffers
constexpr int kRPP = (kRPieces + 7) / 8;                // 6 DMA pieces per wave and slab
constexpr int kRFr = kRH;                               // B fragments (32 pixels each) per k16 step
constexpr int kRCin = 256, kRSteps = 9 * kRCin / 16;    // 144 k16 steps
constexpr int kRRing = 6;                               // k16 steps of A fragments in flight per wave (a step is 8 MFMAs)
constexpr int kRMaxCout = 1024;
constexpr int kROffStat = kRBufs * kRSlab;              // GN_INPUT statistics of (segment, image): 2 x [32][2] fp32
constexpr int kROffGB = kROffStat + 512;                // per group: gamma [256], beta [256] fp32
constexpr int kROffBias = kROffGB + 2 * 2048;           // per group: bias fp32 [Cout <= 1024]
constexpr int kROffRed = kROffBias + 2 * kRMaxCout * 4; // [32 groups][2] fp32, finalize flag at +256
constexpr int kROffFin = kROffRed + 512;                // GN_FINALIZE reduction scratch: 32 x 32 x 2 fp32
constexpr int kROffAB = kROffFin + 32 * 32 * 2 * 4;     // GN_INPUT: per tile parity, a[256] | b[256] fp32 of the tile's (layer, image)
constexpr int kRSmem = kROffAB + 2 * 2048;
static_assert(kRSmem <= 160 * 1024, "LDS budget");
constexpr int kRDumpBytes = 128 * 1024;                 // 512 threads x 16 rows of 16 B

// Vector-memory program order of a wave inside a tile (steady state; A(s + 8) of steps >= 136 are the next tile's first):
//   step s: [wait A(s)] MFMAs | A(s + 8) | rp_post(s) more operations:
//     steps 36 sl + 1 .. 36 sl + 6: one patch piece of the slab two ahead (sl = 0, 1: this tile's slabs 2, 3; sl = 2, 3: the next
//     tile's slabs 0, 1); 56 (GN_INPUT): the statistics piece of the next tile's image; 143: the tile's 16 row stores.
// rp_wait(j) = operations issued after A(j) and before the wait for it (vmcnt retires in order).  Steps 0..7 look back
// into the previous tile; the FIRST tile of a workgroup has the prologue there instead (12 patch pieces, A(0..7), 16 dummy stores:
// the same queue).
#ifndef DAFNE_RP_BAR
#define DAFNE_RP_BAR 12
#endif
constexpr int kRBar = DAFNE_RP_BAR;       // the one barrier of a slab period sits behind this step of it (see the kernel)
constexpr int kRStatStep = 56;            // GN_INPUT: the next tile's statistics piece, early enough for the a / b table of step 70
constexpr int kRTabStep = 70;
constexpr int rp_post(int s, bool gnin) {
    int n = 0;
    if (gnin && s == kRStatStep) n += 1;
    const int t = s % 36;
    if (t >= kRBar + 1 && t <= kRBar + kRPP) n += 1;
    if (s == kRSteps - 1) n += 2 * kRFr;
    return n;
}
// (ONE sequence for every tile -- round 5: a first-tile special case `if (first) wait(kF) else wait(kN)` on the ring register made
// the compiler put a COPY of the register in front of the steady-state wait.  The prologue issues 16 dummy dword stores to the dump
// area behind A(0..7): same queue, same counts, no branch.)
constexpr int rp_wait(int j, bool gnin) {
    int n = rp_post((j - kRRing + kRSteps) % kRSteps, gnin);
    for (int s = j - kRRing + 1; s < j; s++) n += 1 + rp_post((s + kRSteps) % kRSteps, gnin);
    return n;
}
static_assert(kRRing != 6 || kRBar != 12 || (rp_wait(0, false) == 21 && rp_wait(2, false) == 21 && rp_wait(5, false) == 21 && rp_wait(6, false) == 5 &&
              rp_wait(13, false) == 5 && rp_wait(14, false) == 6 && rp_wait(19, false) == 11 && rp_wait(20, false) == 10 && rp_wait(24, false) == 6 &&
              rp_wait(25, false) == 5 && rp_wait(52, false) == 8 && rp_wait(kRStatStep + 6, true) == 6 && rp_wait(kRStatStep + 6, false) == 5 &&
              rp_wait(143, true) == 5), "vmcnt bookkeeping");
static_assert(kRBar >= 2 * kRPP && 36 - kRBar - 1 >= kRRing + 1, "a period's requests are covered by the wave's own waits before its normalisation starts");

template <int J>
__device__ __forceinline__ void rp_load(bf16x8 (&ar)[kRRing], const char* wf_cur, const char* wf_nxt, unsigned voff) {
    const char* sb = (J < kRSteps ? wf_cur : wf_nxt) + (size_t)(J % kRSteps) * 1024;
    // "+v": the destination is loop-carried and stays ONE register from the zero-initialisation on.  With an output-only
    // operand every load defines a new value, which the compiler may move between registers (at the loop's back edge, inside
    // the plain-C++ epilogue) while the load is still in flight: the copy takes the OLD bits.  Round 5 found it as run-to-run
    // different detections in the deferred layout (tests/test_gpu_headline.py::test_headline_timed_layout_vs_oracle) once the
    // barrier at the end of a tile -- which had been giving the eight loads time to land -- was gone.
    asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(ar[J % kRRing]) : "v"(voff), "s"(sb) : "memory");
}
template <int J, bool GNIN>
__device__ __forceinline__ void rp_wait_for(bf16x8 (&ar)[kRRing]) {
    constexpr int kN = rp_wait(J, GNIN);
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(ar[J % kRRing]) : "n"(kN) : "memory");
}
// ---- the 16x16x32 form (round 6; conv3x3_rp_kernel<GNIN, true>, flag DAFNE_CONV_FRAG16): the same tile, slabs, ring and schedule with
// v_mfma_f32_16x16x32_bf16 -- 6-8 % fewer joules per flop on this package (half the accumulator traffic; profiles/NOTES_r06.md).
// A k32 group = the k16 steps 2m, 2m + 1 of the same (slab, kh, kw).  Fragment F(2m + cb) = output channels 16 cb .. 16 cb + 15 of the
// wave x the 32 k of group m (lane: channel lane & 15, k 8 (lane >> 4) .. + 8); a B fragment = 16 pixels x those 32 k (lane: pixel
// lane & 15, the same k).  Step 2m + h multiplies BOTH fragments of its group with tile rows 4h .. 4h + 3 (8 B fragments: row, column
// half), so a fragment lives for two steps: it is requested FIVE steps ahead into the slot of the fragment that died one step ago
// (F(j + 5) at the end of step j into slot (j + 5) % 6), and the even step of a pair waits for the pair's younger fragment.
constexpr int kRDist16 = kRRing - 1;
constexpr int rp_wait16(int j, bool gnin) {            // even j: operations issued after F(j + 1) (end of step j + 1 - kRDist16) and before this wait
    int n = rp_post((j + 1 - kRDist16 + kRSteps) % kRSteps, gnin);
    for (int s = j + 2 - kRDist16; s < j; s++) n += 1 + rp_post((s + kRSteps) % kRSteps, gnin);
    return n;
}
static_assert(kRRing != 6 || kRBar != 12 || (rp_wait16(0, false) == 19 && rp_wait16(2, false) == 19 && rp_wait16(4, false) == 3 && rp_wait16(14, false) == 4 &&
              rp_wait16(20, false) == 6), "vmcnt bookkeeping (16x16x32 form)");
template <int J, bool GNIN>
__device__ __forceinline__ void rp_wait16_for(bf16x8 (&ar)[kRRing]) {
    static_assert((J & 1) == 0, "the even step of a pair waits");
    constexpr int kN = rp_wait16(J, GNIN);
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(ar[J % kRRing]), "+v"(ar[(J + 1) % kRRing]) : "n"(kN) : "memory");
}
// four of the eight B fragments of step J in the 16x16x32 form: tile rows 4 (J & 1) + 2 HF, + 1 (patch rows + kh) at tap column kw, both
// column halves (16 pixels = 2048 B apart: the swizzle of column q + 16 is that of q), the 32 channels 32 kc2 .. of the slab
template <int J, int HF>
__device__ __forceinline__ void rp_bread16(bf16x8 (&b)[4], const unsigned (&pb)[3], unsigned sbase) {
    constexpr int t = J % 36, kh = t / 12, kw = (t >> 2) % 3, kc2 = (t >> 1) & 1;
    unsigned pq = pb[kw];
    asm volatile("" : "+v"(pq));
    const unsigned ad = sbase + (pq ^ (unsigned)(kc2 << 6));
    constexpr int r0 = kh + 4 * (J & 1) + 2 * HF;
    constexpr int o0 = r0 * kRCols * 128, o1 = o0 + 2048, o2 = (r0 + 1) * kRCols * 128, o3 = o2 + 2048;
    static_assert(o3 <= 65535, "ds_read immediate");
    asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"
                 : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3])
                 : "v"(ad), "n"(o0), "n"(o1), "n"(o2), "n"(o3)
                 : "memory");
}

template <int I, int N, class F>
__device__ __forceinline__ void rp_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        rp_static_for<I + 1, N>(f);
    }
}

// sum over the 64 lanes in a fixed tree (deterministic), DPP only; the total is valid in lane 63
__device__ __forceinline__ float rp_wave_total(float v) {
    auto dpp = [](float x, auto CTRL, auto RM) {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(CTRL)::value, decltype(RM)::value, 0xF, false));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xF>{});      // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xF>{});      // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xF>{});     // row_half_mirror
    v += dpp(v, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xF>{});     // row_mirror: every lane = its row's sum
    v += dpp(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xA>{});     // row_bcast15 into rows 1, 3
    v += dpp(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xC>{});     // row_bcast31 into rows 2, 3
    return v;
}
// the same tree without its last step: the sums of lanes 0..31 / 32..63, valid in lanes 31 / 63
__device__ __forceinline__ float rp_half_total(float v) {
    auto dpp = [](float x, auto CTRL, auto RM) {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(CTRL)::value, decltype(RM)::value, 0xF, false));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xF>{});
    v += dpp(v, std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xF>{});
    v += dpp(v, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xF>{});
    v += dpp(v, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xF>{});
    v += dpp(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xA>{});
    return v;
}

// four of the eight B fragments of k16 step J: patch rows kh + 4 HF .. kh + 4 HF + 3 at tap column kw, chunk kc of the slab whose
// ring buffer starts at LDS address `sbase` (runtime: a slab's buffer follows from the global slab count); inline asm, completion
// is awaited by the caller (lgkmcnt)
template <int J, int HF>
__device__ __forceinline__ void rp_bread(bf16x8 (&b)[4], const unsigned (&pb)[3], unsigned sbase) {
    constexpr int t = J % 36, kh = t / 12, kw = (t >> 2) % 3, kc = t & 3;
    // (the XOR is computed here, at every request: left to the compiler the twelve (kw, kc) combinations become twelve registers
    // held across the whole unrolled tile loop)
    unsigned pq = pb[kw];
    asm volatile("" : "+v"(pq));
    const unsigned ad = sbase + (pq ^ (unsigned)(kc << 5));
    constexpr int r0 = kh + 4 * HF;
    constexpr int o0 = (r0 + 0) * kRCols * 128, o1 = (r0 + 1) * kRCols * 128, o2 = (r0 + 2) * kRCols * 128, o3 = (r0 + 3) * kRCols * 128;
    static_assert(o3 <= 65535, "ds_read immediate");
    asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"
                 : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3])
                 : "v"(ad), "n"(o0), "n"(o1), "n"(o2), "n"(o3)
                 : "memory");
}

struct RpTile {
    int nt, mt, si, img, Y0, X0, H, W, valid, grp;
};

template <bool GNIN, bool M16>
__global__ void __launch_bounds__(512, 2) conv3x3_rp_kernel(ConvDev PA, ConvDev PB, int n_groups, char* dump) {
    constexpr int NT = 512, NW = 8;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, half = lane >> 5;
    // the lane id, recomputed where the tile loop needs it (two instructions; all 64 lanes are active everywhere in this kernel): a
    // per-lane value held across the unrolled tile loop is one register too many (it spilled, and its reload drains vmcnt)
    // (volatile asm: the builtin form is loop-invariant for the compiler, which hoists it -- and whatever is derived from it --
    // out of the tile loop again)
    auto lane_now = []() {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;

    // up to two GROUPS of tiles in one launch (two layers of identical shape and flags with their own tensors, weights and
    // GroupNorm state: cls_tower.i and center_tower.i): group 0's tiles first, then group 1's
    auto PG = [&](int g) -> const ConvDev& { return g ? PB : PA; };
    const ConvDev& P = PA;                                // launch-wide properties (flags, Cout, N, ntiles) are the same in both
    const int T0 = PA.mtiles * PA.ntiles;
    const int T = T0 + (n_groups > 1 ? PB.mtiles * PB.ntiles : 0);
    const int G = (int)gridDim.x;
    const int pos = xcd_remap(blockIdx.x, G);             // workgroups of one XCD take neighbouring tiles (shared halo rows in its L2)
    const int nmine = (T - pos + G - 1) / G;              // tiles pos, pos + G, ...   (G <= T: nmine >= 1)

    auto decode = [&](int t) {
        RpTile c;
        c.valid = t < T;
        t = t < T ? t : T - 1;
        c.grp = t >= T0 ? 1 : 0;
        t -= c.grp * T0;
        const ConvDev& Q = PG(c.grp);
        c.nt = t % Q.ntiles;
        c.mt = t / Q.ntiles;
        int si = 0;
#pragma unroll
        for (int k = 1; k < kMaxSegs; k++)
            if (k < Q.n_segs && c.mt >= Q.seg[k].tile0) si = k;
        c.si = si;
        const SegDev& S = Q.seg[si];
        const int tloc = c.mt - S.tile0;
        c.img = tloc / S.tiles_per_img;
        const int tt = tloc - c.img * S.tiles_per_img;
        const int ty = tt / S.tiles_x;
        c.Y0 = ty * kRH;
        c.X0 = (tt - ty * S.tiles_x) * kRW;
        c.H = S.Hout;
        c.W = S.Wout;
        return c;
    };

    // ---- patch DMA map: piece pc = 8 consecutive patch pixels (patch pixel pp = p * 34 + q <-> haloed input pixel
    // (Y0 + p, X0 + q)); wave w moves pieces w, w + 8, .., w + 40 of every slab -- the five waves without a sixth
    // piece re-load THEIR OWN fifth piece (same wave, in order: it lands before the wave touches the piece)
    int ppc[kRPP];
#pragma unroll
    for (int ii = 0; ii < kRPP; ii++) {
        int pc = wave + NW * ii;
        if (pc >= kRPieces) pc -= NW;
        ppc[ii] = pc;
    }
    // piece ii of channel slab sl of tile c's patch into ring buffer `buf`.  The per-lane source offset is computed at every issue
    // (about ten vector instructions; six offsets per tile held in registers across the tile loop spill): the piece's first patch
    // pixel is wave-uniform (scalar division by the row pitch), a lane adds its 0..7
    auto patch_piece = [&](const RpTile& c, int sl, int buf, int ii) {
        int lane = lane_now();
        asm volatile("" : "+v"(lane));
        const int Hp = c.H + 2, Wp = c.W + 2;
        const unsigned max_pix = (unsigned)(P.N * Hp * Wp - 1);
        const int S = ppc[ii] * 8, pS = S / kRCols, qS = S - pS * kRCols;      // scalar
        int q = qS + (lane >> 3);
        const int wrap = q >= kRCols ? 1 : 0;
        q -= wrap * kRCols;
        const int p = pS + wrap;
        unsigned g = (unsigned)((c.img * Hp + c.Y0 + p) * Wp + c.X0 + q);
        g = g < max_pix ? g : max_pix;                       // ragged tiles reach past the image (and the buffer)
        const unsigned ofs = g * (unsigned)(kRCin * 2) + (unsigned)(((lane & 7) ^ ((q >> 1) & 7)) * 16);
        __builtin_amdgcn_global_load_lds((gvoid*)(PG(c.grp).seg[c.si].in + ofs + sl * 128), (lvoid*)(lds + buf * kRSlab + ppc[ii] * 1024), 16, 0, 0);
    };
    // GroupNorm + ReLU of one landed patch piece of tile c, in place (inline-asm LDS ops: a plain LDS access would make the
    // compiler drain vmcnt).  A lane handles LOGICAL chunk lane&7 (8 channels = one group) of pixel lane>>3 of the piece.
    // scale / shift form (round 5): y = max(a x + b, 0) with a = rstd gamma, b = beta - mean a of the tile's (layer, image),
    // tabulated once per tile (ab_table).  The piece's patch coordinates come from the wave-uniform part of its pixel index
    // (scalar division by the row pitch) plus the lane's 0..7; the LDS address is lane * 16 XOR the column swizzle; packed fp32
    // math, ReLU as a packed 16-bit integer max on the rounded pairs (max(round(y), 0) = round(max(y, 0)): rounding is monotonic
    // and keeps the sign bit), out-of-image pixels masked to zero.
    auto gn_piece = [&](const RpTile& c, int sl, int buf, int ii, int statbuf) {
        if (ii == kRPP - 1 && wave + (kRPP - 1) * NW >= kRPieces) return; // duplicate of this wave's previous piece
        int lane = lane_now();
        asm volatile("" : "+v"(lane));       // recompute the per-lane parts at every call: hoisted out of the tile loop they spill
        const int pc = ppc[ii];                                           // wave-uniform
        const int S = pc * 8, pS = S / kRCols, qS = S - pS * kRCols;      // scalar
        int q = qS + (lane >> 3);
        const int wrap = q >= kRCols ? 1 : 0;
        q -= wrap * kRCols;
        const int p = pS + wrap;
        const bool inside = (unsigned)(c.Y0 + p - 1) < (unsigned)c.H && (unsigned)(c.X0 + q - 1) < (unsigned)c.W && p < kRRows;
        const unsigned msk = inside ? 0xffffffffu : 0u;
        const unsigned ad = (((unsigned)lane << 4) ^ (((unsigned)q << 3) & 0x70u)) + (lds_base + (unsigned)(buf * kRSlab + pc * 1024));
        const unsigned ta = (((unsigned)lane << 5) & 0xe0u) + (lds_base + (unsigned)(kROffAB + statbuf * 2048 + sl * kBK * 4));
        typedef __attribute__((ext_vector_type(2))) short s16x2;
        auto cvt = [&](unsigned w, f32x2 a, f32x2 b) -> unsigned {       // two channels: bf16 pair -> a x + b -> ReLU -> bf16 pair
            const f32x2 x = {__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
            const f32x2 y = __builtin_elementwise_fma(a, x, b);
            const s16x2 r = __builtin_elementwise_max(__builtin_bit_cast(s16x2, pack_bf16(y[0], y[1])), (s16x2){0, 0});
            return __builtin_bit_cast(unsigned, r) & msk;
        };
        u32x4 v, o;
        f32x4 aq, bq;
        asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %4\n\tds_read_b128 %2, %4 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(v), "=&v"(aq), "=&v"(bq)
                     : "v"(ad), "v"(ta)
                     : "memory");
        o.x = cvt(v.x, (f32x2){aq[0], aq[1]}, (f32x2){bq[0], bq[1]});
        o.y = cvt(v.y, (f32x2){aq[2], aq[3]}, (f32x2){bq[2], bq[3]});
        asm volatile("ds_read_b128 %0, %2 offset:16\n\tds_read_b128 %1, %2 offset:1040\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(aq), "=&v"(bq)
                     : "v"(ta)
                     : "memory");
        o.z = cvt(v.z, (f32x2){aq[0], aq[1]}, (f32x2){bq[0], bq[1]});
        o.w = cvt(v.w, (f32x2){aq[2], aq[3]}, (f32x2){bq[2], bq[3]});
        asm volatile("ds_write_b128 %0, %1" ::"v"(ad), "v"(o) : "memory");
    };
    // the a / b table of tile c's (layer, image) -> AB buffer b, by the first 256 threads (one channel each) from the landed
    // statistics piece and the layer's gamma / beta; inline-asm LDS operations (a plain access would make the compiler drain vmcnt)
    auto ab_table = [&](const RpTile& c, int b) {
        if (wave < kRCin / 64) {
            int td = wave * 64 + lane_now();
            asm volatile("" : "+v"(td));
            const unsigned tsd = lds_base + (unsigned)(kROffStat + b * 256) + (unsigned)(td >> 3) * 8u;
            const unsigned tgd = lds_base + (unsigned)(kROffGB + c.grp * (2 * kRCin * 4)) + (unsigned)td * 4u;
            u32x2 ms;
            float gm, bt;
            asm volatile("ds_read_b64 %0, %3\n\tds_read_b32 %1, %4\n\tds_read_b32 %2, %4 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(ms), "=&v"(gm), "=&v"(bt)
                         : "v"(tsd), "v"(tgd)
                         : "memory");
            const float av = __uint_as_float(ms.y) * gm;
            const float bv = __builtin_fmaf(-__uint_as_float(ms.x), av, bt);
            const unsigned tad = lds_base + (unsigned)(kROffAB + b * 2048) + (unsigned)td * 4u;
            asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:1024" ::"v"(tad), "v"(av), "v"(bv) : "memory");
        }
    };
    // statistics (mean, rstd of the 32 groups) of tile c's image -> stat buffer b: one 256-byte DMA piece, issued by every wave
    auto stat_piece = [&](const RpTile& c, int b) {
        const float* st = PG(c.grp).in_stats + ((size_t)c.si * P.N + c.img) * (kRCin / 8) * 2;
        __builtin_amdgcn_global_load_lds((gvoid*)(st + lane), (lvoid*)(lds + kROffStat + b * 256), 4, 0, 0);
    };

    // ---- B fragment offsets: patch row p (0..5) at tap column kw (0..2): pixel p * 34 + q, q = kw + frow; k16 step kc reads
    // the 16-byte chunk (2 kc + half) ^ sw, sw = (q >> 1) & 7, i.e. (pb[kw] ^ (kc << 5)) + p * 34 * 128 with
    // pb[kw] = q * 128 | ((half ^ (sw & 1)) << 4) | ((sw >> 1) << 5)
    // (16x16x32 form: q = kw + (lane & 15); the 32-channel group kc2 reads chunk (4 kc2 + (lane >> 4)) ^ sw, i.e. (pb[kw] ^ (kc2 << 6)) +
    // p * 34 * 128 with pb[kw] = q * 128 | (((lane >> 4) ^ (sw & 3)) << 4) | ((sw >> 2) << 6); the right column half is 2048 B further)
    unsigned pb[3];
#pragma unroll
    for (int kw = 0; kw < 3; kw++) {
        const int q = kw + (M16 ? (lane & 15) : frow);
        const int sw = (q >> 1) & 7;
        pb[kw] = M16 ? (unsigned)(q * 128 + (((lane >> 4) ^ (sw & 3)) << 4) + ((sw >> 2) << 6))
                     : (unsigned)(q * 128 + ((half ^ (sw & 1)) << 4) + ((sw >> 1) << 5));
    }

    const unsigned voff = (unsigned)(wave * kRSteps * 1024 + lane * 16);
    bf16x8 ar[kRRing];
#pragma unroll
    for (int k = 0; k < kRRing; k++) ar[k] = bf16x8{};
    auto barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    const bool relu = P.flags & DAFNE_CONV_RELU;
    const bool gn = P.flags & DAFNE_CONV_GN_STATS;
    const bool fin = gn && (P.flags & DAFNE_CONV_GN_FINALIZE);     // wave-uniform
    const int G8 = P.Cout / 8;
    const int gnlate = wave >> 2;                                  // GN_INPUT: waves 4..7 convert their pieces six steps later

    // ---- prologue: layer constants -> LDS (plain loads: nothing asynchronous is in flight yet), first tile's patch
    RpTile cur = decode(pos);
    {
        float* gb = (float*)(lds + kROffGB);
        float* lb = (float*)(lds + kROffBias);
        for (int g = 0; g < n_groups; g++) {
            if (GNIN && tid < kRCin) {
                gb[g * 2 * kRCin + tid] = PG(g).in_gamma[tid];
                gb[g * 2 * kRCin + kRCin + tid] = PG(g).in_beta[tid];
            }
            for (int k = tid; k < P.Cout; k += NT) lb[g * kRMaxCout + k] = PG(g).bias[k];
        }
        if (GNIN && tid < 64)
            ((float*)(lds + kROffStat))[tid] = (PG(cur.grp).in_stats + ((size_t)cur.si * P.N + cur.img) * (kRCin / 8) * 2)[tid];
        __syncthreads();
        if (GNIN) {
            ab_table(cur, 0);
            __syncthreads();
        }
    }
#pragma unroll
    for (int sl = 0; sl < 2; sl++)            // slabs 2, 3 follow during slabs 0, 1 of the tile, like every tile's
#pragma unroll
        for (int ii = 0; ii < kRPP; ii++) patch_piece(cur, sl, sl, ii);
    {
        const char* wf0 = PG(cur.grp).w + (size_t)cur.nt * (NW * kRSteps * 1024);
        rp_static_for<0, (M16 ? kRDist16 : kRRing)>([&](auto J) { rp_load<decltype(J)::value>(ar, wf0, wf0, voff); });
    }
    // 16 dummy dword stores (into the wave's part of the dump area): the queue behind A(0..7) now looks like a steady-state tile's
    // -- A(136..143), then the previous tile's 16 row stores -- so steps 0..7 of the first tile wait with the same counts
    {
        char* dd = dump + (size_t)tid * 256;
#pragma unroll
        for (int k = 0; k < 2 * kRFr; k++) asm volatile("global_store_dword %0, %1, off offset:%2" :: "v"(dd), "v"(0), "n"(k * 4) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"((M16 ? kRDist16 : kRRing) + 2 * kRFr) : "memory");         // this wave's 12 patch pieces have landed (the A loads + 16 stores are younger)
    if (GNIN) {
        // slab 0 only: slab 1 is normalised in the first slab period of the loop, like every tile's
#pragma unroll
        for (int ii = 0; ii < kRPP; ii++) gn_piece(cur, 0, 0, ii, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();             // slab 0 of the first tile is published (later slabs: the barrier of step 12 of every period)
    __builtin_amdgcn_sched_barrier(0);

    // 128 accumulator registers either way: 8 fragments of 32 pixels x 32 channels, or 32 of 16 pixels x 16 channels -- (tile row b,
    // column half, channel half cb) at index (b * 2 + side) * 2 + cb
    f32x16 acc[M16 ? 1 : kRFr];
    f32x4 acc16[M16 ? 4 * kRFr : 1];
    bf16x8 bfr[2][4];                         // [half of the step's eight fragments][4]
    int gs0 = 0;                              // ring buffer of the current tile's slab 0 (global slab count mod 3)
    float gsum[4], gsq[4];                    // GroupNorm sums of a tile (this wave's 4 groups): live inside its epilogue only
                                              // (16x16x32 form: a lane's 4 channels of fragment cb lie in group 2 cb + (lane >> 5): two sums)
#pragma unroll
    for (int g = 0; g < 4; g++) gsum[g] = gsq[g] = 0.f;
    RpTile prv = cur;
    prv.valid = 0;

#ifdef DAFNE_RP_TIMING
    unsigned long long rp_ts[8];
#define RP_STAMP(i) rp_ts[i] = __builtin_amdgcn_s_memtime()
#else
#define RP_STAMP(i)
#endif
    // the GroupNorm partial sums of the previous tile: behind a barrier that follows the redb writes of step 9
    auto gn_publish = [&]() {
        if (gn && prv.valid && wave == 0 && lane_now() < 32) {
            int td = lane_now();
            asm volatile("" : "+v"(td));              // (a hoisted tid * 8 spills: its reload would drain vmcnt)
            const float* redb = (const float*)(lds + kROffRed);
            const float sv = redb[td * 2 + 0], qv = redb[td * 2 + 1];
            float* o = PG(prv.grp).gn_partial + ((size_t)prv.mt * G8 + prv.nt * 32 + td) * 2;
            if (fin) {
                typedef unsigned long long u64a;
                const u64a vv = ((u64a)__float_as_uint(qv) << 32) | (u64a)__float_as_uint(sv);
                __hip_atomic_store((u64a*)o, vv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                o[0] = sv;
                o[1] = qv;
            }
        }
    };
    auto gn_ticket = [&]() {
        if (fin && wave == 0 && lane_now() == 0) {
            int last = 0;
            if (prv.valid) {
                const int old = __hip_atomic_fetch_add(PG(prv.grp).gn_counters + prv.si * P.N + prv.img, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = old == PG(prv.grp).seg[prv.si].tiles_per_img - 1;
            }
            *(int*)(lds + kROffRed + 256) = last;
        }
    };
    // the last tile of an image to arrive reduces all its partials in gn_finalize_kernel's order (cf. gn_fused_finalize);
    // every thread of the workgroup calls this behind a barrier that follows gn_ticket
    auto gn_last_tile = [&]() {
        if (fin && *(const int*)(lds + kROffRed + 256)) {
            typedef unsigned long long u64a;
            const ConvDev& Q = PG(prv.grp);
            const SegDev& S = Q.seg[prv.si];
            float* scratch = (float*)(lds + kROffFin);
            const int t0 = S.tile0 + prv.img * S.tiles_per_img;
            int tid = wave * 64 + lane_now();              // (recomputed: hoisted per-thread values spill)
            asm volatile("" : "+v"(tid));
            const int g = tid & 31;
#pragma unroll
            for (int kk = 0; kk < 2; kk++) {
                const int sl2 = (tid >> 5) + 16 * kk;
                float sum = 0.f, sqs = 0.f;
                if (g < G8)
                    for (int tt = sl2; tt < S.tiles_per_img; tt += 32) {
                        const u64a vv = __hip_atomic_load((const u64a*)(Q.gn_partial + ((size_t)(t0 + tt) * G8 + g) * 2), __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_AGENT);
                        sum += __uint_as_float((unsigned)vv);
                        sqs += __uint_as_float((unsigned)(vv >> 32));
                    }
                scratch[(sl2 * 32 + g) * 2 + 0] = sum;
                scratch[(sl2 * 32 + g) * 2 + 1] = sqs;
            }
            __syncthreads();
            if (tid < 32 && tid < G8) {
                float a = 0.f, b = 0.f;
#pragma unroll
                for (int kk = 0; kk < 32; kk++) {
                    a += scratch[(kk * 32 + tid) * 2 + 0];
                    b += scratch[(kk * 32 + tid) * 2 + 1];
                }
                const float cnt = (float)(S.Hout * S.Wout * 8);
                const float mean = a / cnt;
                float var = b / cnt - mean * mean;
                var = var > 0.f ? var : 0.f;
                float* o = Q.gn_stats_out + (((size_t)prv.si * P.N + prv.img) * G8 + tid) * 2;
                o[0] = mean;
                o[1] = rsqrtf(var + P.gn_eps);
            }
            if (tid == 0) __hip_atomic_store(Q.gn_counters + prv.si * P.N + prv.img, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
        }
    };
    auto gn_reduce_write = [&]() {             // after the 8 rp_wave_total: lane 63 holds the wave's sums
        if (gn && lane == 63) {
            float* redb = (float*)(lds + kROffRed);
#pragma unroll
            for (int g = 0; g < 4; g++) {
                redb[(wave * 4 + g) * 2 + 0] = gsum[g];
                redb[(wave * 4 + g) * 2 + 1] = gsq[g];
            }
        }
    };

    for (int k = 0; k < nmine; k++) {
        RP_STAMP(0);
#ifdef DAFNE_RP_TIMING
        const unsigned long long rp_rt0 = __builtin_amdgcn_s_memrealtime();
#endif
        const RpTile nxt = decode(pos + (k + 1 < nmine ? k + 1 : k) * G);     // the last tile re-fetches itself (nobody reads it)
        const char* wf_cur = PG(cur.grp).w + (size_t)cur.nt * (NW * kRSteps * 1024);
        const char* wf_nxt = PG(nxt.grp).w + (size_t)nxt.nt * (NW * kRSteps * 1024);
        const int sb_cur = k & 1, sb_nxt = (k + 1) & 1;
        if constexpr (M16) {
#pragma unroll
            for (int b = 0; b < 4 * kRFr; b++) acc16[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
#pragma unroll
            for (int b = 0; b < kRFr; b++)
#pragma unroll
                for (int kk = 0; kk < 16; kk++) acc[b][kk] = 0.f;
        }
        // ring buffers of this tile's four slab periods: computing (gs0 + sl) % 3; the DMA of period sl fills (gs0 + sl + 2) % 3
        // (the buffer the period's opening barrier has just retired); the normalisation of period sl works on (gs0 + sl + 1) % 3
        int bufc[4];
#pragma unroll
        for (int q = 0; q < 4; q++) bufc[q] = (gs0 + q) % kRBufs;
        unsigned sbase[4];
#pragma unroll
        for (int q = 0; q < 4; q++) sbase[q] = lds_base + (unsigned)(bufc[q] * kRSlab);

        rp_static_for<0, kRSteps>([&](auto J) {
            constexpr int j = decltype(J)::value;
            constexpr int sl = j / 36, t = j % 36;
            if constexpr (!M16) rp_wait_for<j, GNIN>(ar);
            else if constexpr ((j & 1) == 0) rp_wait16_for<j, GNIN>(ar);
            // ---- GN_INPUT, steps 0..11 of a period: the pieces this wave requested in the PREVIOUS period (slab sl + 1 of this tile; in
            // period 3: slab 0 of the next tile), one per step; waves w and w + 4 share a SIMD: the second half of the workgroup converts
            // six steps later, so that one of the two always has MFMAs for the matrix pipe
            if constexpr (GNIN) {
                if constexpr (t < 2 * kRPP) {
                    if ((t / kRPP) == gnlate) {
                        if constexpr (sl < 3) gn_piece(cur, sl + 1, (gs0 + sl + 1) % kRBufs, t % kRPP, sb_cur);
                        else gn_piece(nxt, 0, (gs0 + 4) % kRBufs, t % kRPP, sb_nxt);
                    }
                }
            }
            static_assert(2 * kRPP <= kRBar, "the normalisation of a period ends in front of its barrier");
            if constexpr (t == kRBar) {
                // THE barrier of a slab period.  It publishes the NEXT period's slab -- every wave's pieces of it have landed (a wave's
                // waits cover its own, requested 30 steps ago) and, GN_INPUT, are normalised -- and retires the PREVIOUS period's (every
                // wave is past its last read of it), whose buffer the requests of steps 13..18 refill.  No barrier at a tile's first
                // step: a wave that is through its epilogue starts the next tile while its SIMD mate is still storing rows
                // (round 6, first 8 x 32 form: a barrier at step 0 cost 6 k of 92 k cycles per tile).
                barrier();
                if constexpr (j == kRBar) RP_STAMP(1);
                if constexpr (j == 36 + kRBar) RP_STAMP(2);
                if constexpr (j == 72 + kRBar) RP_STAMP(3);
                if constexpr (j == 108 + kRBar) RP_STAMP(4);
            }
            // the next tile's a / b table: its statistics piece (step 56) is covered by this wave's wait of step 63; published by
            // the barrier of step 84, first read at step 108
            if constexpr (GNIN && j == kRTabStep) ab_table(nxt, sb_nxt);
            // ---- the previous tile's GroupNorm sums (reduced over the wave and parked in LDS by its epilogue)
            if constexpr (j == kRBar + 2) gn_publish();        // behind the barrier of step 12: the sums of all 32 groups
            if constexpr (j == kRBar + 14) gn_ticket();        // wave 0's waits since step 20 cover its partial-sum stores of step 14
            if constexpr (j == 36 + kRBar + 1) gn_last_tile(); // behind the barrier of step 48
            // ---- this tile's matrix work: eight B fragments per step in two halves of four (two register sets).  The second half is
            // requested before the first half's MFMAs, the next step's first half before the second half's; the counted lgkmcnt
            // leaves exactly the four younger reads in flight (other LDS / scalar-memory operations in flight only make the wait
            // stricter).  Requests cross slab boundaries (the next slab was published at step 12 of this period) but not the tile's end:
            // the epilogue sits there.
            if constexpr (!M16) {
                if constexpr (j == 0) rp_bread<j, 0>(bfr[0], pb, sbase[0]);
                rp_bread<j, 1>(bfr[1], pb, sbase[sl]);
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(bfr[0][0]), "+v"(bfr[0][1]), "+v"(bfr[0][2]), "+v"(bfr[0][3]) :: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; r++) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[j % kRRing], bfr[0][r], acc[r], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (j + 1 < kRSteps) {
                    rp_bread<j + 1, 0>(bfr[0], pb, sbase[(j + 1) / 36]);
                    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(bfr[1][0]), "+v"(bfr[1][1]), "+v"(bfr[1][2]), "+v"(bfr[1][3]) :: "memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bfr[1][0]), "+v"(bfr[1][1]), "+v"(bfr[1][2]), "+v"(bfr[1][3]) :: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; r++) acc[4 + r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[j % kRRing], bfr[1][r], acc[4 + r], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                rp_load<j + kRRing>(ar, wf_cur, wf_nxt, voff);
            } else {
                // 16x16x32 form: the same two halves of four B fragments -- (row 4h, left | right), (row 4h + 1, ..) then rows 4h + 2, + 3,
                // h = j & 1 -- each against both channel halves of the wave: 16 instructions of half the size
                constexpr int jb = j & ~1, h8 = (j & 1) * 16;
                if constexpr (j == 0) rp_bread16<j, 0>(bfr[0], pb, sbase[0]);
                rp_bread16<j, 1>(bfr[1], pb, sbase[sl]);
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(bfr[0][0]), "+v"(bfr[0][1]), "+v"(bfr[0][2]), "+v"(bfr[0][3]) :: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int cb = 0; cb < 2; cb++)
                        acc16[h8 + 2 * r + cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ar[(jb + cb) % kRRing], bfr[0][r], acc16[h8 + 2 * r + cb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (j + 1 < kRSteps) {
                    rp_bread16<j + 1, 0>(bfr[0], pb, sbase[(j + 1) / 36]);
                    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(bfr[1][0]), "+v"(bfr[1][1]), "+v"(bfr[1][2]), "+v"(bfr[1][3]) :: "memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bfr[1][0]), "+v"(bfr[1][1]), "+v"(bfr[1][2]), "+v"(bfr[1][3]) :: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int cb = 0; cb < 2; cb++)
                        acc16[h8 + 8 + 2 * r + cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ar[(jb + cb) % kRRing], bfr[1][r], acc16[h8 + 8 + 2 * r + cb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                rp_load<j + kRDist16>(ar, wf_cur, wf_nxt, voff);
            }
            // ---- counted vector-memory operations behind the weight load (rp_post)
            if constexpr (GNIN && j == kRStatStep) stat_piece(nxt, sb_nxt);
            // the slab two periods ahead, one piece per step: this tile's slabs 2, 3 in periods 0, 1, the next tile's slabs 0, 1 in
            // periods 2, 3
            if constexpr (t >= kRBar + 1 && t <= kRBar + kRPP) {
                if constexpr (sl < 2) patch_piece(cur, sl + 2, (gs0 + sl + 2) % kRBufs, t - kRBar - 1);
                else patch_piece(nxt, sl - 2, (gs0 + sl + 2) % kRBufs, t - kRBar - 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        gs0 = (gs0 + 4) % kRBufs;
        RP_STAMP(5);
        // ---- end of tile: this tile's epilogue.  No barrier and no patch pieces here: a wave that is done with its 144 steps goes
        // straight into its epilogue while its SIMD mate still has the matrix pipe.
        RP_STAMP(6);
        if constexpr (M16) {
            // 16x16x32 form: a lane holds, of fragment (tile row b, column half, channel half cb), pixel column 16 side + (lane & 15) and
            // the 4 channels 16 cb + 4 (lane >> 4) ..: v_permlane16_swap pairs the two channel halves across neighbouring rows of 16
            // lanes -- q = lane >> 4 even: its own channels 4q .. 4q + 3 and row q + 1's 4q + 4 .. of cb 0; q odd: the same of cb 1 --
            // 16 contiguous bytes per lane at channel 16 (q & 1) + 8 (q >> 1), one store per fragment pair, 16 per tile as before
            f32x4 bia[2];
            int le = lane_now();
            asm volatile("" : "+v"(le));
            const int col = le & 15, q4 = le >> 4;
            const unsigned bad = lds_base + (unsigned)(kROffBias + (cur.grp * kRMaxCout + cur.nt * 256 + wave * 32 + 4 * q4) * 4);
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:64\n\ts_waitcnt lgkmcnt(0)" : "=&v"(bia[0]), "=&v"(bia[1]) : "v"(bad) : "memory");
#pragma unroll
            for (int g = 0; g < 4; g++) gsum[g] = gsq[g] = 0.f;
            const int Wp = cur.W + 2;
            const size_t rowpitch = (size_t)Wp * P.Cout * 2;
            char* obase = PG(cur.grp).seg[cur.si].out + ((size_t)(cur.img * (cur.H + 2) + cur.Y0 + 1) * Wp + cur.X0 + col + 1) * P.Cout * 2
                          + (cur.nt * 256 + wave * 32 + 16 * (q4 & 1) + 8 * (q4 >> 1)) * 2;
            char* const dbase = dump + (size_t)(wave * 64 + le) * 256;
            const bool colok0 = cur.valid && (cur.X0 + col) < cur.W, colok1 = cur.valid && (cur.X0 + col + 16) < cur.W;
            const float lo = relu ? 0.f : -__builtin_inff();
            const size_t sidepitch = (size_t)16 * P.Cout * 2;
#pragma unroll
            for (int b = 0; b < kRFr; b++) {
                const bool rowok = (cur.Y0 + b) < cur.H;
#pragma unroll
                for (int side = 0; side < 2; side++) {
                    const bool valid = rowok && (side ? colok1 : colok0);
                    u32x2 pk[2];
#pragma unroll
                    for (int cb = 0; cb < 2; cb++) {
                        const f32x4 a4 = acc16[(b * 2 + side) * 2 + cb];
                        const float v0 = fmaxf(a4[0] + bia[cb][0], lo), v1 = fmaxf(a4[1] + bia[cb][1], lo);
                        const float v2 = fmaxf(a4[2] + bia[cb][2], lo), v3 = fmaxf(a4[3] + bia[cb][3], lo);
                        const float s4 = (v0 + v1) + (v2 + v3), qq = (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
                        gsum[cb] += valid ? s4 : 0.f;
                        gsq[cb] += valid ? qq : 0.f;
                        pk[cb].x = pack_bf16(v0, v1);
                        pk[cb].y = pack_bf16(v2, v3);
                    }
                    const auto r0 = __builtin_amdgcn_permlane16_swap(pk[0].x, pk[1].x, false, false);
                    const auto r1 = __builtin_amdgcn_permlane16_swap(pk[0].y, pk[1].y, false, false);
                    const u32x4 v = {r0[0], r1[0], r0[1], r1[1]};
                    char* const ad = valid ? obase + side * sidepitch : dbase;        // (a plain store: see the swap -> store hazard below)
                    *(u32x4*)ad = v;
                }
                obase += rowpitch;
            }
            if (gn) {
                // group 2 cb + (lane >> 5) of the wave: over the 32 lanes of a half wave (DPP, fixed tree), lanes 31 / 63 park the sums
#pragma unroll
                for (int cb = 0; cb < 2; cb++) {
                    gsum[cb] = rp_half_total(gsum[cb]);
                    gsq[cb] = rp_half_total(gsq[cb]);
                }
                if ((lane & 31) == 31) {
                    float* redb = (float*)(lds + kROffRed);
#pragma unroll
                    for (int cb = 0; cb < 2; cb++) {
                        redb[(wave * 4 + 2 * cb + (lane >> 5)) * 2 + 0] = gsum[cb];
                        redb[(wave * 4 + 2 * cb + (lane >> 5)) * 2 + 1] = gsq[cb];
                    }
                }
            }
        } else
        {
            f32x4 bia4[4];
            int le = lane_now();                  // per-lane parts recomputed per tile (hoisted out of the tile loop they spill)
            asm volatile("" : "+v"(le));
            const int frow = le & 31, half = le >> 5;
            const unsigned bad = lds_base + (unsigned)(kROffBias + (cur.grp * kRMaxCout + cur.nt * 256 + wave * 32 + 4 * half) * 4);
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:32\n\tds_read_b128 %2, %4 offset:64\n\tds_read_b128 %3, %4 offset:96\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(bia4[0]), "=&v"(bia4[1]), "=&v"(bia4[2]), "=&v"(bia4[3])
                         : "v"(bad)
                         : "memory");
#pragma unroll
            for (int g = 0; g < 4; g++) gsum[g] = gsq[g] = 0.f;
            // output row of this lane: pixel (Y0 + b, X0 + frow), 16 bytes at channel nt*256 + wave*32 + 8*(2 gp + half)
            const int Wp = cur.W + 2;
            const size_t rowpitch = (size_t)Wp * P.Cout * 2;
            char* obase = PG(cur.grp).seg[cur.si].out + ((size_t)(cur.img * (cur.H + 2) + cur.Y0 + 1) * Wp + cur.X0 + frow + 1) * P.Cout * 2
                          + (cur.nt * 256 + wave * 32 + 8 * half) * 2;
            // rows of out-of-image pixels: every such store of a lane goes to the same 48 bytes of the dump area (never read; one
            // address instead of sixteen -- the compiler hoists loop-invariant addresses out of the tile loop and spills them)
            char* const dbase = dump + (size_t)(wave * 64 + le) * 256;
            const bool colok = cur.valid && (cur.X0 + frow) < cur.W;
            // round 5: branch-free (the runtime flags were exec-mask branches around every group: 414 vector + 215 scalar
            // instructions per tile): ReLU as a max with 0 or -inf, the GroupNorm sums always formed and ADDED under a select
            // (x + 0 is exact: the sums of a plain / out-of-image row are unchanged bit for bit)
            const float lo = relu ? 0.f : -__builtin_inff();
#pragma unroll
            for (int b = 0; b < kRFr; b++) {
                const bool valid = colok && (cur.Y0 + b) < cur.H;
                u32x2 pk[4];
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const float v0 = fmaxf(acc[b][4 * g] + bia4[g][0], lo), v1 = fmaxf(acc[b][4 * g + 1] + bia4[g][1], lo);
                    const float v2 = fmaxf(acc[b][4 * g + 2] + bia4[g][2], lo), v3 = fmaxf(acc[b][4 * g + 3] + bia4[g][3], lo);
                    const float s4 = (v0 + v1) + (v2 + v3), q4 = (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
                    gsum[g] += valid ? s4 : 0.f;
                    gsq[g] += valid ? q4 : 0.f;
                    pk[g].x = pack_bf16(v0, v1);
                    pk[g].y = pack_bf16(v2, v3);
                }
#pragma unroll
                for (int gp = 0; gp < 2; gp++) {
                    // lanes 0..31 get (group 2gp: own channels 0..3 | the upper half-wave's 4..7), lanes 32..63 the same of group 2gp+1
                    u32x2 a = pk[2 * gp], c2 = pk[2 * gp + 1];
                    const auto r0 = __builtin_amdgcn_permlane32_swap(a.x, c2.x, false, false);
                    const auto r1 = __builtin_amdgcn_permlane32_swap(a.y, c2.y, false, false);
                    const u32x4 v = {r0[0], r1[0], r0[1], r1[1]};
                    // (a plain store: v_permlane32_swap -> vector-memory read of its result is a hazard the compiler pads with wait states
                    // only when it can see the store -- the first 8 x 32 form issued it from inline asm right behind the swaps and
                    // lost dwords of the quad-3 lanes)
                    char* const ad = valid ? obase : dbase;
#ifdef DAFNE_RP_NOSTORE                          // timing ablation (wrong results): what do the row stores cost?
                    if (gp == 1 || (b & 3) != 0) { asm volatile("" :: "v"(v)); continue; }
#endif
                    *(u32x4*)(ad + gp * 32) = v;
                }
                obase += rowpitch;
            }
            // the tile's GroupNorm sums: over the wave (DPP, fixed tree), lane 63 parks them in LDS; published behind the barrier of the
            // next tile's step 36 (round 6: rounds 3-5 carried the eight partial sums in registers into the next tile's steps 1..9)
            if (gn) {
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    gsum[g] = rp_wave_total(gsum[g]);
                    gsq[g] = rp_wave_total(gsq[g]);
                }
                gn_reduce_write();
            }
        }
        prv = cur;
        cur = nxt;
#ifdef DAFNE_RP_TIMING
        RP_STAMP(7);
        if (tid == 0 && k == 2) {                 // the third tile of every workgroup: steady state
            unsigned long long* o = (unsigned long long*)(P.gn_partial + (size_t)4096 * 64) + blockIdx.x * 16;
            for (int q = 0; q < 8; q++) o[q] = rp_ts[q];
            o[8] = rp_rt0;
            o[9] = __builtin_amdgcn_s_memrealtime();
        }
#endif
    }

    // the fragment loads past the last tile (nobody reads them) are still in flight INTO ar[]: dead registers for the compiler, which
    // hands them to the temporaries below (the 16x16x32 form did: scripts/check_async_loads.py) -- the operands keep them allocated
    // until the queue is empty
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < kRRing; k++) asm volatile("" : "+v"(ar[k]) :: "memory");
    // ---- the last tile's GroupNorm sums (in LDS since its epilogue), straight-line
    if (gn) {
        barrier();
        gn_publish();
        if (fin) {
            if (wave == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // wave 0's partial-sum stores are complete
            gn_ticket();
            barrier();
            gn_last_tile();
        }
    }
}

// ---------------------------------------------------------------------------------------
// fp8 (OCP e4m3) twin of conv3x3_patch_kernel: BASELINE config 5 (fp8 weights, CDNA4 fp8 MFMA conv path).
//
// Weights are e4m3 bytes in the same [Cout][Cin/64][KH][KW][64] order (64 B per tap row) with one fp32
// dequantisation scale per output channel; activations stay bf16 in HBM and are quantised ON LOAD: the bf16
// input patch of a 64-channel slab is DMA'd into a staging buffer and converted (after the optional
// GroupNorm + ReLU of GN_INPUT, in fp32: y * in_qscale, clamp to +-448, v_cvt_pk_fp8_f32 = round to nearest even)
// into an fp8 patch (64 B per pixel, double-buffered across slabs) by the wave that loaded the piece.  The matrix
// instruction is v_mfma_f32_32x32x64_f8f6f4 (both operands e4m3, no block scales): one instruction covers a whole
// 64-channel tap, so a K step of this kernel moves exactly the bytes a HALF step of the bf16 kernel moves
// (256 x 64 B of weights, 32 B per lane and fragment) and the schedule carries over unchanged -- two wave
// groups one phase apart, 4 weight buffers, counted vmcnt -- with half as many steps per slab (9) and twice
// the flops per phase.  fp32 accumulation; epilogue acc * oscale[cout] + bias (oscale = weight scale / in_qscale),
// then exactly the bf16 kernel's epilogue (ReLU, GroupNorm partial sums, bf16 NHWC store).
//   * fp8 patch pixel (py, px), 16-byte chunk c (16 channels): byte py*2304 + px*64 + (c ^ ((px>>2)&3))*16 (lines
//     padded to 36 pixels = 9 bank rows of 256 B).  ds_read_b128 is serviced in the lane groups {0-3,12-15,20-27} /
//     {4-11,16-19,28-31} over a 256-byte bank row (MI355X_MICROARCH.md, LDS): with that XOR the 16 pixels of a
//     group hit 16 distinct 16-byte slots for every tap offset kw, so the 2 x ds_read_b128 of a B fragment (32
//     consecutive px, 32 channels) are conflict-free (a first layout swizzled by (px>>1)&7 was 2-way conflicted:
//     SQ_LDS_BANK_CONFLICT 24 % of the LDS cycles);
//   * the next slab's 6 bf16 patch pieces per wave are issued in both phases of steps 0..2; a piece has landed
//     three steps after its issue and piece i is converted in the read phase of step 3 + i (one piece per phase:
//     two made that phase longer than the other group's MFMA phase).
typedef __attribute__((ext_vector_type(8))) int i32x8;
constexpr int kQLine = 36 * 64;                     // fp8 patch line: 34 px x 64 B, padded to 9 bank rows
constexpr int kQBuf = (kPH + 2) * kQLine;           // fp8 patch of one slab: 23 040 B
constexpr int kQOffStage = 2 * kPAStage;            // bf16 staging of one slab's patch (single buffer)
constexpr int kQOffPatch = kQOffStage + kPBuf;
constexpr int kQOffTab = kQOffPatch + 2 * kQBuf;
constexpr int kQSmem = kQOffTab + kPTabMaxC * 9;
static_assert(kQSmem <= 160 * 1024, "LDS budget");

template <bool GNIN>
__global__ void __launch_bounds__(512) conv3x3_patch_fp8_kernel(ConvDev P) {
    constexpr int NW = 8, WP = 2, TC = 2, TP = 4, BN = 256, BM = 256, NT = 512;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave / WP, wp = wave % WP;
    const int grp = wave >> 2;
    const int frow = lane & 31, half = lane >> 5;

    const int T = P.mtiles * P.ntiles;
    const int bid = xcd_remap(blockIdx.x, T);
    const int nt = bid % P.ntiles;
    const int mt = bid / P.ntiles;
    int si = 0;
#pragma unroll
    for (int k = 1; k < kMaxSegs; k++)
        if (k < P.n_segs && mt >= P.seg[k].tile0) si = k;
    const SegDev& S = P.seg[si];
    const int tloc = mt - S.tile0;
    const int img = tloc / S.tiles_per_img;
    const int tt = tloc - img * S.tiles_per_img;
    const int ty = tt / S.tiles_x, tx = tt - ty * S.tiles_x;
    const int Y0 = ty * kPH, X0 = tx * kPW;
    const int H = S.Hout, W = S.Wout, Hp = H + 2, Wp = W + 2;
    const int nslab = P.Cin / kBK;
    const int V = 9 * nslab;                // K steps: one 64-channel tap each

    float* tab_stats = (float*)(lds + kQOffTab);
    float* tab_gamma = tab_stats + P.Cin / 4;
    float* tab_beta = tab_gamma + P.Cin;
    if (GNIN) {
        const float* st = P.in_stats + ((size_t)si * P.N + img) * (P.Cin / 8) * 2;
        for (int k = tid; k < P.Cin / 4; k += NT) tab_stats[k] = st[k];
        for (int k = tid; k < P.Cin; k += NT) {
            tab_gamma[k] = P.in_gamma[k];
            tab_beta[k] = P.in_beta[k];
        }
        __syncthreads();
    }
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const float qs = P.in_qscale;

    // ---- per-lane source offsets ------------------------------------------------------------
    unsigned hofs[2];                      // weight rows of this wave's two pieces per step
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int r = (i * NW + wave) * 16 + (lane >> 2);
        const int q = (lane & 3) ^ ((r >> 2) & 3);
        hofs[i] = (unsigned)(nt * BN + r) * (unsigned)P.kbytes + (unsigned)q * 16u;
    }
    unsigned pofs[6];                      // bf16 patch pixels of this wave's six pieces per slab
    int ppi[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        int pi = i * NW + wave;
        if (pi >= kPPieces) pi -= NW;      // pieces 43..47: the wave re-fetches its previous piece
        ppi[i] = pi;
        int r = pi * 8 + (lane >> 3);
        r = r < kPRows ? r : kPRows - 1;
        const int py = r / kPCols, px = r - py * kPCols;
        int gy = Y0 + py, gx = X0 + px;
        gy = gy < Hp ? gy : Hp - 1;
        gx = gx < Wp ? gx : Wp - 1;
        const int q = (lane & 7) ^ ((px >> 1) & 7);
        pofs[i] = ((unsigned)(img * Hp + gy) * (unsigned)Wp + (unsigned)gx) * (unsigned)(P.Cin * 2) + (unsigned)q * 16u;
    }
    // B-fragment offsets inside an fp8 patch line: pixel column frow+kw, 16-byte chunks 2*half, 2*half+1
    unsigned boff[3][2];
#pragma unroll
    for (int kw = 0; kw < 3; kw++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int px = frow + kw;
            boff[kw][j] = (unsigned)px * 64u + (unsigned)(((2 * half + j) ^ ((px >> 2) & 3)) * 16);
        }
    const int fsw4 = (frow >> 2) & 3;
    unsigned hroff[2];
#pragma unroll
    for (int j = 0; j < 2; j++) hroff[j] = (unsigned)frow * 64u + (unsigned)(((2 * half + j) ^ fsw4) * 16);
    const int arow0 = wc * TC * 32;

    // Steps are walked in the order v = (slab*3 + kw)*3 + kh: the three kh taps of one kw are consecutive, so the
    // B fragments of 6 patch lines are read once and serve all three.  4 weight buffers of 256 x 64 B.
    auto piece_a = [&](int v, int i) {
        if (v >= V) return;
        const int kh = v % 3, g = v / 3;
        const int kw = g % 3, sl = g / 3;
        const unsigned koff = (unsigned)((sl * 9 + kh * 3 + kw) * 64);
        char* dst = lds + (v & 3) * kPAHalf + (i * NW + wave) * 1024;
        __builtin_amdgcn_global_load_lds((gvoid*)(P.w + hofs[i] + koff), (lvoid*)dst, 16, 0, 0);
    };
    auto piece_p = [&](int slab, int i) {
        char* dst = lds + kQOffStage + ppi[i] * 1024;
        __builtin_amdgcn_global_load_lds((gvoid*)(S.in + pofs[i] + (unsigned)slab * 128u), (lvoid*)dst, 16, 0, 0);
    };
    // bf16 staging -> (GroupNorm + ReLU) -> e4m3 patch of `slab`; a lane handles logical chunk lane&7 (8 channels)
    // of pixel row lane>>3 of the piece and writes 8 bytes.
    auto cvt_pieces = [&](int slab, int i0, int n) {
        float gmean = 0.f, grstd = 1.f;
        float gam[8], bet[8];
        if (GNIN) {
            const int ch = slab * kBK + (lane & 7) * 8;
            const unsigned ts = lds_base + (unsigned)(kQOffTab + (ch >> 3) * 8);
            const unsigned tg = lds_base + (unsigned)(kQOffTab + P.Cin + ch * 4);
            const unsigned tb = tg + (unsigned)P.Cin * 4u;
            u32x2 ms;
            f32x4 g0, g1, b0, b1;
            asm volatile("ds_read_b64 %0, %5\n\tds_read_b128 %1, %6\n\tds_read_b128 %2, %6 offset:16\n\t"
                         "ds_read_b128 %3, %7\n\tds_read_b128 %4, %7 offset:16\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(ms), "=&v"(g0), "=&v"(g1), "=&v"(b0), "=&v"(b1)
                         : "v"(ts), "v"(tg), "v"(tb)
                         : "memory");
            gmean = __uint_as_float(ms.x);
            grstd = __uint_as_float(ms.y);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                gam[k] = g0[k]; gam[4 + k] = g1[k];
                bet[k] = b0[k]; bet[4 + k] = b1[k];
            }
        }
        for (int i = i0; i < i0 + n; i++) {
            if (i == 5 && 5 * NW + wave >= kPPieces) continue;        // duplicate of piece i = 4
            const int pi = ppi[i];
            const int r = pi * 8 + (lane >> 3);
            const int py = r / kPCols, px = r - py * kPCols;
            const int gy = Y0 + py, gx = X0 + px;
            const bool inside = gy >= 1 && gy <= H && gx >= 1 && gx <= W && r < kPRows;
            const int lc = lane & 7;
            const int phys = lc ^ ((px >> 1) & 7);
            const unsigned ad = lds_base + (unsigned)(kQOffStage + pi * 1024 + (lane >> 3) * 128 + phys * 16);
            u32x4 v;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(ad) : "memory");
            const unsigned u[4] = {v.x, v.y, v.z, v.w};
            float y[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const float x = bf2f((unsigned short)(k & 1 ? u[k >> 1] >> 16 : u[k >> 1] & 0xffff));
                float t = x;
                if (GNIN) t = fmaxf((x - gmean) * grstd * gam[k] + bet[k], 0.f);      // expression of gn_apply_kernel
                t = t * qs;
                y[k] = inside ? fminf(fmaxf(t, -448.f), 448.f) : 0.f;
            }
            unsigned o0 = 0u, o1 = 0u;
            o0 = __builtin_amdgcn_cvt_pk_fp8_f32(y[0], y[1], o0, false);
            o0 = __builtin_amdgcn_cvt_pk_fp8_f32(y[2], y[3], o0, true);
            o1 = __builtin_amdgcn_cvt_pk_fp8_f32(y[4], y[5], o1, false);
            o1 = __builtin_amdgcn_cvt_pk_fp8_f32(y[6], y[7], o1, true);
            const u32x2 o = {o0, o1};
            const unsigned qd = lds_base + (unsigned)(kQOffPatch + (slab & 1) * kQBuf + py * kQLine + px * 64 +
                                                       (((lc >> 1) ^ ((px >> 2) & 3)) * 16) + (lc & 1) * 8);
            if (r < kPRows) asm volatile("ds_write_b64 %0, %1" ::"v"(qd), "v"(o) : "memory");
        }
        // the last piece of a slab is converted in the slab's last step: its write must have landed before the barrier
        // that lets the other group read the patch
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    f32x16 acc[TC][TP];
#pragma unroll
    for (int a = 0; a < TC; a++)
#pragma unroll
        for (int b = 0; b < TP; b++)
#pragma unroll
            for (int k = 0; k < 16; k++) acc[a][b][k] = 0.f;
    i32x8 af[TC], bfr[TP + 2];
    auto ld256 = [](const char* p0, const char* p1) {
        typedef __attribute__((ext_vector_type(4))) int i32x4;
        const i32x4 lo = *(const i32x4*)p0, hi = *(const i32x4*)p1;
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto read_a = [&](int v) {
        const char* sb = lds + (v & 3) * kPAHalf;
#pragma unroll
        for (int a = 0; a < TC; a++) af[a] = ld256(sb + (arow0 + a * 32) * 64 + hroff[0], sb + (arow0 + a * 32) * 64 + hroff[1]);
    };
    auto read_b6 = [&](int slab, int kw) {      // patch lines wp*4 .. wp*4+5 at column offset kw
        const char* pb = lds + kQOffPatch + (slab & 1) * kQBuf;
#pragma unroll
        for (int rr = 0; rr < TP + 2; rr++)
            bfr[rr] = ld256(pb + (wp * TP + rr) * kQLine + boff[kw][0], pb + (wp * TP + rr) * kQLine + boff[kw][1]);
    };
    auto mma8 = [&](int kh) {
#pragma unroll
        for (int a = 0; a < TC; a++)
#pragma unroll
            for (int b = 0; b < TP; b++)
                acc[a][b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[a], bfr[b + kh], acc[a][b], 0, 0, 0, 0, 0, 0);
    };
    // odd phases wait: the 4 youngest weight pieces plus the patch pieces issued in this and the previous step
    // (np = 0, 2 or 4) may stay in flight across the barrier
    auto phase_end = [&](bool odd, int t, int np) {
        if (odd) {
            if (t <= 2 * V - 7) {
                if (np == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (np == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: patch of slab 0 (DMA + conversion), weights of steps 0..2 -----------------------------
#pragma unroll
    for (int i = 0; i < 6; i++) piece_p(0, i);
    piece_a(0, 0); piece_a(0, 1); piece_a(1, 0); piece_a(1, 1); piece_a(2, 0); piece_a(2, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");          // this wave's patch pieces have landed
    cvt_pieces(0, 0, 6);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");          // step 0 is complete
    phase_end(false, -1, 0);

    // Global phase t = 2 v + p (p = 0: fragment reads of group 0 / MFMAs of group 1, p = 1: the reverse) issues
    // weight piece p of step v + 3 and, in steps 0..2 of a slab, patch piece 2*UU + p of the next slab.
#define NPATCH(UU) (more ? (((UU) <= 2 ? 2 : 0) + ((UU) >= 1 && (UU) <= 3 ? 2 : 0)) : 0)
#define QPATCH_G0(UU, KW_, KH_)                                                                              \
    {                                                                                                        \
        const int v = slab * 9 + (UU);                                                                       \
        piece_a(v + 3, 0);                                                                                   \
        if (more && (UU) <= 2) piece_p(slab + 1, 2 * (UU));                                                  \
        if (more && (UU) >= 3) cvt_pieces(slab + 1, (UU) - 3, 1);                                       \
        read_a(v);                                                                                           \
        if ((KH_) == 0) read_b6(slab, KW_);                                                                  \
        phase_end(false, 2 * v, 0);                                                                          \
        piece_a(v + 3, 1);                                                                                   \
        if (more && (UU) <= 2) piece_p(slab + 1, 2 * (UU) + 1);                                              \
        mma8(KH_);                                                                                           \
        phase_end(true, 2 * v + 1, NPATCH(UU));                                                              \
    }
    // group 1 runs one phase behind: reads in p = 1 of step v, MFMAs in p = 0 of step v + 1
#define QPATCH_G1(UU, KW_, KH_)                                                                              \
    {                                                                                                        \
        const int v = slab * 9 + (UU);                                                                       \
        piece_a(v + 3, 1);                                                                                   \
        if (more && (UU) <= 2) piece_p(slab + 1, 2 * (UU) + 1);                                              \
        if (more && (UU) >= 3) cvt_pieces(slab + 1, (UU) - 3, 1);                                       \
        read_a(v);                                                                                           \
        if ((KH_) == 0) read_b6(slab, KW_);                                                                  \
        phase_end(true, 2 * v + 1, NPATCH(UU));                                                              \
        piece_a(v + 4, 0);                                                                                   \
        if ((UU) < 8) {                                                                                      \
            if (more && (UU) + 1 <= 2) piece_p(slab + 1, 2 * ((UU) + 1));                                    \
        } else if (slab + 2 < nslab) {                                                                       \
            piece_p(slab + 2, 0);                                                                            \
        }                                                                                                    \
        mma8(KH_);                                                                                           \
        if (v + 1 < V) phase_end(false, 2 * v + 2, 0);                                                       \
    }
#define QPATCH_SLAB(M)                                                                                       \
    M(0, 0, 0) M(1, 0, 1) M(2, 0, 2) M(3, 1, 0) M(4, 1, 1) M(5, 1, 2) M(6, 2, 0) M(7, 2, 1) M(8, 2, 2)
    if (grp == 0) {
        for (int slab = 0; slab < nslab; slab++) {
            const bool more = slab + 1 < nslab;
            QPATCH_SLAB(QPATCH_G0)
        }
    } else {
        // phase 0 of the whole loop: this group idles one phase (issues its share of the loads only)
        piece_a(3, 0);
        if (nslab > 1) piece_p(1, 0);
        phase_end(false, 0, 0);
        for (int slab = 0; slab < nslab; slab++) {
            const bool more = slab + 1 < nslab;
            QPATCH_SLAB(QPATCH_G1)
        }
    }
#undef QPATCH_SLAB
#undef QPATCH_G0
#undef QPATCH_G1
#undef NPATCH

    // ------------------------------------------------------------ epilogue (scale, bias, ReLU, GN sums, bf16)
    const bool relu = P.flags & DAFNE_CONV_RELU;
    const float relu_lo = relu ? 0.f : -__builtin_inff();
    const bool gn = P.flags & DAFNE_CONV_GN_STATS;
    const bool fin = gn && (P.flags & DAFNE_CONV_GN_FINALIZE);     // wave-uniform
    float fin_sq[2] = {0.f, 0.f};
    bool fin_writer = false;
    constexpr int ROWB = BN * 2 + 16;
    constexpr int CHB = BN / 8;
    char* stg = lds;
    float gsum[TC][4], gsq[TC][4];
#pragma unroll
    for (int a = 0; a < TC; a++)
#pragma unroll
        for (int g = 0; g < 4; g++) gsum[a][g] = gsq[a][g] = 0.f;
    __syncthreads();   // every wave is done with the weight stages and the patch
#pragma unroll
    for (int a = 0; a < TC; a++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int co = nt * BN + (wc * TC + a) * 32 + 8 * g + 4 * half;
            const float4 bia = P.bias ? *(const float4*)(P.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 osc = *(const float4*)(P.oscale + co);
#pragma unroll
            for (int b = 0; b < TP; b++) {
                const int px = (wp * TP + b) * 32 + frow;
                const bool valid = (Y0 + wp * TP + b) < H && (X0 + frow) < W;
                // (branch-free: ReLU as a max with 0 / -inf, the GroupNorm sums added under a select -- x + 0 is exact; round 5)
                const float v0 = fmaxf(acc[a][b][4 * g] * osc.x + bia.x, relu_lo), v1 = fmaxf(acc[a][b][4 * g + 1] * osc.y + bia.y, relu_lo);
                const float v2 = fmaxf(acc[a][b][4 * g + 2] * osc.z + bia.z, relu_lo), v3 = fmaxf(acc[a][b][4 * g + 3] * osc.w + bia.w, relu_lo);
                gsum[a][g] += valid ? (v0 + v1) + (v2 + v3) : 0.f;
                gsq[a][g] += valid ? (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3) : 0.f;
                uint2 pk;
                pk.x = pack_bf16(v0, v1);
                pk.y = pack_bf16(v2, v3);
                *(uint2*)(stg + px * ROWB + ((wc * TC + a) * 32 + 8 * g + 4 * half) * 2) = pk;
            }
        }
    float* redb = (float*)(lds + BM * ROWB);   // [NW][TC*4][2]
    if (gn) {
#pragma unroll
        for (int a = 0; a < TC; a++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                float sv = gsum[a][g], qv = gsq[a][g];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    sv += __shfl_xor(sv, o, 64);
                    qv += __shfl_xor(qv, o, 64);
                }
                if (lane == 0) {
                    redb[(wave * TC * 4 + a * 4 + g) * 2 + 0] = sv;
                    redb[(wave * TC * 4 + a * 4 + g) * 2 + 1] = qv;
                }
            }
    }
    __syncthreads();
    if (gn && tid < BN / 8) {
        const int wcc = tid / (TC * 4), ag = tid % (TC * 4);
        float sv = 0.f, qv = 0.f;
#pragma unroll
        for (int p2 = 0; p2 < WP; p2++) {
            sv += redb[((wcc * WP + p2) * TC * 4 + ag) * 2 + 0];
            qv += redb[((wcc * WP + p2) * TC * 4 + ag) * 2 + 1];
        }
        const int group = (nt * BN) / 8 + tid;
        if (group < P.Cout / 8) {
            if (fin) {
                fin_sq[0] = sv;
                fin_sq[1] = qv;
                fin_writer = true;
            } else {
                float* o = P.gn_partial + ((size_t)mt * (P.Cout / 8) + group) * 2;
                o[0] = sv;
                o[1] = qv;
            }
        }
    }
    constexpr int CPT = BM * CHB / NT;       // 16-byte chunks per thread
#pragma unroll
    for (int i = 0; i < CPT; i++) {
        const int idx = tid + i * NT;
        const int p = idx / CHB, cc = idx - p * CHB;
        const int gy = Y0 + (p >> 5), gx = X0 + (p & 31);
        if (gy < H && gx < W) {
            const size_t opix = (size_t)(img * Hp + gy + 1) * Wp + gx + 1;
            const uint4 v = *(const uint4*)(stg + p * ROWB + cc * 16);
            *(uint4*)(S.out + (opix * P.Cout + nt * BN + cc * 8) * 2) = v;
        }
    }
    if (fin) {
        __syncthreads();                     // the staging tile has been read: its LDS is free for the reduction
        gn_fused_finalize(P, S, si, img, mt, fin_sq, fin_writer, tid, (float*)lds, (int*)(lds + 32 * 32 * 2 * 4));
    }
}

// ---------------------------------------------------------------------------------------
// 3x3 stride-1 convolution with <= 32 output channels and fp32 NHWC output: the prediction convolutions of
// the head (cls_logits 15, center_pred 2, corners_pred+ctrness 9; dafne.py:318-344).  1/8 of a tower layer's
// MFMA work on the same input bytes: these layers are bound by operand staging and barriers, not by the matrix
// pipe, so the structure is the opposite of the kernels above -- per 64-channel slab BOTH operands are complete
// in LDS (input patch 10x34 px, 43.5 KB; weights 9 taps x 32 cout x 64 ch, 36 KB; both double-buffered) and
// the nine taps run with NO barrier in between: two barriers per slab instead of 36.  8 waves = 4 pixel-row
// pairs x 2 K halves (the halves are summed through LDS at the end); the B fragments of 4 patch lines serve
// the three kh taps (7 fragment reads per 6 MFMAs).  GN_INPUT as in conv3x3_patch_kernel: the last tower
// layer's GroupNorm + ReLU is applied to the patch in LDS, so no normalisation pass is left in the head.
constexpr int kSlabPatch = kPRows * 128;            // 43 520 B (the last DMA piece is half masked)
constexpr int kSlabW = 9 * 32 * 128;                // 36 864 B
constexpr int kSlabOffW = 2 * kSlabPatch;
constexpr int kSlabOffTab = kSlabOffW + 2 * kSlabW;
constexpr int kSlabTabC = 256;                      // GN_INPUT: Cin <= 256 (stats + gamma + beta = 9 * Cin bytes)
constexpr int kSlabSmem = kSlabOffTab + 9 * kSlabTabC;
static_assert(kSlabSmem <= 160 * 1024, "LDS budget");

template <bool GNIN>
__global__ void __launch_bounds__(512) conv3x3_slab_kernel(ConvDev P) {
    constexpr int NW = 8, NT = 512;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rp = wave & 3, kq = wave >> 2;         // pixel-row pair (rows 2rp, 2rp+1), K half (chunks 2kq, 2kq+1)
    const int frow = lane & 31, half = lane >> 5;

    const int mt = xcd_remap(blockIdx.x, P.mtiles);
    int si = 0;
#pragma unroll
    for (int k = 1; k < kMaxSegs; k++)
        if (k < P.n_segs && mt >= P.seg[k].tile0) si = k;
    const SegDev& S = P.seg[si];
    const int tloc = mt - S.tile0;
    const int img = tloc / S.tiles_per_img;
    const int tt = tloc - img * S.tiles_per_img;
    const int ty = tt / S.tiles_x, tx = tt - ty * S.tiles_x;
    const int Y0 = ty * kPH, X0 = tx * kPW;
    const int H = S.Hout, W = S.Wout, Hp = H + 2, Wp = W + 2;
    const int nslab = P.Cin / kBK;

    float* tab_stats = (float*)(lds + kSlabOffTab);
    float* tab_gamma = tab_stats + P.Cin / 4;
    float* tab_beta = tab_gamma + P.Cin;
    if (GNIN) {
        const float* st = P.in_stats + ((size_t)si * P.N + img) * (P.Cin / 8) * 2;
        for (int k = tid; k < P.Cin / 4; k += NT) tab_stats[k] = st[k];
        for (int k = tid; k < P.Cin; k += NT) {
            tab_gamma[k] = P.in_gamma[k];
            tab_beta[k] = P.in_beta[k];
        }
        __syncthreads();
    }
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;

    // ---- DMA maps: patch pieces (8 px x 128 B, 6 per wave) and weight pieces (8 rows x 128 B, 36 per slab)
    unsigned pofs[6];
    int ppi[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        int pi = i * NW + wave;
        if (pi >= kPPieces) pi -= NW;
        ppi[i] = pi;
        int r = pi * 8 + (lane >> 3);
        r = r < kPRows ? r : kPRows - 1;
        const int py = r / kPCols, px = r - py * kPCols;
        int gy = Y0 + py, gx = X0 + px;
        gy = gy < Hp ? gy : Hp - 1;
        gx = gx < Wp ? gx : Wp - 1;
        const int q = (lane & 7) ^ ((px >> 1) & 7);
        pofs[i] = ((unsigned)(img * Hp + gy) * (unsigned)Wp + (unsigned)gx) * (unsigned)(P.Cin * 2) + (unsigned)q * 16u;
    }
    auto load_slab = [&](int slab) {
        char* pb = lds + (slab & 1) * kSlabPatch;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            // piece 42 holds patch pixels 336..339 only: its upper half would run into the next buffer
            if (ppi[i] < kPPieces - 1 || lane < 32)
                __builtin_amdgcn_global_load_lds((gvoid*)(S.in + pofs[i] + (unsigned)slab * 128u), (lvoid*)(pb + ppi[i] * 1024), 16, 0, 0);
        }
        char* wb = lds + kSlabOffW + (slab & 1) * kSlabW;
        for (int p = wave; p < 36; p += NW) {             // piece p: tap p/4, weight rows (p%4)*8 .. +8
            const int tap = p >> 2;
            const int r = (p & 3) * 8 + (lane >> 3);
            const int q = (lane & 7) ^ ((r >> 1) & 7);
            const unsigned off = (unsigned)r * (unsigned)P.kbytes + (unsigned)((slab * 9 + tap) * kRowBytes) + (unsigned)q * 16u;
            __builtin_amdgcn_global_load_lds((gvoid*)(P.w + off), (lvoid*)(wb + p * 1024), 16, 0, 0);
        }
    };
    auto gn_slab = [&](int slab) {                         // this wave's patch pieces, in place (see conv3x3_patch_kernel)
        const int ch = slab * kBK + (lane & 7) * 8;
        const unsigned ts = lds_base + (unsigned)(kSlabOffTab + (ch >> 3) * 8);
        const unsigned tg = lds_base + (unsigned)(kSlabOffTab + P.Cin + ch * 4);
        const unsigned tb = tg + (unsigned)P.Cin * 4u;
        u32x2 ms;
        f32x4 g0, g1, b0, b1;
        asm volatile("ds_read_b64 %0, %5\n\tds_read_b128 %1, %6\n\tds_read_b128 %2, %6 offset:16\n\t"
                     "ds_read_b128 %3, %7\n\tds_read_b128 %4, %7 offset:16\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(ms), "=&v"(g0), "=&v"(g1), "=&v"(b0), "=&v"(b1)
                     : "v"(ts), "v"(tg), "v"(tb)
                     : "memory");
        const float gmean = __uint_as_float(ms.x), grstd = __uint_as_float(ms.y);
        const float gam[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
        const float bet[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        // all six pieces are read first (one LDS round trip instead of six), then converted and written back one by one
        u32x4 vv[6];
        unsigned adr[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const int pi = ppi[i];
            int r = pi * 8 + (lane >> 3);
            r = r < kPRows ? r : kPRows - 1;
            const int px = r % kPCols;
            const int phys = (lane & 7) ^ ((px >> 1) & 7);
            adr[i] = lds_base + (unsigned)((slab & 1) * kSlabPatch + pi * 1024 + (lane >> 3) * 128 + phys * 16);
            asm volatile("ds_read_b128 %0, %1" : "=&v"(vv[i]) : "v"(adr[i]) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vv[0]), "+v"(vv[1]), "+v"(vv[2]), "+v"(vv[3]), "+v"(vv[4]), "+v"(vv[5]) : : "memory");
#pragma unroll
        for (int i = 0; i < 6; i++) {
            if (i == 5 && 5 * NW + wave >= kPPieces) continue;
            const int pi = ppi[i];
            const int r = pi * 8 + (lane >> 3);
            if (r >= kPRows) continue;                     // masked half of the last piece
            const int py = r / kPCols, px = r - py * kPCols;
            const int gy = Y0 + py, gx = X0 + px;
            const bool inside = gy >= 1 && gy <= H && gx >= 1 && gx <= W;
            const unsigned ad = adr[i];
            const u32x4 v = vv[i];
            const unsigned u[4] = {v.x, v.y, v.z, v.w};
            float y[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const float x = bf2f((unsigned short)(k & 1 ? u[k >> 1] >> 16 : u[k >> 1] & 0xffff));
                y[k] = fmaxf((x - gmean) * grstd * gam[k] + bet[k], 0.f);      // expression of gn_apply_kernel
            }
            u32x4 o;
            o.x = inside ? pack_bf16(y[0], y[1]) : 0u;
            o.y = inside ? pack_bf16(y[2], y[3]) : 0u;
            o.z = inside ? pack_bf16(y[4], y[5]) : 0u;
            o.w = inside ? pack_bf16(y[6], y[7]) : 0u;
            asm volatile("ds_write_b128 %0, %1" ::"v"(ad), "v"(o) : "memory");
        }
    };
    auto wait_all = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // fragment offsets: weights (128-B rows, swizzle (row>>1)&7), patch lines (column frow + kw)
    const int fsw = (frow >> 1) & 7;
    unsigned aoff[2], boff[3][2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int ks = 2 * kq + c;
        aoff[c] = (unsigned)frow * 128u + (unsigned)(((2 * ks + half) ^ fsw) * 16);
#pragma unroll
        for (int kw = 0; kw < 3; kw++)
            boff[kw][c] = (unsigned)(frow + kw) * 128u + (unsigned)(((2 * ks + half) ^ (((frow + kw) >> 1) & 7)) * 16);
    }

    f32x16 acc[2];
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
        for (int k = 0; k < 16; k++) acc[b][k] = 0.f;

    load_slab(0);
    wait_all();
    if (GNIN) gn_slab(0);
    for (int slab = 0; slab < nslab; slab++) {
        barrier();                                    // slab's patch (normalised) and weights are complete in LDS
        if (slab + 1 < nslab) load_slab(slab + 1);    // lands under the 36 MFMAs below
        const char* pb = lds + (slab & 1) * kSlabPatch;
        const char* wb = lds + kSlabOffW + (slab & 1) * kSlabW;
#pragma unroll
        for (int kw = 0; kw < 3; kw++)
#pragma unroll
            for (int c = 0; c < 2; c++) {
                bf16x8 bfr[4], af[3];
#pragma unroll
                for (int rr = 0; rr < 4; rr++) bfr[rr] = *(const bf16x8*)(pb + ((2 * rp + rr) * kPCols) * 128 + boff[kw][c]);
#pragma unroll
                for (int kh = 0; kh < 3; kh++) af[kh] = *(const bf16x8*)(wb + (kh * 3 + kw) * 4096 + aoff[c]);
#pragma unroll
                for (int kh = 0; kh < 3; kh++)
#pragma unroll
                    for (int b = 0; b < 2; b++)
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kh], bfr[b + kh], acc[b], 0, 0, 0);
            }
        if (slab + 1 < nslab) {
            wait_all();                               // this wave's pieces of the next slab have landed
            if (GNIN) gn_slab(slab + 1);
        }
    }

    // ---- epilogue: sum the two K halves through LDS, bias, fp32 NHWC store (Cout <= 32 channels per pixel)
    barrier();                                        // all waves are done with the operand buffers
    float* stg = (float*)lds;                         // [256 px][33] fp32 (row pad against bank conflicts)
    constexpr int ROW = 33;
    if (kq == 1) {
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int px = (2 * rp + b) * 32 + frow;
#pragma unroll
            for (int g = 0; g < 4; g++)
#pragma unroll
                for (int k = 0; k < 4; k++) stg[px * ROW + 8 * g + 4 * half + k] = acc[b][4 * g + k];
        }
    }
    __syncthreads();
    if (kq == 0) {
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int px = (2 * rp + b) * 32 + frow;
#pragma unroll
            for (int g = 0; g < 4; g++)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int co = 8 * g + 4 * half + k;
                    stg[px * ROW + co] += acc[b][4 * g + k] + (P.bias && co < P.Cout ? P.bias[co] : 0.f);
                }
        }
    }
    __syncthreads();
    float* out = (float*)S.out;
    for (int idx = tid; idx < 256 * P.Cout; idx += NT) {
        const int p = idx / P.Cout, c = idx - p * P.Cout;
        const int gy = Y0 + (p >> 5), gx = X0 + (p & 31);
        if (gy < H && gx < W) out[((size_t)(img * H + gy) * W + gx) * P.Cout + c] = stg[p * ROW + c];
    }
}

// ---------------------------------------------------------------------------------------
// The slab kernel's persistent form for the head's own shapes: 256 input channels, <= 16 output channels (cls_logits 15 / 16,
// corners_pred + ctrness 9, center_pred 2; dafne.py:318-344).  At 1/16 of a tower layer's matrix work these launches are
// bound by the 512 B per input pixel they read, so everything else is taken off the critical path:
//   * v_mfma_f32_16x16x32_bf16 (16 output channels per instruction: no padding to 32) with ALL weights resident in LDS,
//     fragment-major (72 k32 steps x 1 KB), loaded once per workgroup instead of once per tile;
//   * the workgroup walks its tiles (8 x 32 output pixels) as one stream of 64-channel slabs: raw pixels go global ->
//     registers two slabs ahead (87 KB in flight per CU), are normalised (GN_INPUT: GroupNorm + ReLU, the expression of
//     gn_apply_kernel) and written to one of two LDS slab buffers one slab ahead; one barrier per slab, the next tile's
//     first slabs are on their way while the current tile's last ones are multiplied;
//   * wave = (row pair, column half): 2 x 16 output pixels over the whole K, so the fp32 results leave from registers.
// K order per output: (slab, kw, k32 half, kh) -- fp32 sums, compared with torch at 2e-3 like the slab kernel.
constexpr int kVSlab = kPRows * 128;                // 43 520 B: 340 patch pixels x 64 channels
constexpr int kVOffW = 2 * kVSlab;
constexpr int kVSteps = 72;                         // k32 steps: (slab, tap, half)
constexpr int kVSmem = kVOffW + kVSteps * 1024;     // 160 768 B
constexpr int kVCin = 256;
static_assert(kVSmem <= 160 * 1024, "LDS budget");

struct QTile {
    int si, img, Y0, X0, H, W;
};

template <bool GNIN>
__global__ void __launch_bounds__(512) conv3x3_pred16_kernel(ConvDev P) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rp = wave & 3, ch = wave >> 2;        // output rows 2rp, 2rp + 1; columns 16ch .. +16
    const int fcol = lane & 15, kg = lane >> 4;     // fragment column / row, 8-element K group
    const int q = tid & 7;                          // this thread's 16-byte chunk (8 channels) of every pixel it moves
    const int T = P.mtiles, G = (int)gridDim.x;
    const int ntl = (T - (int)blockIdx.x + G - 1) / G;      // tiles of this workgroup (>= 1: G <= T)

    // ---- the six 16-byte pieces a thread moves per slab: patch pixel r = i * 64 + tid / 8, chunk q
    int ppos[6];                                    // py | px << 8
    unsigned wofs[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        int r = i * 64 + (tid >> 3);
        r = r < kPRows ? r : kPRows - 1;
        const int py = r / kPCols, px = r - py * kPCols;
        ppos[i] = py | (px << 8);
        wofs[i] = (unsigned)(r * 128 + ((q ^ ((px >> 1) & 7)) << 4));
    }
    const bool last_piece = tid < (kPRows - 5 * 64) * 8;    // piece 5 exists for patch pixels 320..339

    float gam[4][8], bet[4][8];
    if (GNIN) {
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const f32x4 g0 = *(const f32x4*)(P.in_gamma + s * 64 + q * 8), g1 = *(const f32x4*)(P.in_gamma + s * 64 + q * 8 + 4);
            const f32x4 b0 = *(const f32x4*)(P.in_beta + s * 64 + q * 8), b1 = *(const f32x4*)(P.in_beta + s * 64 + q * 8 + 4);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                gam[s][k] = g0[k]; gam[s][4 + k] = g1[k];
                bet[s][k] = b0[k]; bet[s][4 + k] = b1[k];
            }
        }
    }
    float bias[4];
#pragma unroll
    for (int k = 0; k < 4; k++) bias[k] = (P.bias && 4 * kg + k < P.Cout) ? P.bias[4 * kg + k] : 0.f;

    auto decode = [&](int v) {
        const int mt = xcd_remap(v < T ? v : T - 1, T);
        QTile c;
        int si = 0;
#pragma unroll
        for (int k = 1; k < kMaxSegs; k++)
            if (k < P.n_segs && mt >= P.seg[k].tile0) si = k;
        c.si = si;
        const SegDev& S = P.seg[si];
        const int tloc = mt - S.tile0;
        c.img = tloc / S.tiles_per_img;
        const int tt = tloc - c.img * S.tiles_per_img;
        const int ty = tt / S.tiles_x;
        c.Y0 = ty * kPH;
        c.X0 = (tt - ty * S.tiles_x) * kPW;
        c.H = S.Hout;
        c.W = S.Wout;
        return c;
    };

    // per-lane source offsets of a tile's pieces (slab 0) and which of them are pixels of the image (GN_INPUT: the halo and
    // the clamped overhang must stay zero after the normalisation)
    unsigned pofs[6];
    const char* pin = nullptr;
    auto patch_map = [&](const QTile& c, unsigned& inside) {
        const int Hp = c.H + 2, Wp = c.W + 2;
        inside = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            int gy = c.Y0 + (ppos[i] & 0xff), gx = c.X0 + (ppos[i] >> 8);
            gy = gy < Hp ? gy : Hp - 1;
            gx = gx < Wp ? gx : Wp - 1;
            if (gy >= 1 && gy <= c.H && gx >= 1 && gx <= c.W) inside |= 1u << i;
            pofs[i] = ((unsigned)(c.img * Hp + gy) * (unsigned)Wp + (unsigned)gx) * (unsigned)(kVCin * 2) + (unsigned)q * 16u;
        }
        pin = P.seg[c.si].in;
    };
    u32x4 raw[2][6];
    auto load = [&](int set, int slab) {
#pragma unroll
        for (int i = 0; i < 6; i++) raw[set][i] = *(const u32x4*)(pin + pofs[i] + (unsigned)slab * 128u);
    };
    auto load_stats = [&](const QTile& c, f32x2 (&ms)[4]) {
        const float* st = P.in_stats + ((size_t)c.si * P.N + c.img) * (kVCin / 8) * 2;
#pragma unroll
        for (int s = 0; s < 4; s++) ms[s] = *(const f32x2*)(st + (s * 8 + q) * 2);
    };
    // piece i of a slab: registers -> (GroupNorm + ReLU) -> LDS slab buffer
    auto convert_piece = [&](int i, int set, int slab, int buf, const f32x2& ms, unsigned inside) {
        u32x4 o = raw[set][i];
        if (GNIN) {
            // the expression of gn_apply_kernel, two channels per instruction (v_pk_add / v_pk_mul / v_pk_fma_f32 round like
            // their scalar forms); ReLU on the rounded pair as 16-bit integers (a negative bf16 is a negative int16; the
            // rounding keeps the sign, so max before or after it is the same number)
            f32x2 m2 = {ms[0], ms[0]}, r2 = {ms[1], ms[1]};
            // both halves MATERIALISED: left to itself the compiler multiplies by the (mean, rstd) pair with `op_sel:[1,0]` (the low
            // lane reading the pair's HIGH half) -- the one packed-fp32 operand form that returns wrong lanes beside matrix kernels of
            // other waves on gfx950 (scratch/pk_probe.py: 1194 of 1500 launches; every other form 0; profiles/NOTES_r05.md)
            asm volatile("" : "+v"(m2), "+v"(r2));
            const unsigned msk = (inside >> i) & 1u ? 0xffffffffu : 0u;
            unsigned w[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const f32x2 x = {__uint_as_float(w[j] << 16), __uint_as_float(w[j] & 0xffff0000u)};
                const f32x2 t = (x - m2) * r2;
                const f32x2 g2 = {gam[slab][2 * j], gam[slab][2 * j + 1]}, b2 = {bet[slab][2 * j], bet[slab][2 * j + 1]};
                const f32x2 y = t * g2 + b2;
                const i16x2 pr = __builtin_bit_cast(i16x2, pack_bf16(y[0], y[1]));
                const i16x2 zero = {0, 0};
                w[j] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(pr, zero)) & msk;
            }
            o.x = w[0]; o.y = w[1]; o.z = w[2]; o.w = w[3];
        }
        if (i < 5 || last_piece) *(u32x4*)(lds + buf * kVSlab + wofs[i]) = o;
    };
    auto convert = [&](int set, int slab, int buf, const f32x2& ms, unsigned inside) {
#pragma unroll
        for (int i = 0; i < 6; i++) convert_piece(i, set, slab, buf, ms, inside);
    };
    auto barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // fragment offsets: weights lane-linear; patch pixel (line, 16ch + kw + fcol), chunk 4 * half + kg, swizzled by the column
    unsigned boff[3][2];
#pragma unroll
    for (int kw = 0; kw < 3; kw++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int pc = ch * 16 + kw + fcol;
            boff[kw][h] = (unsigned)((2 * rp * kPCols + pc) * 128 + (((4 * h + kg) ^ ((pc >> 1) & 7)) << 4));
        }
    f32x4 acc[2];
    // one slab: six groups (kw, k32 half) of 4 patch-line + 3 weight fragments feeding 6 MFMAs; the fragments of group g + 1
    // are requested before the MFMAs of group g (pinned with sched_group_barrier: left alone the compiler keeps one or two
    // reads in flight and the wave sits in LDS latency)
    auto compute = [&](int slab, int buf) {
        const char* pb = lds + buf * kVSlab;
        const char* wb = lds + kVOffW + slab * (18 * 1024) + lane * 16;
        bf16x8 bfr[2][4], af[2][3];
        auto fetch = [&](int g, int set) {
            const int kw = g >> 1, h = g & 1;
#pragma unroll
            for (int l = 0; l < 4; l++) bfr[set][l] = *(const bf16x8*)(pb + l * (kPCols * 128) + boff[kw][h]);
#pragma unroll
            for (int kh = 0; kh < 3; kh++) af[set][kh] = *(const bf16x8*)(wb + ((kh * 3 + kw) * 2 + h) * 1024);
        };
        fetch(0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 7, 0);
#pragma unroll
        for (int g = 0; g < 6; g++) {
            if (g + 1 < 6) {
                fetch(g + 1, (g + 1) & 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 7, 0);
            }
#pragma unroll
            for (int kh = 0; kh < 3; kh++)
#pragma unroll
                for (int b = 0; b < 2; b++)
                    acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[g & 1][kh], bfr[g & 1][b + kh], acc[b], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
        }
    };
    const int nco = P.Cout - 4 * kg < 0 ? 0 : (P.Cout - 4 * kg > 4 ? 4 : P.Cout - 4 * kg);     // this lane's channels 4kg .. 4kg + nco
    auto store_tile = [&](const QTile& c) {
        float* out = (float*)P.seg[c.si].out;
        const int gx = c.X0 + ch * 16 + fcol;
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int gy = c.Y0 + 2 * rp + b;
            if (gy < c.H && gx < c.W && nco > 0) {
                float* o = out + ((size_t)(c.img * c.H + gy) * c.W + gx) * P.Cout + 4 * kg;
                const float v0 = acc[b][0] + bias[0], v1 = acc[b][1] + bias[1], v2 = acc[b][2] + bias[2], v3 = acc[b][3] + bias[3];
                // one store per pixel row: the address is 4-byte aligned only (Cout = 15, 9), which global stores allow
                if (nco == 4) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(o), "v"((f32x4){v0, v1, v2, v3}) : "memory");
                else if (nco == 3) asm volatile("global_store_dwordx3 %0, %1, off" ::"v"(o), "v"((f32x3){v0, v1, v2}) : "memory");
                else if (nco == 2) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(o), "v"((f32x2){v0, v1}) : "memory");
                else asm volatile("global_store_dword %0, %1, off" ::"v"(o), "v"(v0) : "memory");
            }
        }
    };

    // ---- prologue: slabs 0 and 1 of the first tile on their way, slab 0 into buffer 0, slab 2 behind it
    QTile cur = decode((int)blockIdx.x);
    unsigned in_cur = 0, in_nxt = 0;
    f32x2 ms_cur[4], ms_nxt[4];
    patch_map(cur, in_cur);
    load(0, 0);
    load(1, 1);
    if (GNIN) load_stats(cur, ms_cur);
    // ---- weights: k32 step st = (slab * 9 + tap) * 2 + half at 1 KB each, lane-linear (rows >= Cout repeat the last one:
    // their results are never stored); requested behind the first tile's pixels
    {
        const int row = fcol < P.Cout ? fcol : P.Cout - 1;
        const char* wsrc = P.w + (size_t)row * P.kbytes + kg * 16;
        for (int st = wave; st < kVSteps; st += 8)
            *(u32x4*)(lds + kVOffW + st * 1024 + lane * 16) = *(const u32x4*)(wsrc + (st >> 1) * 128 + (st & 1) * 64);
    }
    convert(0, 0, 0, ms_cur[0], in_cur);
    load(0, 2);

    for (int it = 0; it < ntl; it++) {
        const bool more = it + 1 < ntl;
        const QTile nxt = decode((int)blockIdx.x + (it + 1) * G);
#pragma unroll
        for (int k = 0; k < 4; k++) acc[0][k] = acc[1][k] = 0.f;
        // slab 0: slab 1 -> buffer 1, slab 3 leaves
        barrier();
        convert(1, 1, 1, ms_cur[1], in_cur);
        load(1, 3);
        // from here on the addresses are the next tile's.  After the last tile the loads still leave (every lane re-reads
        // the first 16 bytes of the map: one cache line): a load under a branch would make the compiler count none of them
        // when it waits for the older ones, and every wait would drain the whole queue
        patch_map(nxt, in_nxt);
        if (!more) {
#pragma unroll
            for (int i = 0; i < 6; i++) pofs[i] = 0;
        }
        if (GNIN) load_stats(nxt, ms_nxt);
        compute(0, 0);
        // slab 1: slab 2 -> buffer 0, the next tile's slab 0 leaves
        barrier();
        convert(0, 2, 0, ms_cur[2], in_cur);
        load(0, 0);
        compute(1, 1);
        // slab 2: slab 3 -> buffer 1, the next tile's slab 1 leaves
        barrier();
        convert(1, 3, 1, ms_cur[3], in_cur);
        load(1, 1);
        compute(2, 0);
        // slab 3: the next tile's slab 0 -> buffer 0 (after the last tile: sixteen bytes of nothing, never read), its slab 2
        // leaves
        barrier();
        convert(0, 0, 0, ms_nxt[0], in_nxt);
        load(0, 2);
        compute(3, 1);
        store_tile(cur);
        cur = nxt;
        in_cur = in_nxt;
#pragma unroll
        for (int s = 0; s < 4; s++) ms_cur[s] = ms_nxt[s];
    }
}

struct Cfg {
    int bn, bm;
};

// Tile choice: 256x256 (8 waves, 1 block/CU) halves the L2->LDS operand traffic per
// FLOP, but only pays when the launch still has >= 2 blocks per CU; otherwise the
// 128x128 (4 waves, 2 blocks/CU) tile keeps the 256 CUs busy.
// Every occupancy heuristic of the kernel choice counts the tiles of kNominalBatch images, not of the batch at hand: which
// kernel runs a layer (and with it the tiling of its GroupNorm partial sums) is a function of the IMAGE shape only, so an image
// gets the same bits whichever batch it sits in (the TTA wrapper's grouped views; tests/test_gpu_model.py).
constexpr int kNominalBatch = 8;

Cfg pick_cfg(int Cout, long long blocks256 = 0, unsigned flags = 0, bool ws_shape = false) {
    static const bool big = getenv("DAFNE_CONV_NO256") == nullptr;
    static const long long min_blocks = getenv("DAFNE_CONV_BIG_MIN_BLOCKS") ? atoll(getenv("DAFNE_CONV_BIG_MIN_BLOCKS")) : 512;
    const bool res = flags & (DAFNE_CONV_RESIDUAL | DAFNE_CONV_UPSAMPLE_ADD);   // HBM-bound epilogue: prefer 2 blocks/CU
    // 1x1 layers with <= 256 input channels stream through the weight-stationary kernel (128-wide tiles) faster than through
    // the 256^2 tile: res3.0's projection shortcut 58 against 77 us at batch 8 (the 512-channel one of res4.0: 63 against 50)
    if (big && !res && !ws_shape && Cout % 256 == 0 && blocks256 >= min_blocks) return {256, 256};
    if (Cout >= 128) return {128, 128};
    if (Cout > 32) return {64, 256};
    return {32, 256};
}

// 3x3 / stride 1 / pad 1 layers with Cout % 256 == 0 and a plain bf16 output go to the patch kernel when
// the launch has enough 8x32 tiles to fill the chip (the head towers, the FPN output convolutions).
// shape conditions of the patch kernels (bf16 and fp8), without the occupancy heuristic
bool patch_shape_ok(const dafne_conv_params* p, const dafne_conv_seg* segs) {
    if (p->KH != 3 || p->KW != 3 || p->stride != 1 || p->pad != 1) return false;
    if (p->Cin % kBK || p->Cout % 256 || !p->d_bias) return false;
    if (p->flags & (DAFNE_CONV_RESIDUAL | DAFNE_CONV_UPSAMPLE_ADD | DAFNE_CONV_OUT_F32)) return false;
    if ((p->flags & DAFNE_CONV_GN_INPUT) && p->Cin > kPTabMaxC) return false;
    for (int s = 0; s < p->n_segs; s++)
        if (segs[s].Hin != segs[s].Hout || segs[s].Win != segs[s].Wout) return false;
    return true;
}

bool patch_eligible(const dafne_conv_params* p, const dafne_conv_seg* segs) {
    static const int mode = getenv("DAFNE_CONV_PATCH") ? atoi(getenv("DAFNE_CONV_PATCH")) : 1;
    if (!mode) return false;
    if (!patch_shape_ok(p, segs)) return false;
    long long tiles = 0;
    for (int s = 0; s < p->n_segs; s++)
        tiles += (long long)((segs[s].Hout + kPH - 1) / kPH) * ((segs[s].Wout + kPW - 1) / kPW) * kNominalBatch;
    static const long long min_tiles = getenv("DAFNE_CONV_PATCH_MIN_TILES") ? atoll(getenv("DAFNE_CONV_PATCH_MIN_TILES")) : 200;
    return (p->flags & DAFNE_CONV_GN_INPUT) || tiles * (p->Cout / 256) >= min_tiles;
}

// 3x3 / stride 1 / pad 1 layers with Cout <= 32 and fp32 output (the prediction convolutions) go to the slab kernel.
bool slab_eligible(const dafne_conv_params* p, const dafne_conv_seg* segs) {
    static const int mode = getenv("DAFNE_CONV_SLAB") ? atoi(getenv("DAFNE_CONV_SLAB")) : 1;
    if (!mode) return false;
    if (p->KH != 3 || p->KW != 3 || p->stride != 1 || p->pad != 1) return false;
    if (p->Cin % kBK || p->Cout > 32 || !(p->flags & DAFNE_CONV_OUT_F32)) return false;
    if (p->flags & (DAFNE_CONV_RESIDUAL | DAFNE_CONV_UPSAMPLE_ADD | DAFNE_CONV_GN_STATS | DAFNE_CONV_RELU)) return false;
    if ((p->flags & DAFNE_CONV_GN_INPUT) && p->Cin > kSlabTabC) return false;
    for (int s = 0; s < p->n_segs; s++)
        if (segs[s].Hin != segs[s].Hout || segs[s].Win != segs[s].Wout) return false;
    return true;
}

// shape conditions of the resident-patch kernel (conv3x3_rp_kernel)
bool rp_shape_ok(const dafne_conv_params* p, const dafne_conv_seg* segs) {
    if (!patch_shape_ok(p, segs) || p->Cin != kRCin || p->Cout > kRMaxCout) return false;
    for (int s = 0; s < p->n_segs; s++)
        if ((long long)p->n_images * (segs[s].Hout + 2) * (segs[s].Wout + 2) * (kRCin * 2) > 0xffffffffll) return false;
    return true;
}

int build(ConvDev& D, const dafne_conv_params* p, const dafne_conv_seg* segs, bool fp8 = false, bool rp = false) {
    if (!p || !segs) return dafne::fail(DAFNE_E_INVALID, "conv: null params");
    D.oscale = nullptr;
    D.in_qscale = 1.f;
    if (p->n_segs < 1 || p->n_segs > kMaxSegs || p->n_images < 1) return dafne::fail(DAFNE_E_INVALID, "conv: bad segment/image count");
    {
        const unsigned known = DAFNE_CONV_RELU | DAFNE_CONV_RESIDUAL | DAFNE_CONV_UPSAMPLE_ADD | DAFNE_CONV_OUT_F32 | DAFNE_CONV_GN_STATS |
                               DAFNE_CONV_GN_INPUT | DAFNE_CONV_GN_FINALIZE | DAFNE_CONV_EXCLUSIVE | (rp ? DAFNE_CONV_FRAG16 : 0u);
        if (p->flags & ~known) return dafne::fail(DAFNE_E_INVALID, "conv: unknown flag bits 0x%x", p->flags & ~known);
    }
    const bool stem = p->Cin == 4 && p->KH == 7 && p->KW == 7 && p->stride == 2;
    if (!stem) {
        if (p->Cin % kBK) return dafne::fail(DAFNE_E_UNSUPPORTED, "conv: Cin %d not a multiple of 64", p->Cin);
        if (!((p->KH == 1 && p->KW == 1 && p->pad == 0) || (p->KH == 3 && p->KW == 3 && p->pad == 1)))
            return dafne::fail(DAFNE_E_UNSUPPORTED, "conv: only 1x1/p0 and 3x3/p1 (and the 7x7 stem)");
        if (p->stride != 1 && p->stride != 2) return dafne::fail(DAFNE_E_UNSUPPORTED, "conv: stride %d", p->stride);
    }
    if (p->Cout < 1) return dafne::fail(DAFNE_E_INVALID, "conv: Cout");
    long long b256 = 0;
    for (int s = 0; s < p->n_segs; s++)
        b256 += (long long)((segs[s].Hout * segs[s].Wout + 255) / 256) * kNominalBatch * (p->Cout / 256);
    const bool ws_shape = p->KH == 1 && p->KW == 1 && p->Cin <= 256;      // (never a function of the flags: plans probe the
                                                                           // tile geometry of a layer without its GN_STATS flag)
    Cfg c = pick_cfg(p->Cout, b256, p->flags, ws_shape);
    D.bn = c.bn;
    D.bm = c.bm;
    D.Cout_pad = (p->Cout + c.bn - 1) / c.bn * c.bn;
    if (!(p->flags & DAFNE_CONV_OUT_F32) && D.Cout_pad != p->Cout)
        return dafne::fail(DAFNE_E_UNSUPPORTED, "conv: bf16 output needs Cout %% %d == 0", c.bn);
    if ((p->flags & DAFNE_CONV_GN_STATS) && (!p->d_gn_partial || (p->Cout % 8))) return dafne::fail(DAFNE_E_INVALID, "conv: GN stats buffer");
    if ((p->flags & (DAFNE_CONV_RESIDUAL | DAFNE_CONV_UPSAMPLE_ADD)) == (DAFNE_CONV_RESIDUAL | DAFNE_CONV_UPSAMPLE_ADD))
        return dafne::fail(DAFNE_E_INVALID, "conv: residual and upsample-add are exclusive");
    if (!p->d_weight) return dafne::fail(DAFNE_E_INVALID, "conv: null weight");
    D.n_segs = p->n_segs; D.N = p->n_images;
    D.Cin = p->Cin; D.Cout = p->Cout; D.KH = p->KH; D.KW = p->KW; D.stride = p->stride; D.pad = p->pad;
    D.flags = p->flags; D.w = (const char*)p->d_weight; D.bias = p->d_bias; D.gn_partial = p->d_gn_partial;
    D.stem = stem ? 1 : 0;
    D.ksteps = stem ? 4 : p->KH * p->KW * p->Cin / kBK;
    D.kbytes = D.ksteps * kRowBytes;
    D.in_stats = p->d_in_gn_stats; D.in_gamma = p->d_in_gn_gamma; D.in_beta = p->d_in_gn_beta;
    D.patch = patch_eligible(p, segs) ? 1 : 0;
    if (fp8) {
        // e4m3 weights: the fp8 patch kernel is the only fp8 kernel (64-byte tap rows)
        if (stem || !patch_shape_ok(p, segs))
            return dafne::fail(DAFNE_E_UNSUPPORTED, "conv fp8w: needs 3x3 s1 p1, Cin %% 64 == 0, Cout %% 256 == 0, bias, bf16 output, "
                                                    "no residual / top-down add (GN_INPUT: Cin <= 512)");
        D.patch = 1;
        D.kbytes = D.ksteps * kBK;
    }
    D.rp = 0;
    if (rp) {
        if (stem || !rp_shape_ok(p, segs))
            return dafne::fail(DAFNE_E_UNSUPPORTED, "conv3x3_c256: needs 3x3 s1 p1, Cin == 256, Cout %% 256 == 0, bias, bf16 output, "
                                                    "no residual / top-down add");
        D.patch = 0;
        D.rp = 1;
    }
    D.slab = !D.patch && !D.rp && slab_eligible(p, segs) ? 1 : 0;
    if ((p->flags & DAFNE_CONV_GN_INPUT) && !D.patch && !D.slab && !D.rp)
        return dafne::fail(DAFNE_E_UNSUPPORTED, "conv: GN_INPUT needs the 3x3 patch kernel (3x3 s1 p1, Cout %% 256 == 0, Cin <= 512, bias, "
                                                "no residual / fp32 output) or the slab kernel (3x3 s1 p1, Cout <= 32, fp32 output, Cin <= 256)");
    if ((p->flags & DAFNE_CONV_GN_INPUT) && (!p->d_in_gn_stats || !p->d_in_gn_gamma || !p->d_in_gn_beta))
        return dafne::fail(DAFNE_E_INVALID, "conv: GN_INPUT without statistics / affine pointers");
    D.gn_stats_out = nullptr; D.gn_counters = nullptr; D.gn_eps = 0.f;
    if (p->flags & DAFNE_CONV_GN_FINALIZE) {
        if (!(p->flags & DAFNE_CONV_GN_STATS) || !(D.patch || D.rp) || p->Cout != 256)
            return dafne::fail(DAFNE_E_UNSUPPORTED, "conv: GN_FINALIZE needs GN_STATS on a 3x3 patch-kernel layer with Cout == 256");
        if (!p->d_gn_stats_out || !p->d_gn_counters || !(p->gn_eps > 0.f))
            return dafne::fail(DAFNE_E_INVALID, "conv: GN_FINALIZE without statistics / counter buffers or eps");
        D.gn_stats_out = p->d_gn_stats_out; D.gn_counters = p->d_gn_counters; D.gn_eps = p->gn_eps;
    }
    if (D.patch) {
        D.bn = 256; D.bm = 256;
        D.Cout_pad = p->Cout;
        c.bn = 256; c.bm = 256;
    }
    if (D.slab) {
        D.bn = 32; D.bm = 256;
        D.Cout_pad = 32;
        c.bn = 32; c.bm = 256;
    }
    if (D.rp) {
        D.bn = 256; D.bm = kRPx;
        D.Cout_pad = p->Cout;
        c.bn = 256; c.bm = kRPx;
    }
    int t = 0;
    for (int s = 0; s < p->n_segs; s++) {
        const dafne_conv_seg& g = segs[s];
        if (!g.d_in || !g.d_out || g.Hout < 1 || g.Wout < 1) return dafne::fail(DAFNE_E_INVALID, "conv: segment %d", s);
        if ((p->flags & (DAFNE_CONV_RESIDUAL | DAFNE_CONV_UPSAMPLE_ADD)) && !g.d_res) return dafne::fail(DAFNE_E_INVALID, "conv: missing residual");
        if ((p->flags & DAFNE_CONV_UPSAMPLE_ADD) && ((g.Hout | g.Wout) & 1)) return dafne::fail(DAFNE_E_UNSUPPORTED, "conv: upsample-add needs even output size");
        // geometry check: every tap of every output pixel must fall inside the haloed input
        if (!stem) {
            if ((g.Hout - 1) * p->stride + p->KH - p->pad > g.Hin + 1 || (g.Wout - 1) * p->stride + p->KW - p->pad > g.Win + 1)
                return dafne::fail(DAFNE_E_INVALID, "conv: segment %d output %dx%d reaches outside the input halo", s, g.Hout, g.Wout);
        } else if ((g.Hout - 1) * 2 + 8 > g.Hin || (g.Wout - 1) * 2 + 8 > g.Win) {
            return dafne::fail(DAFNE_E_INVALID, "conv: stem input must be padded to 2*Hout+6");
        }
        if ((long long)g.Hout * g.Wout > (1 << 20) || g.Wout > (1 << 12))
            return dafne::fail(DAFNE_E_UNSUPPORTED, "conv: segment %d output %dx%d above 2^20 pixels per image", s, g.Hout, g.Wout);
        SegDev& o = D.seg[s];
        o.in = (const char*)g.d_in; o.out = (char*)g.d_out; o.res = (const char*)g.d_res;
        o.Hin = g.Hin; o.Win = g.Win; o.Hout = g.Hout; o.Wout = g.Wout;
        o.tiles_x = (g.Wout + kPW - 1) / kPW;
        o.tiles_per_img = (D.patch || D.slab) ? o.tiles_x * ((g.Hout + kPH - 1) / kPH)
                          : D.rp ? o.tiles_x * ((g.Hout + kRH - 1) / kRH) : (g.Hout * g.Wout + c.bm - 1) / c.bm;
        o.tile0 = t;
        t += o.tiles_per_img * p->n_images;
    }
    D.mtiles = t;
    D.ntiles = D.Cout_pad / c.bn;
    return DAFNE_OK;
}

template <int WC, int WP, int TC, int TP, int NS = 2>
int launch(const ConvDev& D, hipStream_t st) {
    constexpr int ROWS = (WC * TC + WP * TP) * 32;
    constexpr int BN = WC * TC * 32, BM = WP * TP * 32, NW = WC * WP;
    constexpr int EN = BN > 128 ? 128 : BN;
    constexpr int stage2 = NS * ROWS * kRowBytes;
    constexpr int epi_f32 = BM * (EN * 4 + 16) + NW * (EN / 8) * 2 * 4;
    constexpr int epi_b16 = BM * (BN * 2 + 16) + NW * (WC * TC * 32 / 8) * 2 * 4;
    constexpr int epi = epi_f32 > epi_b16 ? epi_f32 : epi_b16;
    constexpr int smem = stage2 > epi ? stage2 : epi;
    DAFNE_MAX_LDS_ONCE(smem, (const void*)conv_igemm_kernel<WC, WP, TC, TP, NS>);
    hipLaunchKernelGGL((conv_igemm_kernel<WC, WP, TC, TP, NS>), dim3(D.mtiles * D.ntiles), dim3(WC * WP * 64), smem, st, D);
    return dafne::check_launch("conv_igemm");
}

// The persistent streaming kernel takes the bf16-output 128x128-tile layers without GroupNorm
// statistics / top-down add (those keep the one-tile-per-workgroup kernel).
bool stream_eligible(const ConvDev& D) {
    static const int mode = getenv("DAFNE_CONV_STREAM") ? atoi(getenv("DAFNE_CONV_STREAM")) : 2;
    if (!mode || D.stem || D.bn != 128 || D.bm != 128) return false;
    if (D.flags & (DAFNE_CONV_UPSAMPLE_ADD | DAFNE_CONV_OUT_F32 | DAFNE_CONV_GN_STATS)) return false;
    if (D.Cout % 128 || !D.bias || D.ksteps < 1) return false;
    for (int s = 0; s < D.n_segs; s++)
        if (D.seg[s].Wout > 1024 || (long long)D.seg[s].Hout * D.seg[s].Wout > (1 << 20)) return false;
    // 3x3 layers are LDS-read bound at this tile shape and lose to the full-K-step loop (measured
    // 700 vs 860 TFLOP/s on res4): only mode 1 (experiments) sends them here
    if (mode == 2 && (D.KH != 1 || D.Cin > 512)) return false;   // K >= 1024: the one-tile kernel is ahead (29 vs 32 us)
    return true;
}

int launch_patch(const ConvDev& D, hipStream_t st) {
    DAFNE_MAX_LDS_ONCE(kPSmem, (const void*)conv3x3_patch_kernel<false>, (const void*)conv3x3_patch_kernel<true>);
    const dim3 grid(D.mtiles * D.ntiles), block(512);
    if (D.flags & DAFNE_CONV_GN_INPUT) hipLaunchKernelGGL(conv3x3_patch_kernel<true>, grid, block, kPSmem, st, D);
    else hipLaunchKernelGGL(conv3x3_patch_kernel<false>, grid, block, kPSmem, st, D);
    return dafne::check_launch("conv3x3_patch");
}

int launch_patch_fp8(const ConvDev& D, hipStream_t st) {
    DAFNE_MAX_LDS_ONCE(kQSmem, (const void*)conv3x3_patch_fp8_kernel<false>, (const void*)conv3x3_patch_fp8_kernel<true>);
    const dim3 grid(D.mtiles * D.ntiles), block(512);
    if (D.flags & DAFNE_CONV_GN_INPUT) hipLaunchKernelGGL(conv3x3_patch_fp8_kernel<true>, grid, block, kQSmem, st, D);
    else hipLaunchKernelGGL(conv3x3_patch_fp8_kernel<false>, grid, block, kQSmem, st, D);
    return dafne::check_launch("conv3x3_patch_fp8");
}

int launch_rp(const ConvDev& D, const ConvDev* D2, char* dump, hipStream_t st) {
    DAFNE_MAX_LDS_ONCE(kRSmem, (const void*)conv3x3_rp_kernel<false, false>, (const void*)conv3x3_rp_kernel<true, false>,
                       (const void*)conv3x3_rp_kernel<false, true>, (const void*)conv3x3_rp_kernel<true, true>);
    int cus = 0;                                           // persistent: at most one workgroup per CU, tiles dealt round-robin
    if (int rc = dafne::device_cus(&cus)) return rc;
    const int T = D.mtiles * D.ntiles + (D2 ? D2->mtiles * D2->ntiles : 0);
    const int ng = D2 ? 2 : 1;
    const ConvDev& E = D2 ? *D2 : D;
    // balanced: with R = ceil(T / CUs) rounds, ceil(T / R) workgroups do R (or R - 1) tiles each -- no half-empty last round
    // (348 tiles: 174 workgroups x 2, not 256 of which 92 do a second tile), and the CUs left over stay free for the
    // kernels of the other sub-batch streams
    static const int cap = getenv("DAFNE_RP_GRID") ? atoi(getenv("DAFNE_RP_GRID")) : 0;
    const int lim = cap > 0 && cap < cus ? cap : cus;
    const int rounds = (T + lim - 1) / lim;
    const int G = (T + rounds - 1) / rounds;
    const dim3 grid(G), block(512);
    const bool gnin = D.flags & DAFNE_CONV_GN_INPUT, m16 = D.flags & DAFNE_CONV_FRAG16;
    if (gnin && m16) hipLaunchKernelGGL((conv3x3_rp_kernel<true, true>), grid, block, kRSmem, st, D, E, ng, dump);
    else if (gnin) hipLaunchKernelGGL((conv3x3_rp_kernel<true, false>), grid, block, kRSmem, st, D, E, ng, dump);
    else if (m16) hipLaunchKernelGGL((conv3x3_rp_kernel<false, true>), grid, block, kRSmem, st, D, E, ng, dump);
    else hipLaunchKernelGGL((conv3x3_rp_kernel<false, false>), grid, block, kRSmem, st, D, E, ng, dump);
    return dafne::check_launch("conv3x3_rp");
}


// the slab kernel's persistent 16-channel form takes the head's prediction layers (256 -> <= 16 channels)
bool pred16_ok(const ConvDev& D) {
    const char* e = getenv("DAFNE_CONV_PRED16");            // read per call: the tests run both kernels in one process
    if ((e && atoi(e) == 0) || !D.slab || D.Cin != kVCin || D.Cout > 16) return false;
    for (int s = 0; s < D.n_segs; s++)
        if ((long long)D.N * (D.seg[s].Hout + 2) * (D.seg[s].Wout + 2) * (kVCin * 2) > 0xffffffffll) return false;
    return true;
}

int launch_pred16(const ConvDev& D, hipStream_t st) {
    DAFNE_MAX_LDS_ONCE(kVSmem, (const void*)conv3x3_pred16_kernel<false>, (const void*)conv3x3_pred16_kernel<true>);
    int cus = 0;
    if (int rc = dafne::device_cus(&cus)) return rc;
    const int T = D.mtiles;
    const int rounds = (T + cus - 1) / cus;                // balanced persistent grid
    const dim3 grid((T + rounds - 1) / rounds), block(512);
    if (D.flags & DAFNE_CONV_GN_INPUT) hipLaunchKernelGGL(conv3x3_pred16_kernel<true>, grid, block, kVSmem, st, D);
    else hipLaunchKernelGGL(conv3x3_pred16_kernel<false>, grid, block, kVSmem, st, D);
    return dafne::check_launch("conv3x3_pred16");
}

int launch_slab(const ConvDev& D, hipStream_t st) {
    if (pred16_ok(D)) return launch_pred16(D, st);
    DAFNE_MAX_LDS_ONCE(kSlabSmem, (const void*)conv3x3_slab_kernel<false>, (const void*)conv3x3_slab_kernel<true>);
    const dim3 grid(D.mtiles), block(512);
    if (D.flags & DAFNE_CONV_GN_INPUT) hipLaunchKernelGGL(conv3x3_slab_kernel<true>, grid, block, kSlabSmem, st, D);
    else hipLaunchKernelGGL(conv3x3_slab_kernel<false>, grid, block, kSlabSmem, st, D);
    return dafne::check_launch("conv3x3_slab");
}

bool ws_eligible(const ConvDev& D) {
    static const bool on = getenv("DAFNE_CONV_WS") == nullptr || atoi(getenv("DAFNE_CONV_WS")) != 0;
    return on && D.KH == 1 && D.KW == 1 && (D.Cin == 64 || D.Cin == 128 || D.Cin == 256);
}

int resident_slots(int* out) {
    int cus = 0;            // resident workgroups: 2 per CU of the current device
    if (int rc = dafne::device_cus(&cus)) return rc;
    *out = 2 * cus;
    return DAFNE_OK;
}

template <int WC, int WP, int KS, bool RES, int RB>
int launch_ws_cfg(const ConvDev& D, hipStream_t st) {
    constexpr int smem = kWRing + kSRes;
    DAFNE_MAX_LDS_ONCE(smem, (const void*)conv_ws_kernel<WC, WP, KS, RES, RB>);
    int slots = 0;
    if (int rc = resident_slots(&slots)) return rc;
    const int T = D.mtiles * D.ntiles;
    const int G = T < slots ? T : slots;
    hipLaunchKernelGGL((conv_ws_kernel<WC, WP, KS, RES, RB>), dim3(G), dim3(256), smem, st, D);
    return dafne::check_launch("conv_ws");
}

int launch_ws(const ConvDev& D, hipStream_t st) {
    const bool res = D.flags & DAFNE_CONV_RESIDUAL;
    static const int rb = getenv("DAFNE_WS_RB") ? atoi(getenv("DAFNE_WS_RB")) : 0;     // experiments: force 64 / 128
    const bool full256 = rb ? rb == 128 : true, full_small = rb == 128;
    if (D.Cin == 256) {
        if (full256) return res ? launch_ws_cfg<4, 1, 4, true, 128>(D, st) : launch_ws_cfg<4, 1, 4, false, 128>(D, st);
        return res ? launch_ws_cfg<4, 1, 4, true, 64>(D, st) : launch_ws_cfg<4, 1, 4, false, 64>(D, st);
    }
    if (D.Cin == 128) {
        if (full_small) return res ? launch_ws_cfg<2, 2, 2, true, 128>(D, st) : launch_ws_cfg<2, 2, 2, false, 128>(D, st);
        return res ? launch_ws_cfg<2, 2, 2, true, 64>(D, st) : launch_ws_cfg<2, 2, 2, false, 64>(D, st);
    }
    if (full_small) return res ? launch_ws_cfg<2, 2, 1, true, 128>(D, st) : launch_ws_cfg<2, 2, 1, false, 128>(D, st);
    return res ? launch_ws_cfg<2, 2, 1, true, 64>(D, st) : launch_ws_cfg<2, 2, 1, false, 64>(D, st);
}

int launch_stream(const ConvDev& D, hipStream_t st) {
    constexpr int smem = kSNS * kSHS + kSRes;
    DAFNE_MAX_LDS_ONCE(smem, (const void*)conv_stream_kernel);
    int slots = 0;          // resident workgroups: 2 per CU
    if (int rc = resident_slots(&slots)) return rc;
    const int T = D.mtiles * D.ntiles;
    const int G = T < slots ? T : slots;
    hipLaunchKernelGGL(conv_stream_kernel, dim3(G), dim3(256), smem, st, D);
    return dafne::check_launch("conv_stream");
}

}  // namespace

extern "C" {

int dafne_conv2d_cout_pad(int Cout) {
    if (Cout < 1) return 0;
    Cfg c = pick_cfg(Cout);
    return (Cout + c.bn - 1) / c.bn * c.bn;
}

int dafne_conv2d_tile_pixels(const dafne_conv_params* prm, const dafne_conv_seg* segs) {
    ConvDev D;
    if (build(D, prm, segs)) return -1;
    return D.bm;
}

int dafne_conv2d_num_tiles(const dafne_conv_params* prm, const dafne_conv_seg* segs) {
    ConvDev D;
    if (build(D, prm, segs)) return -1;
    return D.mtiles;
}

int dafne_conv2d_tiles_per_image(const dafne_conv_params* prm, const dafne_conv_seg* segs, int32_t* out) {
    ConvDev D;
    int rc = build(D, prm, segs);
    if (rc) return rc;
    if (!out) return dafne::fail(DAFNE_E_INVALID, "conv: null output");
    for (int s = 0; s < D.n_segs; s++) out[s] = D.seg[s].tiles_per_img;
    return DAFNE_OK;
}

int dafne_conv2d_fp8w_num_tiles(const dafne_conv_params* prm, const dafne_conv_seg* segs) {
    ConvDev D;
    if (build(D, prm, segs, true)) return -1;
    return D.mtiles;
}

int dafne_conv2d_fp8w_tiles_per_image(const dafne_conv_params* prm, const dafne_conv_seg* segs, int32_t* out) {
    ConvDev D;
    int rc = build(D, prm, segs, true);
    if (rc) return rc;
    if (!out) return dafne::fail(DAFNE_E_INVALID, "conv fp8w: null output");
    for (int s = 0; s < D.n_segs; s++) out[s] = D.seg[s].tiles_per_img;
    return DAFNE_OK;
}

/* conv3x3_rp_kernel: tile geometry, eligibility, launch */
int dafne_conv3x3_c256_ok(const dafne_conv_params* prm, const dafne_conv_seg* segs) {
    ConvDev D;
    return build(D, prm, segs, false, true) == DAFNE_OK ? 1 : 0;
}

int dafne_conv3x3_c256_num_tiles(const dafne_conv_params* prm, const dafne_conv_seg* segs) {
    ConvDev D;
    if (build(D, prm, segs, false, true)) return -1;
    return D.mtiles;
}

int dafne_conv3x3_c256_tiles_per_image(const dafne_conv_params* prm, const dafne_conv_seg* segs, int32_t* out) {
    ConvDev D;
    int rc = build(D, prm, segs, false, true);
    if (rc) return rc;
    if (!out) return dafne::fail(DAFNE_E_INVALID, "conv3x3_c256: null output");
    for (int s = 0; s < D.n_segs; s++) out[s] = D.seg[s].tiles_per_img;
    return DAFNE_OK;
}

size_t dafne_conv3x3_c256_scratch_bytes(void) { return (size_t)kRDumpBytes; }

int dafne_conv3x3_c256_hip(const dafne_conv_params* prm, const dafne_conv_seg* segs, const void* d_wfrag, void* d_scratch,
                           size_t scratch_bytes, void* stream) {
    ConvDev D;
    int rc = build(D, prm, segs, false, true);
    if (rc) return rc;
    if (!d_wfrag) return dafne::fail(DAFNE_E_INVALID, "conv3x3_c256: null fragment-major weights");
    if (!d_scratch || scratch_bytes < (size_t)kRDumpBytes) return dafne::fail(DAFNE_E_WORKSPACE, "conv3x3_c256: scratch %zu < %d", scratch_bytes, kRDumpBytes);
    D.w = (const char*)d_wfrag;
    return launch_rp(D, nullptr, (char*)d_scratch, (hipStream_t)stream);
}


int dafne_conv3x3_c256_pair_hip(const dafne_conv_params* prm_a, const dafne_conv_seg* segs_a, const void* d_wfrag_a,
                                const dafne_conv_params* prm_b, const dafne_conv_seg* segs_b, const void* d_wfrag_b,
                                void* d_scratch, size_t scratch_bytes, void* stream) {
    ConvDev A, B;
    int rc = build(A, prm_a, segs_a, false, true);
    if (rc) return rc;
    rc = build(B, prm_b, segs_b, false, true);
    if (rc) return rc;
    if (!d_wfrag_a || !d_wfrag_b) return dafne::fail(DAFNE_E_INVALID, "conv3x3_c256_pair: null fragment-major weights");
    if (!d_scratch || scratch_bytes < (size_t)kRDumpBytes) return dafne::fail(DAFNE_E_WORKSPACE, "conv3x3_c256_pair: scratch %zu < %d", scratch_bytes, kRDumpBytes);
    if (A.flags != B.flags || A.Cout != B.Cout || A.N != B.N || A.ntiles != B.ntiles || A.gn_eps != B.gn_eps)
        return dafne::fail(DAFNE_E_INVALID, "conv3x3_c256_pair: the two layers must agree in flags, Cout, image count and gn_eps");
    A.w = (const char*)d_wfrag_a;
    B.w = (const char*)d_wfrag_b;
    return launch_rp(A, &B, (char*)d_scratch, (hipStream_t)stream);
}

int dafne_conv2d_kernel_id(const dafne_conv_params* prm, const dafne_conv_seg* segs) {
    ConvDev D;
    if (build(D, prm, segs)) return -1;
    if (D.patch) return 6;
    if (D.slab) return pred16_ok(D) ? 8 : 7;
    if (stream_eligible(D)) return ws_eligible(D) ? 5 : 4;
    return D.bn == 256 ? 3 : D.bn == 128 ? 2 : D.bn == 64 ? 1 : 0;
}

int dafne_conv2d_nhwc_bf16_hip(const dafne_conv_params* prm, const dafne_conv_seg* segs, void* stream) {
    ConvDev D;
    int rc = build(D, prm, segs);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (D.patch) return launch_patch(D, st);
    if (D.slab) return launch_slab(D, st);
    if (stream_eligible(D)) {
        if (ws_eligible(D)) return launch_ws(D, st);
        return launch_stream(D, st);
    }
    if (D.bn == 256) return launch<4, 2, 2, 4>(D, st);
    if (D.bn == 128) {
        // at most one tile per CU and a long K loop: the 4-stage ring (one workgroup per CU, no drain per step)
        const char* e = getenv("DAFNE_CONV_RING");          // read per call: the tests run both forms in one process
        int cus = 0;
        if (int rc = dafne::device_cus(&cus)) return rc;
        // DAFNE_CONV_RING: 0 never, 1 always (tests, A/B runs); default: when the caller says the launch is alone on the GPU
        // (+1 % in the serial layout; next to other streams' launches the 128 KB of LDS cost 0.5 %)
        const bool ring = e ? atoi(e) != 0 : (D.flags & DAFNE_CONV_EXCLUSIVE) != 0;
        if (ring && D.mtiles * D.ntiles <= cus && D.ksteps >= 8) return launch<2, 2, 2, 2, 4>(D, st);
        return launch<2, 2, 2, 2>(D, st);
    }
    if (D.bn == 64) return launch<1, 4, 2, 2>(D, st);
    return launch<1, 4, 1, 2>(D, st);
}

int dafne_conv2d_nhwc_fp8w_hip(const dafne_conv_params* prm, const dafne_conv_seg* segs, const float* d_oscale,
                               float in_qscale, void* stream) {
    ConvDev D;
    int rc = build(D, prm, segs, true);
    if (rc) return rc;
    if (!d_oscale) return dafne::fail(DAFNE_E_INVALID, "conv fp8w: null output scale");
    if (!(in_qscale > 0.f)) return dafne::fail(DAFNE_E_INVALID, "conv fp8w: in_qscale must be positive");
    D.oscale = d_oscale;
    D.in_qscale = in_qscale;
    return launch_patch_fp8(D, (hipStream_t)stream);
}

}  // extern "C"
