// Small-M convolutions (res5, the top FPN levels, stride-2 projections: a few thousand output pixels, thousands of K) as
// 128-pixel x 256-channel tiles with the weights streamed L2 -> REGISTERS and a deterministic split-K (gfx950).
// [detectron2 BottleneckBlock / FPN, recalled; built by build_dafne_resnet_fpn_backbone, backbone/fpn.py:58-91;
//  LastLevelP6P7, backbone/fpn.py:16-37]
//
// Why: conv_igemm_kernel<2,2,2,2> stages BOTH operands of a 128 x 128 tile through LDS -- 32 KB of L2 -> LDS ingest per
// 2.1 MFLOP K step -- and these layers have at most one such tile per CU: each CU runs at the ingest rate of ONE CU
// (res5 conv2: 0.67 us per K step against 0.24 us of MFMA issue) while the launch has 64 pixel tiles for 256 CUs.
// Here a workgroup (8 waves, one per CU) owns 128 pixels x 256 output channels over a SLICE of K:
//   * a wave owns 32 output channels x all 128 pixels: its 1-KiB weight fragment of a k16 step goes L2 -> registers
//     (fragment-major packing, ring of 8 steps, counted vmcnt) and feeds 4 MFMAs; nothing but the 16-KB pixel stage
//     of a K64 step passes through LDS: 48 KB of ingest per 4.2 MFLOP (a third of the 128 x 128 tile's per flop);
//   * the pixel operand is im2col by DMA as in conv_igemm (haloed NHWC input: every tap of every pixel is a plain
//     in-bounds 128-byte segment), 16-byte chunk XOR by pixel -> conflict-free ds_read_b128, ring of three stages,
//     ONE barrier per K64 step (16 MFMAs per wave between barriers);
//   * split-K: S workgroups share a tile, each writes its fp32 partial slab (128 KB, accumulator layout, coalesced
//     1-KiB stores), release + ticket; the LAST arriver acquires, sums the S slabs IN SLICE ORDER (a fixed order:
//     results do not depend on which workgroup is last) and runs the epilogue -- bias, residual / top-down add, ReLU,
//     bf16, 16-byte row stores.  S = 1: no exchange; K order and epilogue expressions are conv_igemm_kernel's, the
//     output is bit-identical to it.  S > 1: the fp32 sum is grouped by slices (fp32 rounding, <= 1 bf16 ulp apart).
#include <type_traits>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((address_space(1))) void gvoid;
typedef __attribute__((address_space(3))) void lvoid;

constexpr int kPx = 128, kCT = 256;                 // tile: pixels x output channels
constexpr int kNominalBatch = 8;                    // images the split-K slice count is sized for (the headline batch; 4 / 6 / 8 measured
                                                    // alike in the timed layout, 8 is 1.8 % faster for a whole batch of 8 on one stream)
constexpr int kNW = 8, kNT = 512;
constexpr int kPF = kPx / 32;                       // pixel fragments per wave
constexpr int kStage = kPx * 128;                   // one K64 step of the pixel operand: [128 px][128 B]
constexpr int kNS = 3;                              // stage ring
constexpr int kSlabY = kPx * 128;                   // epilogue buffer: [4 slabs of 64 ch][128 px][128 B]
constexpr int kOffFlag = 4 * kSlabY;                // ticket broadcast
constexpr int kSmemTotal = kOffFlag + 64;
constexpr int kSlabBytes = kPx * kCT * 4;           // one split-K partial: 128 KB
constexpr int kCounterBytes = 64 * 1024;            // workspace head: arrival tickets of up to 16384 tiles (a FIXED size: calls of
                                                    // different shapes share one workspace and the tickets must stay zero)
static_assert(kNS * kStage <= 4 * kSlabY, "stages alias the epilogue buffer");

struct WrDev {
    const char* in;      // bf16 [N, Hin+2, Win+2, Cin], zero halo
    const char* wf;      // bf16 [Cout/256][8 waves][K/16][64 lanes][8]
    const float* bias;   // [Cout]
    const char* res;     // bf16 [N, Hout+2, Wout+2, Cout] (RESIDUAL) or [N, Hout/2+2, Wout/2+2, Cout] (UPSAMPLE_ADD) or null
    char* out;           // bf16 [N, Hout+2, Wout+2, Cout], interior written
    char* slabs;         // split-K partials [tiles][S][128 KB]
    int* counters;       // [tiles] arrival tickets, zero between launches
    int N, Hin, Win, Hout, Wout, Cin, Cout, KK, KW, stride, pad;
    int nsteps;          // K64 steps: KH * KW * Cin / 64
    int ptiles, ctiles, S;
    int total_px;        // N * Hout * Wout
    unsigned flags;
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

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Vector-memory program order of a wave (P = the wave's two DMA pieces of a pixel stage, W = one weight fragment):
//   prologue  P(0) | W0..W3 P(1) | W4..W7 P(2)
//   step i    [W(4i+8) behind sub-step 0] [W(4i+9)] [W(4i+10)] [W(4i+11) P(i+3) behind sub-step 3]
// vmcnt retires in order, so "W(4i+s) has landed" is `vmcnt(11)` at every (i, s) -- behind it in the queue: the rest of
// its own step group (3 - s fragments + 2 pieces), the 6 operations of the next group and the s fragments of this step --
// and "my pieces of stage i+1 have landed" is `vmcnt(9)` in front of sub-step 3.  Steps past the end of the slice re-load
// the last step (dead stages / ring slots): the counts stay constant, no wait ever drains the queue.
__global__ void __launch_bounds__(512, 2) conv_wr_kernel(WrDev P) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, half = lane >> 5;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;

    // block -> (tile, slice): a tile's slices are neighbours (same XCD after the remap); tiles of one channel tile are
    // neighbours (they re-read the same weights from that XCD's L2)
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int slice = id % P.S;
    const int t = id / P.S;
    const int pt = t % P.ptiles, ct = t / P.ptiles;
    const int k_begin = (int)((long long)slice * P.nsteps / P.S);
    const int k_end = (int)((long long)(slice + 1) * P.nsteps / P.S);
    const int nst = k_end - k_begin;

    const int Wpi = P.Win + 2;
    const int HW = P.Hout * P.Wout;
    // tile pixel px <-> flat output pixel pt * 128 + px over (image, row, column); past the end: clamped (loads), dropped (stores)
    auto out_coords = [&](int px, int& img, int& oy, int& ox) {
        int gp = pt * kPx + px;
        gp = gp < P.total_px ? gp : P.total_px - 1;
        img = gp / HW;                                 // (a handful of divisions per lane and launch)
        const int m = gp - img * HW;
        oy = m / P.Wout;
        ox = m - oy * P.Wout;
    };

    // ---- pixel DMA map: a stage is 16 pieces of 8 px x 128 B; wave w moves pieces w and w + 8
    unsigned pofs[2];
#pragma unroll
    for (int ii = 0; ii < 2; ii++) {
        const int px = (wave + kNW * ii) * 8 + (lane >> 3);
        int img, oy, ox;
        out_coords(px, img, oy, ox);
        const unsigned g = (unsigned)((img * (P.Hin + 2) + oy * P.stride + 1 - P.pad) * Wpi + ox * P.stride + 1 - P.pad);
        pofs[ii] = g * (unsigned)(P.Cin * 2) + (unsigned)(((lane & 7) ^ ((px >> 1) & 7)) * 16);
    }
    auto step_off = [&](int g) -> unsigned {           // byte offset of K64 step g = (64-channel slab, kh, kw)
        if (P.KK == 1) return (unsigned)g * 128u;
        const int c = g / P.KK, tp = g - c * P.KK;
        const int kh = tp / P.KW, kw = tp - kh * P.KW;
        return (unsigned)(((kh * Wpi + kw) * P.Cin + c * 64) * 2);
    };
    auto issue_P = [&](int i, int stage) {             // i: step relative to k_begin (clamped)
        const int g = k_begin + (i < nst ? i : nst - 1);
        const unsigned so = step_off(g);
        __builtin_amdgcn_global_load_lds((gvoid*)(P.in + (size_t)(pofs[0] + so)), (lvoid*)(lds + stage * kStage + wave * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gvoid*)(P.in + (size_t)(pofs[1] + so)), (lvoid*)(lds + stage * kStage + (wave + kNW) * 1024), 16, 0, 0);
    };

    // ---- A operand: fragment-major weights, L2 -> registers
    const int k16_total = P.nsteps * 4;
    const char* wbase = P.wf + ((size_t)(ct * kNW + wave) * k16_total) * 1024;
    const unsigned voff = (unsigned)(lane * 16);
    bf16x8 ar[8];
#pragma unroll
    for (int k = 0; k < 8; k++) ar[k] = bf16x8{};
    auto issue_W = [&](int j, bf16x8& dst) {           // j: k16 step relative to 4 * k_begin (clamped)
        const int jj = j < 4 * nst ? j : 4 * nst - 1;
        const char* sb = wbase + (size_t)(4 * k_begin + jj) * 1024;
        // "+v": the destination is loop-carried; an output-only operand would let the compiler copy it right behind the
        // outstanding load
        asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(dst) : "v"(voff), "s"(sb) : "memory");
    };

    // ---- B fragments: k16 sub-step s of a [128 px][128 B] stage; pixel fragments sit 4096 B apart
    unsigned bs[4];
#pragma unroll
    for (int s = 0; s < 4; s++) bs[s] = (unsigned)(frow * 128 + (((2 * s + half) ^ ((frow >> 1) & 7)) * 16));
    bf16x8 bfr[2][kPF];
    auto bread = [&](int stage, int s, bf16x8 (&b)[kPF]) {
        const unsigned ad = lds_base + (unsigned)(stage * kStage) + bs[s];
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:4096\n\tds_read_b128 %2, %4 offset:8192\n\tds_read_b128 %3, %4 offset:12288"
                     : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3])
                     : "v"(ad)
                     : "memory");
    };
    auto barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    f32x16 acc[kPF];
#pragma unroll
    for (int b = 0; b < kPF; b++)
#pragma unroll
        for (int k = 0; k < 16; k++) acc[b][k] = 0.f;

    // ---- prologue
    issue_P(0, 0);
    static_for<0, 4>([&](auto J) { issue_W(decltype(J)::value, ar[decltype(J)::value]); });
    issue_P(1, 1);
    static_for<4, 8>([&](auto J) { issue_W(decltype(J)::value, ar[decltype(J)::value]); });
    issue_P(2, 2);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");          // my two pieces of stage 0 (behind them: 8 fragments + 4 pieces)
    barrier();
    bread(0, 0, bfr[0]);

    // ---- K loop over the slice's K64 steps, unrolled by two (the weight ring's slots are compile-time registers)
    int cur = 0;                                               // stage of step i
    auto step = [&](int i, auto PAR) {
        constexpr int par = decltype(PAR)::value;
        const int nxt = cur == kNS - 1 ? 0 : cur + 1;
        static_for<0, 4>([&](auto SS) {
            constexpr int s = decltype(SS)::value;
            if constexpr (s < 3) {
                asm volatile("s_waitcnt vmcnt(11)" : "+v"(ar[par * 4 + s]) :: "memory");
                bread(cur, s + 1, bfr[(s + 1) & 1]);
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(bfr[s & 1][0]), "+v"(bfr[s & 1][1]), "+v"(bfr[s & 1][2]), "+v"(bfr[s & 1][3]) :: "memory");
            } else {
                // sub-step 3: its B fragments are in registers, so every wave is done READING stage `cur` once it has passed this
                // barrier; behind it stage i+1 (mine landed: vmcnt(9) also covers W(4i+3); everybody's: the barrier) is readable
                // and stage `cur` may be overwritten by step i+3
                asm volatile("s_waitcnt vmcnt(9)" : "+v"(ar[par * 4 + s]) :: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bfr[1][0]), "+v"(bfr[1][1]), "+v"(bfr[1][2]), "+v"(bfr[1][3]) :: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                bread(nxt, 0, bfr[0]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < kPF; b++) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[par * 4 + s], bfr[s & 1][b], acc[b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            issue_W(4 * i + 8 + s, ar[par * 4 + s]);
            if constexpr (s == 3) issue_P(i + 3, cur);
        });
        cur = nxt;
    };
    int i = 0;
    for (; i + 1 < nst; i += 2) {
        step(i, std::integral_constant<int, 0>{});
        step(i + 1, std::integral_constant<int, 1>{});
    }
    if (i < nst) step(i, std::integral_constant<int, 0>{});
    // the re-loads past the end of the slice are still in flight INTO ar[] / bfr[0]: both are dead for the compiler, which would
    // hand their registers to the next temporary (an address, a division) while the loads can still land on them -- the
    // operands keep them allocated until the queues are empty
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                 : "+v"(ar[0]), "+v"(ar[1]), "+v"(ar[2]), "+v"(ar[3]), "+v"(ar[4]), "+v"(ar[5]), "+v"(ar[6]), "+v"(ar[7]),
                   "+v"(bfr[0][0]), "+v"(bfr[0][1]), "+v"(bfr[0][2]), "+v"(bfr[0][3])
                 :: "memory");
    barrier();                                                 // LDS is free

    // ---- split-K: partial slab out, ticket; the last arriver sums the slabs in slice order
    if (P.S > 1) {
        char* my = P.slabs + ((size_t)t * P.S + slice) * kSlabBytes;
#pragma unroll
        for (int b = 0; b < kPF; b++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const f32x4 v = {acc[b][4 * q], acc[b][4 * q + 1], acc[b][4 * q + 2], acc[b][4 * q + 3]};
                // write-through (sc1) stores: the slab is published by the stores themselves -- a release fence would write
                // back EVERY dirty line of this XCD's L2, once per workgroup (measured: 57 -> us for res5 conv2 at batch 8)
                char* a = my + ((size_t)((wave * 16 + b * 4 + q) * 64 + lane)) * 16;
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(a), "v"(v) : "memory");
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const int ticket = __hip_atomic_fetch_add(P.counters + t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *(volatile int*)(lds + kOffFlag) = ticket;
        }
        __syncthreads();
        const int ticket = *(volatile int*)(lds + kOffFlag);
        if (ticket != P.S - 1) return;
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(P.counters + t, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // zero for the next launch
        }
        __syncthreads();
        for (int s = 0; s < P.S; s++) {
            const char* sl = P.slabs + ((size_t)t * P.S + s) * kSlabBytes;
            f32x4 v[kPF * 4];
#pragma unroll
            for (int k = 0; k < kPF * 4; k++) v[k] = *(const f32x4*)(sl + ((size_t)((wave * 16 + k) * 64 + lane)) * 16);
#pragma unroll
            for (int b = 0; b < kPF; b++)
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int r = 0; r < 4; r++) acc[b][4 * q + r] = s == 0 ? v[b * 4 + q][r] : acc[b][4 * q + r] + v[b * 4 + q][r];
        }
    }

    // ---- epilogue: (acc + bias) + residual -> ReLU -> bf16, in place in the [4 slabs][128 px][128 B] buffer
    const bool relu = P.flags & DAFNE_CONV_RELU;
    const float relu_lo = relu ? 0.f : -__builtin_inff();
    const bool has_res = P.flags & DAFNE_CONV_RESIDUAL, has_up = P.flags & DAFNE_CONV_UPSAMPLE_ADD;
    if (has_res || has_up) {
        // residual rows by DMA: a slab is 16 pieces of 8 px x 128 B; wave w moves pieces w and w + 8 of every slab
#pragma unroll
        for (int ii = 0; ii < 2; ii++) {
            const int px = (wave + kNW * ii) * 8 + (lane >> 3);
            int img, oy, ox;
            out_coords(px, img, oy, ox);
            const size_t rpix = has_up ? ((size_t)(img * (P.Hout / 2 + 2) + oy / 2 + 1) * (P.Wout / 2 + 2) + ox / 2 + 1)
                                       : ((size_t)(img * (P.Hout + 2) + oy + 1) * (P.Wout + 2) + ox + 1);
            const char* src = P.res + (rpix * P.Cout + ct * kCT) * 2 + ((lane & 7) ^ ((px >> 1) & 7)) * 16;
#pragma unroll
            for (int sl = 0; sl < 4; sl++)
                __builtin_amdgcn_global_load_lds((gvoid*)(src + sl * 128), (lvoid*)(lds + sl * kSlabY + (wave + kNW * ii) * 1024), 16, 0, 0);
        }
    }
    f32x4 bia[4];
#pragma unroll
    for (int g = 0; g < 4; g++) bia[g] = *(const f32x4*)(P.bias + ct * kCT + wave * 32 + 8 * g + 4 * half);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
        const unsigned rmask = (has_res || has_up) ? 0xffffffffu : 0u;
        char* ebase = lds + (wave >> 1) * kSlabY + frow * 128 + 8 * half;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            char* ea = ebase + ((((wave & 1) * 4 + g) ^ ((frow >> 1) & 7)) * 16);
#pragma unroll
            for (int b = 0; b < kPF; b++) {
                u32x2 r = *(const u32x2*)(ea + b * 4096);
                r.x &= rmask;
                r.y &= rmask;
                const float v0 = fmaxf((acc[b][4 * g] + bia[g][0]) + __uint_as_float(r.x << 16), relu_lo);       // (a max with 0 / -inf: no branch)
                const float v1 = fmaxf((acc[b][4 * g + 1] + bia[g][1]) + __uint_as_float(r.x & 0xffff0000u), relu_lo);
                const float v2 = fmaxf((acc[b][4 * g + 2] + bia[g][2]) + __uint_as_float(r.y << 16), relu_lo);
                const float v3 = fmaxf((acc[b][4 * g + 3] + bia[g][3]) + __uint_as_float(r.y & 0xffff0000u), relu_lo);
                u32x2 o;
                o.x = pack_bf16(v0, v1);
                o.y = pack_bf16(v2, v3);
                *(u32x2*)(ea + b * 4096) = o;
            }
        }
    }
    __syncthreads();
    // rows out: 8 consecutive threads write one pixel's 128 B of a slab
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int px = h * 64 + (tid >> 3);
        const int q = tid & 7;
        if (pt * kPx + px < P.total_px) {
            int img, oy, ox;
            out_coords(px, img, oy, ox);
            char* dst = P.out + (((size_t)(img * (P.Hout + 2) + oy + 1) * (P.Wout + 2) + ox + 1) * P.Cout + ct * kCT) * 2 + q * 16;
#pragma unroll
            for (int sl = 0; sl < 4; sl++) {
                const u32x4 v = *(const u32x4*)(lds + sl * kSlabY + px * 128 + ((q ^ ((px >> 1) & 7)) * 16));
                *(u32x4*)(dst + sl * 128) = v;
            }
        }
    }
}

int wr_build(WrDev& D, const dafne_conv_params* prm, const dafne_conv_seg* segs, const char** why) {
    *why = nullptr;
    if (!prm || !segs) { *why = "null argument"; return 0; }
    const unsigned allowed = DAFNE_CONV_RELU | DAFNE_CONV_RESIDUAL | DAFNE_CONV_UPSAMPLE_ADD | DAFNE_CONV_EXCLUSIVE;
    if (prm->n_segs != 1) { *why = "one segment"; return 0; }
    if (prm->flags & ~allowed) { *why = "flags"; return 0; }
    if ((prm->flags & DAFNE_CONV_RESIDUAL) && (prm->flags & DAFNE_CONV_UPSAMPLE_ADD)) { *why = "residual and top-down add"; return 0; }
    if (!((prm->KH == 1 && prm->KW == 1 && prm->pad == 0) || (prm->KH == 3 && prm->KW == 3 && prm->pad == 1))) { *why = "1x1 p0 or 3x3 p1"; return 0; }
    if (prm->stride != 1 && prm->stride != 2) { *why = "stride"; return 0; }
    if (prm->Cin < 64 || prm->Cin % 64 || prm->Cout < kCT || prm->Cout % kCT) { *why = "Cin % 64, Cout % 256"; return 0; }
    if (!prm->d_bias) { *why = "bias"; return 0; }
    const dafne_conv_seg& S = segs[0];
    if (prm->n_images < 1 || S.Hout < 1 || S.Wout < 1 || (long long)S.Hout * S.Wout > (1 << 20)) { *why = "size"; return 0; }
    if (S.Hout != (S.Hin + 2 * prm->pad - prm->KH) / prm->stride + 1 || S.Wout != (S.Win + 2 * prm->pad - prm->KW) / prm->stride + 1) { *why = "geometry"; return 0; }
    if ((prm->flags & DAFNE_CONV_UPSAMPLE_ADD) && ((S.Hout & 1) || (S.Wout & 1))) { *why = "top-down add on an odd map"; return 0; }
    const long long in_bytes = (long long)prm->n_images * (S.Hin + 2) * (S.Win + 2) * prm->Cin * 2;
    const long long total = (long long)prm->n_images * S.Hout * S.Wout;
    if (in_bytes > 0xffffffffll || total > (1ll << 21)) { *why = "too large"; return 0; }
    D.in = (const char*)S.d_in; D.out = (char*)S.d_out; D.res = (const char*)S.d_res;
    D.bias = (const float*)prm->d_bias;
    D.N = prm->n_images; D.Hin = S.Hin; D.Win = S.Win; D.Hout = S.Hout; D.Wout = S.Wout;
    D.Cin = prm->Cin; D.Cout = prm->Cout; D.KK = prm->KH * prm->KW; D.KW = prm->KW; D.stride = prm->stride; D.pad = prm->pad;
    D.nsteps = D.KK * (D.Cin / 64);
    D.total_px = (int)total;
    D.ptiles = (int)((total + kPx - 1) / kPx);
    D.ctiles = D.Cout / kCT;
    D.flags = prm->flags;
    if ((long long)D.ptiles * D.ctiles > kCounterBytes / 4) { *why = "too many tiles"; return 0; }
    // slices: fill the chip, at least 4 K64 steps per slice, at most 8 slabs for the reducer.  The slice count enters the fp32
    // grouping of the sum, so it is a function of the PER-IMAGE shape and the CU count only: not of the EXCLUSIVE hint (a batch
    // gives the same bits whether its plan has the GPU to itself or shares it) and not of the batch size (an image gives the same
    // bits whichever images share its batch: the TTA wrapper's grouped views, tests/test_gpu_model.py).  It is sized for the
    // headline batch (kNominalBatch images); other batch sizes get that count.
    int cus = 0;
    if (dafne::device_cus(&cus)) cus = 256;
    const long long px_nom = (long long)kNominalBatch * S.Hout * S.Wout;
    const int tiles = (int)((px_nom + kPx - 1) / kPx) * D.ctiles;
    int s = cus / tiles;
    if (s > D.nsteps / 4) s = D.nsteps / 4;
    if (s > 8) s = 8;
    if (s < 1) s = 1;
    D.S = s;
    return 1;
}

}  // namespace

extern "C" {

int dafne_conv2d_wr_ok(const dafne_conv_params* prm, const dafne_conv_seg* segs) {
    WrDev D;
    const char* why;
    return wr_build(D, prm, segs, &why);
}

int dafne_conv2d_wr_splits(const dafne_conv_params* prm, const dafne_conv_seg* segs) {
    WrDev D;
    const char* why;
    return wr_build(D, prm, segs, &why) ? D.S : 0;
}

size_t dafne_conv2d_wr_workspace_bytes(const dafne_conv_params* prm, const dafne_conv_seg* segs) {
    WrDev D;
    const char* why;
    if (!wr_build(D, prm, segs, &why)) return 0;
    const size_t tiles = (size_t)D.ptiles * D.ctiles;
    return D.S > 1 ? (size_t)kCounterBytes + tiles * D.S * (size_t)kSlabBytes : 0;
}

int dafne_conv2d_wr_hip(const dafne_conv_params* prm, const dafne_conv_seg* segs, const void* d_wfrag, void* d_workspace,
                        size_t workspace_bytes, void* stream) {
    WrDev D;
    const char* why;
    if (!wr_build(D, prm, segs, &why)) return dafne::fail(why && why[0] == 'n' ? DAFNE_E_INVALID : DAFNE_E_UNSUPPORTED, "conv2d_wr: %s", why ? why : "?");
    if (!d_wfrag || !D.in || !D.out) return dafne::fail(DAFNE_E_INVALID, "conv2d_wr: null argument");
    if ((D.flags & (DAFNE_CONV_RESIDUAL | DAFNE_CONV_UPSAMPLE_ADD)) && !D.res) return dafne::fail(DAFNE_E_INVALID, "conv2d_wr: residual flag without d_res");
    const size_t tiles = (size_t)D.ptiles * D.ctiles;
    const size_t need = D.S > 1 ? (size_t)kCounterBytes + tiles * D.S * (size_t)kSlabBytes : 0;
    if (D.S > 1 && (!d_workspace || workspace_bytes < need)) return dafne::fail(DAFNE_E_WORKSPACE, "conv2d_wr: workspace %zu < %zu", workspace_bytes, need);
    D.wf = (const char*)d_wfrag;
    D.counters = (int*)d_workspace;
    D.slabs = (char*)d_workspace + kCounterBytes;
    DAFNE_MAX_LDS_ONCE(kSmemTotal, (const void*)conv_wr_kernel);
    hipLaunchKernelGGL(conv_wr_kernel, dim3((unsigned)(tiles * D.S)), dim3(kNT), kSmemTotal, (hipStream_t)stream, D);
    return dafne::check_launch("conv_wr");
}

}  // extern "C"
