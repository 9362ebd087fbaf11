// Bottleneck tail + next bottleneck head in one kernel (gfx950), for the res4 stage of ResNet-50/101
// [detectron2 BottleneckBlock, recalled; built by build_dafne_resnet_fpn_backbone, backbone/fpn.py:58-91]:
//
//     Y = relu(conv3_b(T) + bias3 + X)            1x1, 256 -> 1024, X = the block's input (identity shortcut)
//     Z = relu(conv1_{b+1}(Y) + bias1)            1x1, 1024 -> 256
//
// Unfused these are two small launches per block (17 GFLOP each at batch 8, 64x64 maps): one round of tiles whose
// duration is prologue + epilogue + operand latency, and Y (67 MB) is written, then read straight back.  Here a
// workgroup owns 128 pixels and ALL channels: for each 256-channel chunk of Y it runs GEMM1 (K = 256, operand T
// resident in LDS), adds bias + residual in place in LDS, stores the chunk to HBM and immediately uses the LDS copy
// as the K-chunk of GEMM2, whose accumulators (128 px x 256 cout) live in registers across the four chunks.  Y is
// never read back; the second launch, its prologue and its epilogue disappear.
//
//   * LDS: T tile 64 KB + Y chunk 64 KB (both as four [128 px][128 B] slabs, 16-byte chunk ^ (px & 7)).  The residual
//     chunk is DMA'd (global_load_lds) into the Y buffer under GEMM1 and updated in place by the epilogue.
//   * No LDS left for a weight ring, so the A operand streams L2 -> REGISTERS: the host packs both weight matrices
//     fragment-major (dafne_b2b_pack layout: [phase][wave][k16 step][lane][8 bf16]), a wave's fragment is one coalesced
//     1-KiB load, 8 fragments are in flight per wave while the previous 8 are consumed.  8 waves = 8 x 32 output
//     channels, each over all 128 pixels (1 A + 4 B fragments per 4 MFMAs).
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
constexpr int kSmem = 2 * kBuf;           // + 5 KB of biases behind it
constexpr int kSmemTotal = kSmem + (1024 + 256) * 4;
constexpr int kCM = 256, kCB = 1024, kChunks = kCB / 256;
constexpr int kPhaseBytes = 8 * 16 * 1024;    // one phase of the fragment-major weights: 8 waves x 16 steps x 1 KB

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

__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    f32x2 v = {a, b};
    bf16x2 r = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(unsigned, r);
}

__global__ void __launch_bounds__(512) conv_b2b_kernel(B2bDev P) {
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
        const int px = (wave + 8 * ii) * 8 + (lane >> 3);
        dpix[ii] = halo_index(px);
        dq[ii] = (unsigned)(((lane & 7) ^ (px & 7)) * 16);
    }
    auto dma_tile = [&](const char* src, unsigned pix_bytes, unsigned col0, int buf) {
#pragma unroll
        for (int sl = 0; sl < 4; sl++)
#pragma unroll
            for (int ii = 0; ii < 2; ii++)
                __builtin_amdgcn_global_load_lds((gvoid*)(src + (size_t)dpix[ii] * pix_bytes + col0 + sl * 128 + dq[ii]),
                                                 (lvoid*)(lds + buf * kBuf + sl * kSlab + (wave + 8 * ii) * 1024), 16, 0, 0);
    };

    // ---- fragment offsets
    unsigned bs[4];                          // B fragment of k16 step s inside a slab, pixel fragment 0
#pragma unroll
    for (int s = 0; s < 4; s++) bs[s] = (unsigned)(frow * 128 + (((2 * s + half) ^ (frow & 7)) * 16));
    unsigned eg[4];                          // epilogue: this lane's 4 channels of group g, pixel fragment 0 (Y buffer)
#pragma unroll
    for (int g = 0; g < 4; g++)
        eg[g] = lds_base + (unsigned)(kBuf + (wave >> 1) * kSlab + frow * 128 + (((((wave & 1) * 4 + g) ^ (frow & 7))) * 16) + 8 * half);

    // ---- A operand: 8 fragments per load group, L2 -> registers through inline asm (the compiler would sink visible
    // loads to their uses and wait for each); readiness is tracked by hand: vmcnt is in-order, every vector-memory
    // instruction of this kernel is issued in a fixed program order, so the number of younger instructions at each
    // wait is a constant (stores of a ragged tile are clamped, not predicated, to keep the count exact).
    const unsigned voff = (unsigned)(wave * 16 * 1024 + lane * 16);
    bf16x8 a0[8], a1[8];
#define B2B_LOAD_A(dst, phase, t0)                                                                              \
    {                                                                                                           \
        const char* sb0 = P.wf + (size_t)(phase) * kPhaseBytes + (t0) * 1024;                                   \
        const char* sb1 = sb0 + 4096;                                                                           \
        asm volatile("global_load_dwordx4 %0, %4, %5\n\tglobal_load_dwordx4 %1, %4, %5 offset:1024\n\t"         \
                     "global_load_dwordx4 %2, %4, %5 offset:2048\n\tglobal_load_dwordx4 %3, %4, %5 offset:3072" \
                     : "=&v"(dst[0]), "=&v"(dst[1]), "=&v"(dst[2]), "=&v"(dst[3])                               \
                     : "v"(voff), "s"(sb0)                                                                      \
                     : "memory");                                                                               \
        asm volatile("global_load_dwordx4 %0, %4, %5\n\tglobal_load_dwordx4 %1, %4, %5 offset:1024\n\t"         \
                     "global_load_dwordx4 %2, %4, %5 offset:2048\n\tglobal_load_dwordx4 %3, %4, %5 offset:3072" \
                     : "=&v"(dst[4]), "=&v"(dst[5]), "=&v"(dst[6]), "=&v"(dst[7])                               \
                     : "v"(voff), "s"(sb1)                                                                      \
                     : "memory");                                                                               \
    }
#define B2B_WAIT_A(dst, n)                                                                                      \
    asm volatile("s_waitcnt vmcnt(" #n ")"                                                                      \
                 : "+v"(dst[0]), "+v"(dst[1]), "+v"(dst[2]), "+v"(dst[3]), "+v"(dst[4]), "+v"(dst[5]), "+v"(dst[6]), \
                   "+v"(dst[7])                                                                                 \
                 :                                                                                              \
                 : "memory")

    f32x16 acc1[4], acc2[4];
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
        for (int k = 0; k < 16; k++) acc2[b][k] = 0.f;

    auto steps = [&](const bf16x8* a, int t0, int buf, f32x16* acc) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int t = t0 + k, sl = t >> 2, s = t & 3;
            bf16x8 bfr[4];
#pragma unroll
            for (int b = 0; b < 4; b++) bfr[b] = *(const bf16x8*)(lds + buf * kBuf + sl * kSlab + b * 4096 + bs[s]);
#pragma unroll
            for (int b = 0; b < 4; b++) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k], bfr[b], acc[b], 0, 0, 0);
        }
    };
    auto barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // acc + bias (+ residual already in the Y buffer) -> ReLU -> bf16, in place in the Y buffer
    const unsigned lbias_off = lds_base + (unsigned)kSmem;      // [1024 conv3 | 256 conv1] fp32
    auto epilogue = [&](const f32x16* acc, int bias0, bool with_res) {
        const unsigned rmask = with_res ? 0xffffffffu : 0u;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            typedef __attribute__((ext_vector_type(4))) float f32x4;
            f32x4 bv;
            u32x2 rc[4];
            const unsigned bad = lbias_off + (unsigned)((bias0 + wave * 32 + 8 * g + 4 * half) * 4);
            // inline asm: a plain LDS read here makes the compiler drain vmcnt (it cannot tell the read from the
            // residual DMA's destination).  The four pixel fragments sit 4096 B apart.
            asm volatile("ds_read_b128 %4, %6\n\tds_read_b64 %0, %5\n\tds_read_b64 %1, %5 offset:4096\n\t"
                         "ds_read_b64 %2, %5 offset:8192\n\tds_read_b64 %3, %5 offset:12288\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(rc[0]), "=&v"(rc[1]), "=&v"(rc[2]), "=&v"(rc[3]), "=&v"(bv)
                         : "v"(eg[g]), "v"(bad)
                         : "memory");
#pragma unroll
            for (int b = 0; b < 4; b++) {
                u32x2 r = rc[b];
                r.x &= rmask;
                r.y &= rmask;
                const float v0 = fmaxf(acc[b][4 * g] + bv.x + bf2f((unsigned short)(r.x & 0xffff)), 0.f);
                const float v1 = fmaxf(acc[b][4 * g + 1] + bv.y + bf2f((unsigned short)(r.x >> 16)), 0.f);
                const float v2 = fmaxf(acc[b][4 * g + 2] + bv.z + bf2f((unsigned short)(r.y & 0xffff)), 0.f);
                const float v3 = fmaxf(acc[b][4 * g + 3] + bv.w + bf2f((unsigned short)(r.y >> 16)), 0.f);
                rc[b].x = pack_bf16(v0, v1);
                rc[b].y = pack_bf16(v2, v3);
            }
            asm volatile("ds_write_b64 %4, %0\n\tds_write_b64 %4, %1 offset:4096\n\tds_write_b64 %4, %2 offset:8192\n\t"
                         "ds_write_b64 %4, %3 offset:12288"
                         :
                         : "v"(rc[0]), "v"(rc[1]), "v"(rc[2]), "v"(rc[3]), "v"(eg[g])
                         : "memory");
        }
    };
    // the Y buffer (256 channels of 128 px) -> global rows: 32 consecutive threads write one pixel's 512 B.
    // EXACTLY 8 stores per lane (vmcnt bookkeeping): rows past the end of a ragged tile re-write the last valid row.
    const int plast = HW - 1 - m0;
    auto store_rows = [&](char* dst, unsigned pix_bytes, unsigned col0) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            int idx = tid + 512 * i;
            asm volatile("" : "+v"(idx));            // recompute the row address at every pass: holding 8 of them spills
            int px = idx >> 5;
            px = px < plast ? px : plast;
            const int j = idx & 31;
            const int sl = j >> 3, q = j & 7;
            u32x4 v;
            const unsigned ad = lds_base + (unsigned)(kBuf + sl * kSlab + px * 128 + ((q ^ (px & 7)) * 16));
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(ad) : "memory");
            char* gp = dst + (size_t)halo_index(px) * pix_bytes + col0 + j * 16;
            *(u32x4*)gp = v;     // compiler-issued: it inserts the wait states a > 64-bit store needs before its data registers are reused
        }
    };

    // ---- prologue: biases -> LDS (plain loads, before any hand-counted load is in flight)
    {
        float* lb = (float*)(lds + kSmem);
        for (int k = tid; k < kCB; k += 512) lb[k] = P.b3[k];
        if (tid < kCM) lb[kCB + tid] = P.b1[tid];
    }
    __syncthreads();
    dma_tile(P.in, kCM * 2, 0, 0);
    dma_tile(P.res, kCB * 2, 0, 1);
    B2B_LOAD_A(a0, 0, 0);
    B2B_WAIT_A(a0, 0);
    barrier();

    // Vector-memory program order per chunk c (loads of 8, L = A fragments, R = residual DMA, ST = stores):
    //   L1 a1<-(2c, 8..15) | L2 a0<-(2c+1, 0..7) | ST chunk c | L3 a1<-(2c+1, 8..15) | L4 a0<-(2c+2, 0..7) | R(c+1)
#pragma unroll
    for (int c = 0; c < kChunks; c++) {
        // GEMM1: Y chunk c = W3[c] . T   (K = 256 over the four slabs of the T tile)
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int k = 0; k < 16; k++) acc1[b][k] = 0.f;
        B2B_LOAD_A(a1, 2 * c, 8);                    // L1
        B2B_WAIT_A(a0, 16);                          // a0 = L4 of the previous chunk; younger: R(c), L1
        steps(a0, 0, 0, acc1);
        __builtin_amdgcn_sched_barrier(0);
        B2B_LOAD_A(a0, 2 * c + 1, 0);                // L2
        B2B_WAIT_A(a1, 8);                           // L1 (and the older residual DMA R(c)) landed; younger: L2
        steps(a1, 8, 0, acc1);
        barrier();                                   // every wave's residual pieces are in the Y buffer
        epilogue(acc1, c * 256, true);
        barrier();
        store_rows(P.out, kCB * 2, (unsigned)c * 512u);      // ST
        __builtin_amdgcn_sched_barrier(0);
        // GEMM2: Z += W1[:, chunk c] . Y chunk
        B2B_LOAD_A(a1, 2 * c + 1, 8);                // L3
        B2B_WAIT_A(a0, 16);                          // L2; younger: ST, L3
        steps(a0, 0, 1, acc2);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < kChunks) {
            B2B_LOAD_A(a0, 2 * c + 2, 0);            // L4
            B2B_WAIT_A(a1, 8);                       // L3; younger: L4
        } else {
            B2B_WAIT_A(a1, 0);
        }
        steps(a1, 8, 1, acc2);
        barrier();                                   // every wave is done reading the Y buffer
        if (c + 1 < kChunks) dma_tile(P.res, kCB * 2, (unsigned)(c + 1) * 512u, 1);      // R(c+1)
        __builtin_amdgcn_sched_barrier(0);
    }
    epilogue(acc2, kCB, false);
    barrier();
    store_rows(P.next, kCM * 2, 0);
#undef B2B_LOAD_A
#undef B2B_WAIT_A
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
    static bool attr_done = false;
    if (!attr_done) {
        DAFNE_HIP_TRY(hipFuncSetAttribute((const void*)conv_b2b_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kSmemTotal));
        attr_done = true;
    }
    hipLaunchKernelGGL(conv_b2b_kernel, dim3(D.tiles), dim3(512), kSmemTotal, (hipStream_t)stream, D);
    return dafne::check_launch("conv_b2b");
}

}  // extern "C"
