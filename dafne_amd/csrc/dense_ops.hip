// HBM-bound companions of the conv kernel (gfx950): image normalisation into the
// stem layout, 3x3/s2 max-pool, GroupNorm finalize/apply(+ReLU), ReLU copy.
//
// Reference call sites: OneStageDetector.preprocess_image
// (dafne/modeling/one_stage_detector.py:100-107), d2 BasicStem max_pool2d
// [recalled], nn.GroupNorm + nn.ReLU in the head towers
// (dafne/modeling/dafne/dafne.py:330-344), LastLevelP6P7's ReLU (fpn.py:34-36).
// All tensors NHWC bf16 with a 1-pixel zero halo unless noted; 16 bytes per lane.
#include "common.h"

namespace {

__device__ __forceinline__ unsigned short f2bf(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// (x - mean) / std, zero pad to the batch size, 3 -> 4 channels, 3-pixel border
// for the 7x7/s2 stem.  out: [N, Hn+6, Wn+6, 4] bf16 (border and 4th channel 0).
// Round 5: one workgroup per OUTPUT ROW (no 64-bit divisions per pixel), the three 256-entry tables bf16((v - mean) / std) built
// per workgroup in LDS with the expression the per-pixel form evaluated (same bits), a lane owns four consecutive output pixels
// (2 x 16-byte stores) and -- planar images whose rows are dword-aligned -- reads them as two dwords per plane (input x =
// 4g - 3 .. 4g: the last three bytes of dword g - 1 and the first of dword g).  38.8 -> us at batch 8 (100 MB of traffic).
__global__ void __launch_bounds__(256) preprocess_kernel(const unsigned char* __restrict__ img, int hwc,
                                                         int H, int W, const int* __restrict__ valid_hw,
                                                         float m0, float m1, float m2, float s0, float s1,
                                                         float s2, int Hn, int Wn, uint2* __restrict__ out,
                                                         int N, int dword_rows) {
    __shared__ unsigned short lut[3][256];
    {
        const float v = (float)threadIdx.x;
        lut[0][threadIdx.x] = f2bf((v - m0) / s0);
        lut[1][threadIdx.x] = f2bf((v - m1) / s1);
        lut[2][threadIdx.x] = f2bf((v - m2) / s2);
    }
    __syncthreads();
    const int WP = Wn + 6, HP = Hn + 6;
    const int rows = N * HP;
    const int nG = (WP + 3) >> 2;
    const size_t pl = (size_t)H * W;
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const int n = row / HP;
        const int y = row - n * HP - 3;
        const int vh = valid_hw ? valid_hw[2 * n] : H, vw = valid_hw ? valid_hw[2 * n + 1] : W;
        const bool live = y >= 0 && y < vh;
        uint2* orow = out + (size_t)row * WP;
        const bool al16 = ((((size_t)row * WP) & 1) == 0);
        for (int g = threadIdx.x; g < nG; g += 256) {
            uint2 o[4];
#pragma unroll
            for (int j = 0; j < 4; j++) o[j] = make_uint2(0u, 0u);
            if (live) {
                unsigned c[3][4];
                if (dword_rows) {
                    const unsigned char* r = img + ((size_t)n * 3 * H + y) * W;
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) {
                        const unsigned a = (g >= 1 && 4 * g - 4 < W) ? *(const unsigned*)(r + ch * pl + 4 * g - 4) : 0u;
                        const unsigned b = (4 * g < W) ? *(const unsigned*)(r + ch * pl + 4 * g) : 0u;
                        c[ch][0] = (a >> 8) & 0xffu; c[ch][1] = (a >> 16) & 0xffu; c[ch][2] = a >> 24; c[ch][3] = b & 0xffu;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        int x = 4 * g - 3 + j;
                        x = x < 0 ? 0 : (x < W ? x : W - 1);          // clamped: masked below
                        if (hwc) {
                            const unsigned char* p = img + (((size_t)n * H + y) * W + x) * 3;
                            c[0][j] = p[0]; c[1][j] = p[1]; c[2][j] = p[2];
                        } else {
                            const unsigned char* p = img + (size_t)n * 3 * pl + (size_t)y * W + x;
                            c[0][j] = p[0]; c[1][j] = p[pl]; c[2][j] = p[2 * pl];
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int x = 4 * g - 3 + j;
                    if (x >= 0 && x < vw) {
                        o[j].x = (unsigned)lut[0][c[0][j]] | ((unsigned)lut[1][c[1][j]] << 16);
                        o[j].y = (unsigned)lut[2][c[2][j]];
                    }
                }
            }
            if (4 * g + 3 < WP && al16) {
                uint4* d = (uint4*)(orow + 4 * g);
                d[0] = make_uint4(o[0].x, o[0].y, o[1].x, o[1].y);
                d[1] = make_uint4(o[2].x, o[2].y, o[3].x, o[3].y);
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (4 * g + j < WP) orow[4 * g + j] = o[j];
            }
        }
    }
}

// 3x3 stride-2 pad-1 max pool; input is post-ReLU (>= 0) so the zero halo acts as -inf.
__global__ void __launch_bounds__(256) maxpool_kernel(const uint4* __restrict__ in, uint4* __restrict__ out,
                                                      int N, int Hin, int Win, int Hout, int Wout, int C8) {
    const long long total = (long long)N * Hout * Wout * C8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C8);
        long long t = i / C8;
        const int wo = (int)(t % Wout);
        t /= Wout;
        const int ho = (int)(t % Hout);
        const int n = (int)(t / Hout);
        float best[8];
#pragma unroll
        for (int k = 0; k < 8; k++) best[k] = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; dy++)
#pragma unroll
            for (int dx = 0; dx < 3; dx++) {
                // padded coords: orig (2ho-1+dy) -> +1
                const size_t px = ((size_t)n * (Hin + 2) + 2 * ho + dy) * (Win + 2) + 2 * wo + dx;
                const uint4 v = in[px * C8 + c];
                const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    best[2 * k] = fmaxf(best[2 * k], bf2f((unsigned short)(u[k] & 0xffff)));
                    best[2 * k + 1] = fmaxf(best[2 * k + 1], bf2f((unsigned short)(u[k] >> 16)));
                }
            }
        uint4 o;
        o.x = (unsigned)f2bf(best[0]) | ((unsigned)f2bf(best[1]) << 16);
        o.y = (unsigned)f2bf(best[2]) | ((unsigned)f2bf(best[3]) << 16);
        o.z = (unsigned)f2bf(best[4]) | ((unsigned)f2bf(best[5]) << 16);
        o.w = (unsigned)f2bf(best[6]) | ((unsigned)f2bf(best[7]) << 16);
        out[(((size_t)n * (Hout + 2) + ho + 1) * (Wout + 2) + wo + 1) * C8 + c] = o;
    }
}

struct GnSeg {
    char* x;          // [N, H+2, W+2, C] bf16, normalised in place
    int H, W;
    int tile0, tiles_per_img;
};
struct GnDev {
    GnSeg seg[5];
    int n_segs, N, C;
    const float* partial;   // [tiles][C/8][2]
    float* stats;           // [n_segs][N][C/8][2] -> mean, rstd
    const float* gamma;
    const float* beta;
    float eps;
};

// one 1024-thread block per (segment, image): thread = (tile slice 0..31, group 0..31+);
// every slice sums its tiles in order (loads batched 4 deep: the chain of dependent global
// loads was the whole cost of this kernel), then the 32 slices are added in order: the
// result does not depend on launch geometry or timing.
constexpr int kGnSlices = 32;
__global__ void __launch_bounds__(1024) gn_finalize_kernel(GnDev P) {
    const int G = P.C / 8;
    const int n = blockIdx.x % P.N, s = blockIdx.x / P.N;
    const GnSeg& S = P.seg[s];
    __shared__ float part[kGnSlices][32][2];
    const int t0 = S.tile0 + n * S.tiles_per_img;
    for (int g0 = 0; g0 < G; g0 += 32) {
        const int g = g0 + (threadIdx.x & 31), sl = threadIdx.x >> 5;
        float sum = 0.f, sq = 0.f;
        if (g < G) {
            int t = sl;
            for (; t + 3 * kGnSlices < S.tiles_per_img; t += 4 * kGnSlices) {
                float2 p[4];
#pragma unroll
                for (int k = 0; k < 4; k++) p[k] = *(const float2*)(P.partial + ((size_t)(t0 + t + k * kGnSlices) * G + g) * 2);
#pragma unroll
                for (int k = 0; k < 4; k++) { sum += p[k].x; sq += p[k].y; }
            }
            for (; t < S.tiles_per_img; t += kGnSlices) {
                const float2 p = *(const float2*)(P.partial + ((size_t)(t0 + t) * G + g) * 2);
                sum += p.x;
                sq += p.y;
            }
        }
        part[sl][threadIdx.x & 31][0] = sum;
        part[sl][threadIdx.x & 31][1] = sq;
        __syncthreads();
        if (threadIdx.x < 32 && g < G) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int k = 0; k < kGnSlices; k++) {
                a += part[k][threadIdx.x][0];
                b += part[k][threadIdx.x][1];
            }
            const float cnt = (float)(S.H * S.W * 8);
            const float mean = a / cnt;
            float var = b / cnt - mean * mean;
            var = var > 0.f ? var : 0.f;
            float* o = P.stats + (((size_t)s * P.N + n) * G + g) * 2;
            o[0] = mean;
            o[1] = rsqrtf(var + P.eps);
        }
        __syncthreads();
    }
}

struct GnApplyDev {
    long long seg_start[6];   // first flat element (16-byte lane) of each segment
};

// y = relu((x - mean) * rstd * gamma + beta), in place; a 16-byte lane == one group of
// 8 channels; all segments in one launch.
__global__ void __launch_bounds__(256) gn_apply_kernel(GnDev P, GnApplyDev A, long long total) {
    const int G = P.C / 8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        int seg_idx = 0;
#pragma unroll
        for (int k = 1; k < 5; k++)
            if (k < P.n_segs && i >= A.seg_start[k]) seg_idx = k;
        const GnSeg& S = P.seg[seg_idx];
        const long long j = i - A.seg_start[seg_idx];
        const int g = (int)(j % G);
        long long t = j / G;
        const int w = (int)(t % S.W);
        t /= S.W;
        const int h = (int)(t % S.H);
        const int n = (int)(t / S.H);
        const float* st = P.stats + (((size_t)seg_idx * P.N + n) * G + g) * 2;
        const float mean = st[0], rstd = st[1];
        uint4* p = (uint4*)(S.x + ((((size_t)n * (S.H + 2) + h + 1) * (S.W + 2) + w + 1) * P.C + g * 8) * 2);
        const uint4 v = *p;
        const unsigned u[4] = {v.x, v.y, v.z, v.w};
        const float4 ga = *(const float4*)(P.gamma + g * 8), gb = *(const float4*)(P.gamma + g * 8 + 4);
        const float4 ba = *(const float4*)(P.beta + g * 8), bb = *(const float4*)(P.beta + g * 8 + 4);
        const float gam[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
        const float bet[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w};
        unsigned short r[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float x = bf2f((unsigned short)(k & 1 ? u[k >> 1] >> 16 : u[k >> 1] & 0xffff));
            const float y = (x - mean) * rstd * gam[k] + bet[k];
            r[k] = f2bf(fmaxf(y, 0.f));
        }
        uint4 o;
        o.x = (unsigned)r[0] | ((unsigned)r[1] << 16);
        o.y = (unsigned)r[2] | ((unsigned)r[3] << 16);
        o.z = (unsigned)r[4] | ((unsigned)r[5] << 16);
        o.w = (unsigned)r[6] | ((unsigned)r[7] << 16);
        *p = o;
    }
}

__global__ void __launch_bounds__(256) relu_copy_kernel(const uint4* __restrict__ in, uint4* __restrict__ out,
                                                        long long n16) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) {
        uint4 v = in[i];
        unsigned* u = (unsigned*)&v;
#pragma unroll
        for (int k = 0; k < 4; k++) {   // bf16 sign bits: negative -> 0
            unsigned lo = u[k] & 0xffffu, hi = u[k] >> 16;
            if (lo & 0x8000u) lo = 0;
            if (hi & 0x8000u) hi = 0;
            u[k] = lo | (hi << 16);
        }
        out[i] = v;
    }
}

inline unsigned grid_for(long long total) {
    long long b = (total + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 256 * 8 ? 256 * 8 : b));
}

}  // namespace

extern "C" {

int dafne_preprocess_image_hip(const uint8_t* d_img, int layout_hwc, int n_images, int H, int W,
                               const int32_t* d_valid_hw, const float* mean3, const float* std3, int Hn,
                               int Wn, void* d_out, void* stream) {
    if (!d_img || !d_out || !mean3 || !std3 || n_images < 1 || H < 1 || W < 1 || Hn < H || Wn < W)
        return dafne::fail(DAFNE_E_INVALID, "preprocess: bad args");
    const long long rows = (long long)n_images * (Hn + 6);
    if (rows > (1ll << 30)) return dafne::fail(DAFNE_E_UNSUPPORTED, "preprocess: too many rows");
    // dword reads of planar rows: every row of every plane starts on a 4-byte boundary
    const int dword_rows = !layout_hwc && (W % 4) == 0 && (((size_t)d_img) & 3) == 0;
    hipLaunchKernelGGL(preprocess_kernel, dim3((unsigned)(rows < 8192 ? rows : 8192)), dim3(256), 0, (hipStream_t)stream, d_img,
                       layout_hwc, H, W, d_valid_hw, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], Hn,
                       Wn, (uint2*)d_out, n_images, dword_rows);
    return dafne::check_launch("preprocess");
}

int dafne_maxpool3x3s2_nhwc_bf16_hip(const void* d_in, void* d_out, int n_images, int Hin, int Win, int C,
                                     void* stream) {
    if (!d_in || !d_out || n_images < 1 || Hin < 2 || Win < 2 || (C % 8) || ((Hin | Win) & 1))
        return dafne::fail(DAFNE_E_INVALID, "maxpool: bad args (even H/W, C %% 8 == 0)");
    const int Hout = Hin / 2, Wout = Win / 2;
    const long long total = (long long)n_images * Hout * Wout * (C / 8);
    hipLaunchKernelGGL(maxpool_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const uint4*)d_in,
                       (uint4*)d_out, n_images, Hin, Win, Hout, Wout, C / 8);
    return dafne::check_launch("maxpool");
}

int dafne_groupnorm_finalize_hip(const dafne_gn_seg* segs, int n_segs, int n_images, int C, const float* d_partial,
                                 float* d_stats, float eps, void* stream) {
    if (!segs || n_segs < 1 || n_segs > 5 || n_images < 1 || (C % 8) || !d_partial || !d_stats)
        return dafne::fail(DAFNE_E_INVALID, "groupnorm_finalize: bad args");
    GnDev D;
    D.n_segs = n_segs; D.N = n_images; D.C = C; D.partial = d_partial; D.stats = d_stats;
    D.gamma = nullptr; D.beta = nullptr; D.eps = eps;
    for (int s = 0; s < n_segs; s++) {
        if (segs[s].H < 1 || segs[s].W < 1) return dafne::fail(DAFNE_E_INVALID, "groupnorm_finalize: segment %d", s);
        D.seg[s].x = (char*)segs[s].d_x; D.seg[s].H = segs[s].H; D.seg[s].W = segs[s].W;
        D.seg[s].tile0 = segs[s].tile0; D.seg[s].tiles_per_img = segs[s].tiles_per_img;
    }
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(n_segs * n_images), dim3(1024), 0, (hipStream_t)stream, D);
    return dafne::check_launch("gn_finalize");
}

int dafne_groupnorm_relu_nhwc_bf16_hip(const dafne_gn_seg* segs, int n_segs, int n_images, int C,
                                       const float* d_partial, float* d_stats, const float* d_gamma,
                                       const float* d_beta, float eps, void* stream) {
    if (!segs || n_segs < 1 || n_segs > 5 || n_images < 1 || (C % 8) || !d_partial || !d_stats || !d_gamma || !d_beta)
        return dafne::fail(DAFNE_E_INVALID, "groupnorm: bad args");
    GnDev D;
    D.n_segs = n_segs; D.N = n_images; D.C = C; D.partial = d_partial; D.stats = d_stats;
    D.gamma = d_gamma; D.beta = d_beta; D.eps = eps;
    for (int s = 0; s < n_segs; s++) {
        if (!segs[s].d_x || segs[s].H < 1 || segs[s].W < 1) return dafne::fail(DAFNE_E_INVALID, "groupnorm: segment %d", s);
        D.seg[s].x = (char*)segs[s].d_x; D.seg[s].H = segs[s].H; D.seg[s].W = segs[s].W;
        D.seg[s].tile0 = segs[s].tile0; D.seg[s].tiles_per_img = segs[s].tiles_per_img;
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(n_segs * n_images), dim3(1024), 0, st, D);
    int rc = dafne::check_launch("gn_finalize");
    if (rc) return rc;
    GnApplyDev A;
    long long total = 0;
    for (int s = 0; s < 6; s++) {
        A.seg_start[s] = total;
        if (s < n_segs) total += (long long)n_images * segs[s].H * segs[s].W * (C / 8);
    }
    hipLaunchKernelGGL(gn_apply_kernel, dim3(grid_for(total)), dim3(256), 0, st, D, A, total);
    return dafne::check_launch("gn_apply");
}

int dafne_relu_copy_bf16_hip(const void* d_in, void* d_out, int64_t n_elems, void* stream) {
    if (!d_in || !d_out || n_elems < 0 || (n_elems % 8)) return dafne::fail(DAFNE_E_INVALID, "relu_copy: bad args");
    if (n_elems == 0) return DAFNE_OK;
    hipLaunchKernelGGL(relu_copy_kernel, dim3(grid_for(n_elems / 8)), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)d_in, (uint4*)d_out, (long long)(n_elems / 8));
    return dafne::check_launch("relu_copy");
}

}  // extern "C"
