// ResizeShortestEdge's pixel resampling on the GPU, bit-exact to Pillow's 8-bit bilinear resize.
//
// Reference path: DotaDatasetMapperTTA (dafne/modeling/tta.py:71-99) -> detectron2 ResizeShortestEdge ->
// ResizeTransform.apply_image, which for uint8 images is `PIL.Image.resize((w, h), BILINEAR)` [detectron2
// v0.5, recalled; Pillow is a third-party dependency, its published algorithm is restated here:
// libImaging/Resample.c -- precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal/Vertical_8bpc].
//   * separable, horizontal pass first into a uint8 intermediate, then the vertical pass;
//   * per output index: center = (i + 0.5) * scale, support = max(scale, 1), taps [int(center - support + .5),
//     int(center + support + .5)) clipped to the image, triangle weights normalised in double precision, then
//     fixed point: k = (int)(+-0.5 + w * 2^22); pixel = clip8((2^21 + sum pix * k) >> 22).
// The coefficients are recomputed per thread in fp64 (this file is built with -ffp-contract=off: same IEEE
// operations as the C code).  Horizontal / vertical flips of the TTA views are folded into the store index.
#include "common.h"

namespace {

constexpr int kPrec = 22;      // PRECISION_BITS = 32 - 8 - 2
constexpr int kMaxTaps = 64;   // downscale factors up to ~31x

struct Taps {
    int xmin, n;
    int k[kMaxTaps];
};

// coefficients of output index i for an axis of in_size -> out_size samples
__device__ __forceinline__ void taps_for(int i, int in_size, int out_size, Taps& t) {
    const double scale = (double)(float)in_size / out_size;      // box = (0, in_size) as floats
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const double center = 0.0 + (i + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    if (xmax > kMaxTaps) xmax = kMaxTaps;       // excluded on the host
    double w[kMaxTaps];
    double ww = 0.0;
    for (int x = 0; x < xmax; x++) {
        double a = (x + xmin - center + 0.5) * ss;
        if (a < 0.0) a = -a;
        const double v = a < 1.0 ? 1.0 - a : 0.0;
        w[x] = v;
        ww += v;
    }
    for (int x = 0; x < xmax; x++) {
        double v = w[x];
        if (ww != 0.0) v /= ww;
        t.k[x] = v < 0 ? (int)(-0.5 + v * (double)(1 << kPrec)) : (int)(0.5 + v * (double)(1 << kPrec));
    }
    t.xmin = xmin;
    t.n = xmax;
}

__device__ __forceinline__ unsigned char clip8(int v) {
    v >>= kPrec;
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: in [C,H,W] (or [H,W,C]) -> tmp [C,H,new_w]
__global__ void __launch_bounds__(256) resize_h_kernel(const unsigned char* __restrict__ in, int hwc, int C, int H, int W,
                                                       int new_w, unsigned char* __restrict__ tmp) {
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    if (xx >= new_w) return;
    Taps t;
    taps_for(xx, W, new_w, t);
    for (int y = blockIdx.y; y < H; y += gridDim.y)
        for (int c = 0; c < C; c++) {
            int acc = 1 << (kPrec - 1);
            for (int x = 0; x < t.n; x++) {
                const int px = hwc ? in[((size_t)y * W + x + t.xmin) * C + c] : in[((size_t)c * H + y) * W + x + t.xmin];
                acc += px * t.k[x];
            }
            tmp[((size_t)c * H + y) * new_w + xx] = clip8(acc);
        }
}

// vertical pass + flips: tmp [C,H,new_w] -> out [C,new_h,new_w]
__global__ void __launch_bounds__(256) resize_v_kernel(const unsigned char* __restrict__ tmp, int C, int H, int new_h, int new_w,
                                                       int hflip, int vflip, unsigned char* __restrict__ out) {
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    const int yy = blockIdx.y;
    if (xx >= new_w || yy >= new_h) return;
    Taps t;
    taps_for(yy, H, new_h, t);
    const int oy = vflip ? new_h - 1 - yy : yy, ox = hflip ? new_w - 1 - xx : xx;
    for (int c = 0; c < C; c++) {
        int acc = 1 << (kPrec - 1);
        for (int y = 0; y < t.n; y++) acc += (int)tmp[((size_t)c * H + y + t.xmin) * new_w + xx] * t.k[y];
        out[((size_t)c * new_h + oy) * new_w + ox] = clip8(acc);
    }
}

}  // namespace

extern "C" {

size_t dafne_resize_workspace_bytes(int C, int H, int new_w) {
    if (C < 1 || H < 1 || new_w < 1) return 0;
    return dafne::align_up((size_t)C * H * new_w, 256);
}

int dafne_resize_bilinear_u8_hip(const uint8_t* d_in, int layout_hwc, int C, int H, int W, int new_h, int new_w,
                                 int hflip, int vflip, uint8_t* d_out, void* d_ws, size_t ws_bytes, void* stream) {
    if (!d_in || !d_out || !d_ws || C < 1 || H < 1 || W < 1 || new_h < 1 || new_w < 1)
        return dafne::fail(DAFNE_E_INVALID, "resize: bad args");
    if (ws_bytes < dafne_resize_workspace_bytes(C, H, new_w))
        return dafne::fail(DAFNE_E_WORKSPACE, "resize: workspace %zu < %zu", ws_bytes, dafne_resize_workspace_bytes(C, H, new_w));
    // tap count of the widest filter: 2 * ceil(support) + 1
    const double sx = (double)W / new_w, sy = (double)H / new_h;
    if ((sx > 1 ? sx : 1) * 2 + 2 > kMaxTaps || (sy > 1 ? sy : 1) * 2 + 2 > kMaxTaps)
        return dafne::fail(DAFNE_E_UNSUPPORTED, "resize: downscale factor above %d", kMaxTaps / 2 - 1);
    hipStream_t st = (hipStream_t)stream;
    unsigned char* tmp = (unsigned char*)d_ws;
    const int gy = H < 1024 ? H : 1024;
    hipLaunchKernelGGL(resize_h_kernel, dim3((new_w + 255) / 256, gy), dim3(256), 0, st, d_in, layout_hwc, C, H, W, new_w, tmp);
    int rc = dafne::check_launch("resize_h");
    if (rc) return rc;
    hipLaunchKernelGGL(resize_v_kernel, dim3((new_w + 255) / 256, new_h), dim3(256), 0, st, tmp, C, H, new_h, new_w, hflip, vflip,
                       d_out);
    return dafne::check_launch("resize_v");
}

}  // extern "C"
