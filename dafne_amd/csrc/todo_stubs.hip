// Temporary: entry points declared in include/dafne_amd.h whose kernels are not
// written yet.  Each returns DAFNE_E_UNSUPPORTED (never a silent success).
#include "common.h"
extern "C" {
size_t dafne_decode_workspace_bytes(const dafne_decode_params*, const dafne_level_desc*) { return 0; }
int dafne_decode_levels_hip(const dafne_decode_params*, const dafne_level_desc*, float*, float*, float*, int32_t*,
                            float*, int32_t*, float*, int32_t*, void*, size_t, void*) {
    return dafne::fail(DAFNE_E_UNSUPPORTED, "decode: not built yet");
}
int dafne_sort_quadrilateral_hip(const float*, float*, int64_t, void*) {
    return dafne::fail(DAFNE_E_UNSUPPORTED, "sort_quadrilateral: not built yet");
}
int dafne_gather_detections_hip(const float*, const float*, const float*, const int32_t*, const float*,
                                const int32_t*, const float*, const int64_t*, const int32_t*, const float*, int,
                                int, int, int, float*, int32_t*, void*) {
    return dafne::fail(DAFNE_E_UNSUPPORTED, "gather: not built yet");
}
int dafne_conv2d_nhwc_bf16_hip(const dafne_conv_params*, const dafne_conv_seg*, void*) {
    return dafne::fail(DAFNE_E_UNSUPPORTED, "conv: not built yet");
}
int dafne_conv2d_num_tiles(const dafne_conv_params*, const dafne_conv_seg*) { return -1; }
}
