// Temporary: entry points declared in include/dafne_amd.h whose kernels are not
// written yet.  Each returns DAFNE_E_UNSUPPORTED (never a silent success).
#include "common.h"
extern "C" {
int dafne_conv2d_nhwc_bf16_hip(const dafne_conv_params*, const dafne_conv_seg*, void*) {
    return dafne::fail(DAFNE_E_UNSUPPORTED, "conv: not built yet");
}
int dafne_conv2d_num_tiles(const dafne_conv_params*, const dafne_conv_seg*) { return -1; }
}
