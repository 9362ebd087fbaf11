// ABI plumbing shared by every translation unit of libdafne_amd.so.
#include "common.h"

namespace dafne {
char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace dafne

extern "C" {
int dafne_abi_version(void) { return 141; }  // 0.4.1: DAFNE_CONV_FRAG16 (dafne_conv3x3_c256_hip in the 16x16x32 fragment order);  // 0.4.0: dafne_conv3x3_c256_fp8w_hip removed, dafne_bottleneck_body_hip takes row-permuted weights (engine.pack_bneck);  // 0.3.1: dafne_stem_pool_conv1_hip;  // 0.3.0: dafne_conv2d_wr_*, dafne_bottleneck_block_{narrow,mid}_hip, bottleneck_body without a head, unknown conv flags rejected;  // 0.2.0: per-call NMS flags replace dafne_poly_nms_set_exact_only;  // 0.1.1: dafne_conv_params grew (GN_FINALIZE), b2b weight layout; .1: narrow tail+head
const char* dafne_last_error(void) { return dafne::err_buf(); }
}
