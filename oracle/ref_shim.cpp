// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE.  extern "C" doorway into the
// reference's own iou_poly (tools/prepare_dota/polyiou.cpp:112, declared in
// polyiou.h:9), compiled from where it lies under /root/reference by
// oracle/Makefile into oracle/_ref/libpolyiou_ref.so.  No reference source is
// copied: this file only declares the symbol and forwards to it.
#include <vector>
double iou_poly(std::vector<double> p, std::vector<double> q);

extern "C" double ref_iou_poly(const double* p8, const double* q8) {
    return iou_poly(std::vector<double>(p8, p8 + 8), std::vector<double>(q8, q8 + 8));
}

extern "C" void ref_iou_poly_pairs(const double* p, const double* q, long n, double* out) {
    for (long i = 0; i < n; i++) out[i] = ref_iou_poly(p + 8 * i, q + 8 * i);
}
