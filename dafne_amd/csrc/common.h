// Shared host-side helpers for libdafne_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/dafne_amd.h"

namespace dafne {

char* err_buf();  // thread-local, 512 bytes (abi.hip)

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(DAFNE_E_HIP, "%s: %s", what, hipGetErrorString(e));
    return DAFNE_OK;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct WsCarver {
    char* base;
    size_t off = 0;
    explicit WsCarver(void* p) : base(static_cast<char*>(p)) {}
    template <typename T>
    T* take(size_t n) {
        off = align_up(off, 256);
        T* r = reinterpret_cast<T*>(base + off);
        off += n * sizeof(T);
        return r;
    }
};

}  // namespace dafne

#define DAFNE_HIP_TRY(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) return dafne::fail(DAFNE_E_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)
