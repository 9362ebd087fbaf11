// Shared host-side helpers for libdafne_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <initializer_list>
#include <mutex>
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

// One-time initialisation PER DEVICE.  hipFuncSetAttribute(MaxDynamicSharedMemorySize) and the CU count are properties of
// the current device, not of the process: a process that launches on a second GPU (shims/poly_nms.poly_gpu_nms(dets, thr,
// device_id) takes any device id) must set them again there.  Thread-safe: lock-free once the current device's bit is set.
struct DeviceOnce {
    std::atomic<unsigned long long> done{0};   // bit d: initialised on device d
    std::mutex mu;
};

template <class F>
inline int device_once(DeviceOnce& o, F&& init) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail(DAFNE_E_HIP, "hipGetDevice: %s", hipGetErrorString(e));
    if (dev < 0 || dev >= 64) return fail(DAFNE_E_HIP, "device id %d out of range", dev);
    const unsigned long long bit = 1ull << dev;
    if (o.done.load(std::memory_order_acquire) & bit) return DAFNE_OK;
    std::lock_guard<std::mutex> g(o.mu);
    if (o.done.load(std::memory_order_relaxed) & bit) return DAFNE_OK;
    if (int rc = init()) return rc;
    o.done.fetch_or(bit, std::memory_order_release);
    return DAFNE_OK;
}

inline int set_max_lds(std::initializer_list<const void*> kernels, int bytes) {
    for (const void* k : kernels) {
        hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return fail(DAFNE_E_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize, %d): %s", bytes, hipGetErrorString(e));
    }
    return DAFNE_OK;
}

// CU count of the CURRENT device (cached per device).
inline int device_cus(int* out) {
    static std::atomic<int> cus[64];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail(DAFNE_E_HIP, "hipGetDevice: %s", hipGetErrorString(e));
    if (dev < 0 || dev >= 64) return fail(DAFNE_E_HIP, "device id %d out of range", dev);
    int c = cus[dev].load(std::memory_order_relaxed);
    if (!c) {
        e = hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return fail(DAFNE_E_HIP, "hipDeviceGetAttribute: %s", hipGetErrorString(e));
        if (c <= 0) c = 256;
        cus[dev].store(c, std::memory_order_relaxed);
    }
    *out = c;
    return DAFNE_OK;
}

}  // namespace dafne

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per device for the listed kernels
#define DAFNE_MAX_LDS_ONCE(bytes, ...)                                                                              \
    do {                                                                                                            \
        static dafne::DeviceOnce _once;                                                                             \
        if (int _rc = dafne::device_once(_once, [&]() -> int { return dafne::set_max_lds({__VA_ARGS__}, (bytes)); })) return _rc; \
    } while (0)

#define DAFNE_HIP_TRY(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) return dafne::fail(DAFNE_E_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)
