/* sh_dev.h -- what the HIP translation units of libscrappie_hip.so share (not part of the ABI): the error text behind
 * scrappie_hip_last_error(), HIPCHK, grow-only device and pinned host buffers. */
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include "sh_numa.h"

int sh_set_err_v(const char *fmt, va_list ap);      /* scrappie_hip.hip: thread-local text; returns -1 */
static inline int set_err(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
static inline int set_err(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    const int rc = sh_set_err_v(fmt, ap);
    va_end(ap);
    return rc;
}

#define HIPCHK(call)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            set_err("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return -1;                                                                      \
        }                                                                                   \
    } while (0)

struct DBuf {   /* device buffer, grow-only */
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        const size_t want = bytes + bytes / 8 + 4096;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            return set_err("hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
        }
        cap = want;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T *as() const { return (T *)p; }
};

struct HBuf {   /* pinned host buffer, grow-only */
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        const size_t want = bytes + bytes / 8 + 4096;
        int dev = 0;
        (void)hipGetDevice(&dev);
        ShNumaScope near_gpu(dev);          /* pages on the socket the current device hangs off (sh_numa.h) */
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e != hipSuccess) { p = nullptr; return set_err("hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e)); }
        cap = want;
        return 0;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
    template <typename T> T *as() const { return (T *)p; }
};
