/* sh_dev.h -- what the HIP translation units of libscrappie_hip.so share (not part of the ABI): the error text behind
 * scrappie_hip_last_error(), HIPCHK, grow-only device and pinned host buffers. */
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <ctime>
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

/* Waiting for the device without burning a CPU.  On this runtime hipEventSynchronize / hipStreamSynchronize POLL (1.0 CPU for as long as the wait lasts,
 * whether or not the event was created with hipEventBlockingSync; only the process-wide hipSetDeviceFlags(hipDeviceScheduleBlockingSync) changes that, and a
 * library has no business setting it: tools/wait_probe.hip, profiles/r6_host_waits.txt) -- an engine kept busy cost 1.6 CPUs of watching, 13 of a
 * 16-CPU quota for an 8-rank job.  These ask (hipEventQuery / hipStreamQuery) and sleep ~70 us in between (a few per cent of a CPU), and what the sleep adds to a wait is
 * hidden behind the next launch group, which is enqueued before the oldest one is collected.  SCRAPPIE_HIP_SPIN_WAIT=1: the runtime's own waits. */
static inline bool sh_spin_wait() { static const bool on = getenv("SCRAPPIE_HIP_SPIN_WAIT") != nullptr; return on; }
static inline timespec sh_wait_nap() {       /* SCRAPPIE_HIP_WAIT_NAP_US (default 20; the kernel's timer slack adds ~50) */
    static const long ns = [] { const char *v = getenv("SCRAPPIE_HIP_WAIT_NAP_US"); const long us = v ? atol(v) : 20; return (us < 1 ? 1 : us > 10000 ? 10000 : us) * 1000L; }();
    return timespec{0, ns};
}
static inline hipError_t sh_wait_done(hipError_t r) { if (r == hipSuccess) (void)hipGetLastError(); return r; }     /* (a "not ready" answer is not an error to remember) */
static inline hipError_t sh_event_wait(hipEvent_t ev) {
    if (sh_spin_wait()) return hipEventSynchronize(ev);
    const timespec nap = sh_wait_nap();
    for (;;) { const hipError_t r = hipEventQuery(ev); if (r != hipErrorNotReady) return sh_wait_done(r); nanosleep(&nap, nullptr); }
}
static inline hipError_t sh_stream_wait(hipStream_t s) {
    if (sh_spin_wait()) return hipStreamSynchronize(s);
    const timespec nap = sh_wait_nap();
    for (;;) { const hipError_t r = hipStreamQuery(s); if (r != hipErrorNotReady) return sh_wait_done(r); nanosleep(&nap, nullptr); }
}

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
