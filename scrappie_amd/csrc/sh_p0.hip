/* sh_p0.hip -- scrappie_hip_prep_*: signal preparation of a batch of reads on the device (k_p0, sh_p0.h).
 *
 * The reference prepares a read on the host thread that basecalls it (scrappie_raw.c:270-277: read_raw ->
 * trim_and_segment_raw -> medmad_normalise_array).  Here the engine consumes 1.5e9 samples/s per GPU and a host
 * thread prepares ~3e7, so the preparation is a kernel over the batch; the host only moves bytes (file -> pinned
 * memory -> device).  The host functions (sh_host.c) remain what the per-read surface uses, and what the tests
 * compare k_p0 with, bit for bit.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>
#include <vector>

#include "scrappie_hip.h"
#include "sh_dev.h"
#include "sh_p0.h"

#define SH_PREP_SLOTS 3      /* a streaming consumer still reads batch k - 1's buffer while batch k runs and batch k + 1 is prepared */
struct scrappie_hip_prep {
    int device = 0;
    hipStream_t stream = nullptr;
    struct Slot {
        HBuf h_sig, h_meta, h_win;
        DBuf d_sig, d_scratch, d_meta, d_win;
        size_t total = 0;
        double ms[3] = {0, 0, 0};
        /* staging handed out to loader threads (scrappie_hip_prep_begin / _alloc): [0, cap_samples) of h_sig */
        std::atomic<size_t> cursor{0};
        size_t cap_samples = 0;
    } slot[SH_PREP_SLOTS];
    hipEvent_t ev[3];
    bool ev_ok = false;
};

extern "C" scrappie_hip_prep *scrappie_hip_prep_create(int device) {
    if (hipSetDevice(device) != hipSuccess) { set_err("no GPU %d", device); return nullptr; }
    scrappie_hip_prep *p = new scrappie_hip_prep;
    p->device = device;
    if (hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) != hipSuccess) { set_err("hipStreamCreate failed"); delete p; return nullptr; }
    p->ev_ok = true;
    for (auto &e : p->ev) if (hipEventCreate(&e) != hipSuccess) p->ev_ok = false;
    return p;
}

extern "C" void scrappie_hip_prep_destroy(scrappie_hip_prep *p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    (void)sh_stream_wait(p->stream);
    for (auto &s : p->slot) {
        s.h_sig.release(); s.h_meta.release(); s.h_win.release();
        s.d_sig.release(); s.d_scratch.release(); s.d_meta.release(); s.d_win.release();
    }
    if (p->ev_ok) for (auto &e : p->ev) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(p->stream);
    delete p;
}

extern "C" void *scrappie_hip_prep_begin(scrappie_hip_prep *p, int slot, size_t capacity_samples) {
    if (!p || slot < 0 || slot >= SH_PREP_SLOTS) { set_err("prep_begin: bad argument"); return nullptr; }
    if (hipSetDevice(p->device) != hipSuccess) { set_err("prep_begin: no GPU %d", p->device); return nullptr; }
    auto &S = p->slot[slot];
    /* the pinned side only: the slot's DEVICE buffer may still be read by the engine (a loader fills the staging of batch k + 3 while batch k
     * is on the GPU); scrappie_hip_prep_run, which the caller issues when the slot is free, grows the device side */
    if (S.h_sig.ensure(std::max<size_t>(capacity_samples, 1) * 4)) return nullptr;
    S.cap_samples = capacity_samples;
    S.cursor.store(0);
    return &S;
}

/* pinned AND device buffers of a slot for batches of up to capacity_samples, made ahead of time (a slot that has to grow in the middle of a run
 * frees and allocates device memory: both synchronise the device).  Only while nothing reads the slot. */
extern "C" int scrappie_hip_prep_reserve(scrappie_hip_prep *p, int slot, size_t capacity_samples) {
    if (!p || slot < 0 || slot >= SH_PREP_SLOTS) return set_err("prep_reserve: bad argument");
    HIPCHK(hipSetDevice(p->device));
    auto &S = p->slot[slot];
    const size_t b = std::max<size_t>(capacity_samples, 1) * 4;
    return (S.h_sig.ensure(b) || S.d_sig.ensure(b) || S.d_scratch.ensure(b)) ? -1 : 0;
}

extern "C" float *scrappie_hip_prep_alloc(void *ctx, size_t nsample) {
    auto *S = (scrappie_hip_prep::Slot *)ctx;
    if (!S || !nsample) return nullptr;
    const size_t need = (nsample + 3) & ~(size_t)3;
    const size_t at = S->cursor.fetch_add(need);
    if (at + need > S->cap_samples) return nullptr;          /* (the cursor stays past the end: later requests fail too, or fit if smaller -- either is fine) */
    return S->h_sig.as<float>() + at;
}

extern "C" int scrappie_hip_prep_owns(scrappie_hip_prep *p, int slot, const float *ptr) {
    if (!p || slot < 0 || slot >= SH_PREP_SLOTS || !ptr) return 0;
    const auto &S = p->slot[slot];
    const float *b = S.h_sig.as<float>();
    return b && ptr >= b && ptr < b + S.cap_samples;
}

static unsigned prep_threads() {
    const char *e = getenv("SCRAPPIE_HIP_PREP_THREADS");
    unsigned n = e ? (unsigned)atoi(e) : std::min(std::max(std::thread::hardware_concurrency(), 1u), 8u);
    return std::max(n, 1u);
}

extern "C" int scrappie_hip_prep_run(scrappie_hip_prep *p, int slot, const raw_table *reads, size_t n,
                                     size_t trim_start, size_t trim_end, size_t varseg_chunk, float varseg_thresh,
                                     const float **d_signal, uint64_t *offsets, uint32_t *lengths,
                                     uint32_t *start, uint32_t *end) {
    if (!p || (!reads && n) || !d_signal || !offsets || !lengths) return set_err("prep_run: null argument");
    if (slot < 0 || slot >= SH_PREP_SLOTS) return set_err("prep_run: slot must be 0, 1 or 2");
    if (n > 0x7fffffffu) return set_err("prep_run: too many reads");
    if (trim_start > 0xffffffffu || trim_end > 0xffffffffu || varseg_chunk > 0xffffffffu) return set_err("prep_run: trim parameter out of range");
    HIPCHK(hipSetDevice(p->device));
    auto &S = p->slot[slot];
    /* layout: a read already in the slot's staging buffer (scrappie_hip_prep_alloc) stays where it is; the others are laid
     * behind what has been handed out, each rounded up to 4 samples (16-byte rows for the copies) */
    std::vector<uint64_t> off(n);
    std::vector<unsigned char> inplace(n, 0);
    const float *hs0 = S.h_sig.as<float>();
    const size_t used = std::min(S.cursor.load(), S.cap_samples);
    size_t total = hs0 ? used : 0;
    for (size_t i = 0; i < n; i++) {
        const raw_table &rt = reads[i];
        if (rt.raw && rt.n > 0xffffffffull) return set_err("prep_run: read %zu has more than 2^32 samples", i);
        if (rt.raw && hs0 && rt.raw >= hs0 && rt.raw + rt.n <= hs0 + used) { inplace[i] = 1; off[i] = (uint64_t)(rt.raw - hs0); continue; }
        off[i] = total;
        total += rt.raw ? ((rt.n + 3) & ~(size_t)3) : 0;
    }
    const size_t meta_words = 2 * n /* off */ + 3 * n /* len, st0, en0 */;
    /* A call that fails leaves the staging buffer -- where the caller's reads lie -- as it was, so that the caller can still prepare the
     * batch elsewhere (scrappie raw: on the host).  So the device side grows first (where running out of memory is likely), and the
     * staging buffer, if reads have to be added behind what has been handed out, moves by allocate / copy / free, not free / allocate. */
    if (S.d_sig.ensure(std::max<size_t>(total, 1) * 4) || S.d_scratch.ensure(std::max<size_t>(total, 1) * 4) ||
        S.h_meta.ensure(std::max<size_t>(meta_words, 1) * 4) || S.d_meta.ensure(std::max<size_t>(meta_words, 1) * 4) ||
        S.h_win.ensure(std::max<size_t>(2 * n, 1) * 4) || S.d_win.ensure(std::max<size_t>(2 * n, 1) * 4))
        return -1;
    if (std::max<size_t>(total, 1) * 4 > S.h_sig.cap) {
        HBuf nb;
        if (nb.ensure(std::max<size_t>(total, 1) * 4)) return -1;
        if (used && hs0) memcpy(nb.p, hs0, used * 4);
        S.h_sig.release();
        S.h_sig = nb;
    }
    S.total = total;
    *d_signal = S.d_sig.as<float>();
    if (n == 0) return 0;
    const auto t0 = std::chrono::steady_clock::now();
    float *hs = S.h_sig.as<float>();
    unsigned long long *h_off = S.h_meta.as<unsigned long long>();
    unsigned *h_len = (unsigned *)(h_off + n), *h_st = h_len + n, *h_en = h_st + n;
    for (size_t i = 0; i < n; i++) {
        const raw_table &rt = reads[i];
        h_off[i] = off[i];
        h_len[i] = rt.raw ? (unsigned)rt.n : 0u;
        h_st[i] = rt.raw ? (unsigned)std::min<size_t>(rt.start, rt.n) : 0u;
        h_en[i] = rt.raw ? (unsigned)std::min<size_t>(rt.end, rt.n) : 0u;
    }
    {   /* gather on several host threads (a memcpy per read; 16 384 x 4000 samples = 262 MB) */
        const unsigned nthr = (total * 4 > ((size_t)8 << 20)) ? prep_threads() : 1u;
        auto part = [&](size_t a, size_t b) {
            for (size_t i = a; i < b; i++)
                if (reads[i].raw && reads[i].n && !inplace[i]) memcpy(hs + off[i], reads[i].raw, reads[i].n * 4);
        };
        if (nthr == 1) part(0, n);
        else {
            std::vector<std::thread> th;
            const size_t step = (n + nthr - 1) / nthr;
            for (unsigned t = 0; t < nthr; t++) { const size_t a = t * step, b = std::min(n, a + step); if (a < b) th.emplace_back(part, a, b); }
            for (auto &x : th) x.join();
        }
    }
    const auto t1 = std::chrono::steady_clock::now();
    if (p->ev_ok) HIPCHK(hipEventRecord(p->ev[0], p->stream));
    HIPCHK(hipMemcpyAsync(S.d_sig.p, hs, total * 4, hipMemcpyHostToDevice, p->stream));
    HIPCHK(hipMemcpyAsync(S.d_meta.p, S.h_meta.p, meta_words * 4, hipMemcpyHostToDevice, p->stream));
    if (p->ev_ok) HIPCHK(hipEventRecord(p->ev[1], p->stream));
    ShP0Args A;
    A.x = S.d_sig.as<float>();
    A.off = S.d_meta.as<unsigned long long>();
    A.len = (const unsigned *)(A.off + n);
    A.st0 = A.len + n;
    A.en0 = A.st0 + n;
    A.win = S.d_win.as<unsigned>();
    A.scratch = S.d_scratch.as<float>();
    A.trim_start = (unsigned)trim_start; A.trim_end = (unsigned)trim_end; A.chunk = (unsigned)varseg_chunk;
    A.perc = varseg_thresh;
    A.nread = (unsigned)n;
    const unsigned grid = (unsigned)std::min<size_t>(n, 256 * 16);
    hipLaunchKernelGGL(k_p0, dim3(grid), dim3(SH_P0_THREADS), 0, p->stream, A);
    HIPCHK(hipGetLastError());
    if (p->ev_ok) HIPCHK(hipEventRecord(p->ev[2], p->stream));
    HIPCHK(hipMemcpyAsync(S.h_win.p, S.d_win.p, 2 * n * 4, hipMemcpyDeviceToHost, p->stream));
    HIPCHK(sh_stream_wait(p->stream));
    const unsigned *w = S.h_win.as<unsigned>();
    for (size_t i = 0; i < n; i++) {
        const unsigned s = w[2 * i], e = w[2 * i + 1];
        const bool live = reads[i].raw && e > s;
        offsets[i] = off[i] + (live ? s : 0);
        lengths[i] = live ? e - s : 0;
        if (start) start[i] = live ? s : 0;
        if (end) end[i] = live ? e : 0;
    }
    S.ms[0] = std::chrono::duration<double, std::milli>(t1 - t0).count();
    if (p->ev_ok) {
        float a = 0, b = 0;
        (void)hipEventElapsedTime(&a, p->ev[0], p->ev[1]);
        (void)hipEventElapsedTime(&b, p->ev[1], p->ev[2]);
        S.ms[1] = a; S.ms[2] = b;
    }
    return 0;
}

extern "C" int scrappie_hip_prep_fetch(scrappie_hip_prep *p, int slot, uint64_t offset, size_t count, float *dst) {
    if (!p || !dst || slot < 0 || slot >= SH_PREP_SLOTS) return set_err("prep_fetch: bad argument");
    auto &S = p->slot[slot];
    if (offset + count > S.total) return set_err("prep_fetch: range outside the slot's %zu samples", S.total);
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpy(dst, S.d_sig.as<float>() + offset, count * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" void scrappie_hip_prep_timing(scrappie_hip_prep *p, int slot, double out[3]) {
    for (int i = 0; i < 3; i++) out[i] = (p && slot >= 0 && slot < SH_PREP_SLOTS) ? p->slot[slot].ms[i] : 0.0;
}
