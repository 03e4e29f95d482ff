/* cli_pipe_stub.c -- a stand-in for the GPU side of libscrappie_hip.so, so that `scrappie raw` (scrappie_amd/csrc/scrappie_raw.c: loader /
 * engine / writer threads over a ring of batches, three preparer slots, streaming engine calls, deferred tickets, shares per GPU) can run
 * on a CPU under -fsanitize=thread (tests/test_host_cpu.py::test_cli_pipeline_under_thread_sanitizer).  TEST INFRASTRUCTURE: nothing the
 * product links.  The host C of the product (sh_host.c, sh_fast5.c, sh_h5mini.c) is linked as it is.
 *
 * What the stubs do, so that a broken hand-over between the stages shows up in the OUTPUT, not only in the sanitizer's report:
 *   prep_run          window of read i = [200, n - 10) (empty below 211 samples); the "device" buffer of the slot is host memory
 *   basecall_device*  a read's call = the CRC-32 of its window's sample bytes, read from the slot's buffer WHEN THE CALL IS DELIVERED
 *   ..._stream        keeps the last third of a batch back and delivers it in the next call or in stream_flush -- from the same buffer,
 *                     which the loader must therefore not have reused yet
 *   deferred          reads of more than 6000 samples get a ticket; their samples are copied when the ticket is made
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "scrappie_hip.h"

struct scrappie_hip_engine { int device; /* streaming */ int live; const float *d; uint64_t *off; uint32_t *len; size_t n0, n1; scrappie_hip_call *out;
                             /* tickets */ pthread_mutex_t mu; long next; struct tk { long id; size_t n; scrappie_hip_call *calls; struct tk *nx; } *open; };
struct slot { float *buf; size_t cap; atomic_size_t cur; size_t total; float *dev; size_t devcap; };      /* buf: pinned staging; dev: the "device" buffer */
struct scrappie_hip_prep { int device; struct slot s[3]; };

static _Thread_local char err[256] = "stub";
const char *scrappie_hip_last_error(void) { return err; }
int scrappie_hip_device_count(void) { return 4; }
scrappie_hip_engine *scrappie_hip_engine_create(int device) { scrappie_hip_engine *e = calloc(1, sizeof *e); e->device = device; e->next = 1; pthread_mutex_init(&e->mu, NULL); return e; }
void scrappie_hip_engine_destroy(scrappie_hip_engine *e) { free(e->off); free(e->len); free(e); }
scrappie_hip_params scrappie_hip_default_params(void) { scrappie_hip_params p; memset(&p, 0, sizeof p); p.local_pen = 2.0f; return p; }
int scrappie_hip_load_model(scrappie_hip_engine *e, const char *name, const char *path) { (void)e; (void)name; (void)path; return 0; }
int scrappie_hip_warm_up(scrappie_hip_engine *e, int model, size_t n, size_t samples) { (void)e; (void)model; (void)n; (void)samples; return 0; }
void scrappie_hip_free_calls(scrappie_hip_call *c, size_t n) { for (size_t i = 0; i < n; i++) { free(c[i].basecall); free(c[i].pos); c[i].basecall = NULL; c[i].pos = NULL; } }

static uint32_t crc32(const void *p, size_t n) {
    uint32_t c = 0xffffffffu;
    const unsigned char *b = p;
    for (size_t i = 0; i < n; i++) { c ^= b[i]; for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xedb88320u & (0u - (c & 1u))); }
    return ~c;
}
static void make_call(scrappie_hip_call *c, const float *x, size_t n) {
    memset(c, 0, sizeof *c);
    c->score = NAN;
    if (!n) return;
    char *s = calloc(16, 1);
    static const char B[4] = {'A', 'C', 'G', 'T'};
    uint32_t h = crc32(x, n * 4);
    for (int k = 0; k < 12; k++) { s[k] = B[h & 3]; h = (h >> 2) | (h << 30); }       /* 24 bits of the checksum as 12 bases */
    c->basecall = s; c->basecall_length = 12; c->nblock = n / 5; c->score = (float)n;
    usleep(3);                                                                          /* (a launch takes a while) */
}

/* ---- preparer ---- */
scrappie_hip_prep *scrappie_hip_prep_create(int device) { scrappie_hip_prep *p = calloc(1, sizeof *p); p->device = device; return p; }
void scrappie_hip_prep_destroy(scrappie_hip_prep *p) { for (int k = 0; k < 3; k++) { free(p->s[k].buf); free(p->s[k].dev); } free(p); }
/* failure injection (round 6): STUB_RESERVE_FAIL_ABOVE=<samples> -- a reservation of more than that fails, as a device out of memory would make it;
 * STUB_PREP_FAIL_AT=<k> -- the k-th scrappie_hip_prep_run of the process fails (1-based) */
static size_t env_size(const char *name) { const char *v = getenv(name); return v ? (size_t)atof(v) : 0; }
int scrappie_hip_prep_reserve(scrappie_hip_prep *p, int slot, size_t cap) {
    (void)p; (void)slot;
    const size_t lim = env_size("STUB_RESERVE_FAIL_ABOVE");
    if (lim && cap > lim) { snprintf(err, sizeof err, "stub: out of memory reserving %zu samples", cap); return -1; }
    return 0;
}
static atomic_size_t n_prep_runs;
static int grow(struct slot *S, size_t cap) { if (cap > S->cap) { float *nb = realloc(S->buf, cap * 4); if (!nb) return -1; S->buf = nb; S->cap = cap; } return 0; }
void *scrappie_hip_prep_begin(scrappie_hip_prep *p, int slot, size_t cap) { struct slot *S = &p->s[slot]; if (grow(S, cap + 4)) return NULL; atomic_store(&S->cur, 0); S->total = cap; return S; }
float *scrappie_hip_prep_alloc(void *ctx, size_t n) { struct slot *S = ctx; if (!S || !n) return NULL; const size_t need = (n + 3) & ~(size_t)3, at = atomic_fetch_add(&S->cur, need); return at + need <= S->total ? S->buf + at : NULL; }
int scrappie_hip_prep_owns(scrappie_hip_prep *p, int slot, const float *ptr) { const struct slot *S = &p->s[slot]; return S->buf && ptr >= S->buf && ptr < S->buf + S->total; }
void scrappie_hip_prep_timing(scrappie_hip_prep *p, int slot, double out[3]) { (void)p; (void)slot; out[0] = out[1] = out[2] = 0; }
int scrappie_hip_prep_run(scrappie_hip_prep *p, int slot, const raw_table *reads, size_t n, size_t ts, size_t te, size_t vc, float vt,
                          const float **d_signal, uint64_t *offsets, uint32_t *lengths, uint32_t *start, uint32_t *end) {
    (void)ts; (void)te; (void)vc; (void)vt;
    if (atomic_fetch_add(&n_prep_runs, 1) + 1 == env_size("STUB_PREP_FAIL_AT")) { snprintf(err, sizeof err, "stub: the preparer's device buffer did not grow"); return -1; }
    struct slot *S = &p->s[slot];
    size_t used = atomic_load(&S->cur), total = used < S->total ? used : S->total, extra = 0;
    for (size_t i = 0; i < n; i++) if (reads[i].raw && !(S->buf && reads[i].raw >= S->buf && reads[i].raw < S->buf + total)) extra += (reads[i].n + 3) & ~(size_t)3;
    const float *old = S->buf;
    if (grow(S, total + extra + 4)) return -1;
    for (size_t i = 0; i < n; i++) {
        const raw_table *rt = &reads[i];
        size_t at;
        if (!rt->raw) { offsets[i] = 0; lengths[i] = 0; if (start) start[i] = 0; if (end) end[i] = 0; continue; }
        if (old && rt->raw >= old && rt->raw < old + total) at = (size_t)(rt->raw - old);       /* already in the staging buffer (realloc kept the bytes) */
        else { at = total; memcpy(S->buf + at, rt->raw, rt->n * 4); total += (rt->n + 3) & ~(size_t)3; }
        const int live = rt->n > 210;
        offsets[i] = at + (live ? 200 : 0); lengths[i] = live ? (uint32_t)(rt->n - 210) : 0;
        if (start) start[i] = live ? 200 : 0;
        if (end) end[i] = live ? (uint32_t)(rt->n - 10) : 0;
    }
    /* "host to device": the device buffer of the slot is overwritten HERE -- the caller must know that nobody reads it any more */
    if (total > S->devcap) { free(S->dev); S->dev = malloc(total * 4 + 16); S->devcap = total; }
    memcpy(S->dev, S->buf, total * 4);
    *d_signal = S->dev;
    return 0;
}

/* ---- engine ---- */
int scrappie_hip_basecall_device(scrappie_hip_engine *e, int model, const float *d, const uint64_t *off, const uint32_t *len, size_t n,
                                 const scrappie_hip_params *p, scrappie_hip_call *out) {
    (void)model; (void)p;
    if (scrappie_hip_stream_flush(e)) return -1;
    for (size_t i = 0; i < n; i++) make_call(&out[i], d + off[i], len[i]);
    return 0;
}
int scrappie_hip_stream_pending(scrappie_hip_engine *e) { return e->live; }
int scrappie_hip_stream_flush(scrappie_hip_engine *e) {
    if (!e->live) return 0;
    for (size_t i = e->n0; i < e->n1; i++) make_call(&e->out[i], e->d + e->off[i], e->len[i]);       /* reads the slot's buffer NOW */
    e->live = 0;
    return 0;
}
long scrappie_hip_basecall_device_deferred_stream(scrappie_hip_engine *e, int model, const float *d, const uint64_t *off, const uint32_t *len, size_t n,
                                                  const scrappie_hip_params *p, scrappie_hip_call *out, unsigned char *deferred) {
    (void)model; (void)p;
    size_t nd = 0;
    for (size_t i = 0; i < n; i++) { deferred[i] = len[i] > 6000; nd += deferred[i]; }
    const size_t keep = nd ? n : n - n / 3;                /* a call that defers reads delivers everything else; otherwise the last third stays in flight */
    for (size_t i = 0; i < keep; i++) { if (deferred[i]) { memset(&out[i], 0, sizeof out[i]); out[i].score = NAN; } else make_call(&out[i], d + off[i], len[i]); }
    for (size_t i = keep; i < n; i++) { memset(&out[i], 0, sizeof out[i]); out[i].score = NAN; }
    if (scrappie_hip_stream_flush(e)) return -1;           /* the previous call's last third, behind this call's first launches */
    if (keep < n) {
        e->off = realloc(e->off, n * sizeof *e->off); e->len = realloc(e->len, n * sizeof *e->len);
        memcpy(e->off, off, n * sizeof *off); memcpy(e->len, len, n * sizeof *len);
        e->d = d; e->n0 = keep; e->n1 = n; e->out = out; e->live = 1;
    }
    if (!nd) return 0;
    struct tk *t = calloc(1, sizeof *t);
    t->n = nd; t->calls = calloc(nd, sizeof *t->calls);
    for (size_t i = 0, k = 0; i < n; i++) if (deferred[i]) make_call(&t->calls[k++], d + off[i], len[i]);
    pthread_mutex_lock(&e->mu); t->id = e->next++; t->nx = e->open; e->open = t; pthread_mutex_unlock(&e->mu);
    return t->id;
}
long scrappie_hip_deferred_collect(scrappie_hip_engine *e, long ticket, scrappie_hip_call *out, size_t cap, int wait) {
    (void)wait;
    pthread_mutex_lock(&e->mu);
    struct tk **pp = &e->open, *t = NULL;
    for (; *pp; pp = &(*pp)->nx) if ((*pp)->id == ticket) { t = *pp; *pp = t->nx; break; }
    pthread_mutex_unlock(&e->mu);
    if (!t || t->n > cap) return -1;
    memcpy(out, t->calls, t->n * sizeof *out);
    const long n = (long)t->n;
    free(t->calls); free(t);
    return n;
}
long scrappie_hip_basecall_batch_deferred(scrappie_hip_engine *e, int model, const raw_table *reads, size_t n, const scrappie_hip_params *p,
                                          scrappie_hip_call *out, unsigned char *deferred) {
    (void)e; (void)model; (void)p;
    memset(deferred, 0, n);
    for (size_t i = 0; i < n; i++) make_call(&out[i], reads[i].raw ? reads[i].raw + reads[i].start : NULL, reads[i].raw && reads[i].end > reads[i].start ? reads[i].end - reads[i].start : 0);
    return 0;
}
int scrappie_hip_basecall_batch(scrappie_hip_engine *e, int model, const raw_table *reads, size_t n, const scrappie_hip_params *p, scrappie_hip_call *out) {
    unsigned char *d = calloc(n ? n : 1, 1);
    const long rc = scrappie_hip_basecall_batch_deferred(e, model, reads, n, p, out, d);
    free(d);
    return rc < 0 ? -1 : 0;
}
int scrappie_hip_basecall_batch_multi(scrappie_hip_engine *const *es, const int *models, size_t ne, const raw_table *reads, size_t n,
                                      const scrappie_hip_params *p, scrappie_hip_call *out) {
    (void)ne; unsigned char *d = calloc(n ? n : 1, 1);
    const long rc = scrappie_hip_basecall_batch_deferred(es[0], models[0], reads, n, p, out, d);
    free(d);
    return rc < 0 ? -1 : 0;
}
unsigned scrappie_hip_host_thread_budget(void) { return 4; }
unsigned scrappie_hip_host_cpu_budget(void) { return 4; }
