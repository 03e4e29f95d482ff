/* include/scrappie_hip.h -- C ABI of libscrappie_hip.so, the MI355X-native
 * implementation of the `scrappie raw` basecalling hot path.
 *
 * Two surfaces:
 *
 *  (1) The reference's per-read surface, same names, signatures, struct
 *      layouts, ownership and NULL/NAN-on-error behaviour, so that a binding
 *      written against the reference (python/pyscrap.h, interface/scrappie.h,
 *      src/decode.h, src/networks.h) loads this library unchanged.  Each
 *      declaration cites the reference prototype it replaces.
 *
 *  (2) An additive batched engine surface (the reference has none: it calls
 *      one read at a time, src/scrappie_raw.c:265).  This is the fast path:
 *      reads are coalesced into launch groups that fill the GPU, decoded on
 *      device, and only paths/bases return over PCIe.
 *
 * Plain C types only: no torch, no HIP types.  Device pointers cross the
 * boundary as `void *` / `const float *` and are documented as such.
 */
#ifndef SCRAPPIE_HIP_H
#define SCRAPPIE_HIP_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------
 * Types shared with the reference (layouts are ABI)
 * ---------------------------------------------------------------------- */

/* src/scrappie_structures.h:24-30 == interface/scrappie.h:29-35 */
typedef struct {
    char *uuid;
    size_t n;
    size_t start;
    size_t end;
    float *raw;
} raw_table;

/* src/scrappie_structures.h:8-22 (events basecaller input) */
typedef struct {
    uint64_t start;
    float length;
    float mean;
    float stdv;
    int pos;
    int state;
} event_t;

typedef struct {
    size_t n;
    size_t start;
    size_t end;
    event_t *event;
} event_table;

/* src/scrappie_matrix.h:10-16 == interface/scrappie.h:38-45.  Column-major,
 * rows padded to nrq = ceil(nr/4) 4-float vectors, stride = 4*nrq, 16-byte
 * aligned.  The `v` arm is `__m128 *` in the reference; any object pointer
 * keeps the layout. */
typedef struct {
    size_t nr, nrq, nc, stride;
    union {
        void *v;
        float *f;
    } data;
} _Mat;
typedef _Mat *scrappie_matrix;
typedef _Mat const *const_scrappie_matrix;

/* src/networks.h:8-14 */
enum raw_model_type {
    SCRAPPIE_MODEL_RAW = 0,
    SCRAPPIE_MODEL_RGRGR_R9_4,
    SCRAPPIE_MODEL_RGRGR_R9_4_1,
    SCRAPPIE_MODEL_RGRGR_R10,
    SCRAPPIE_MODEL_RNNRF_R9_4,
    SCRAPPIE_MODEL_INVALID
};

/* src/homopolymer.h */
enum homopolymer_calculation {
    HOMOPOLYMER_NOCHANGE = 0,
    HOMOPOLYMER_MEAN,
    HOMOPOLYMER_INVALID
};

/* src/networks.h:22 */
typedef scrappie_matrix (*posterior_function_ptr)(const raw_table, float, float, float, bool);

/* ------------------------------------------------------------------------
 * (1) Per-read surface -- drop-in for the reference symbols
 * ---------------------------------------------------------------------- */

/* src/networks.c:17, :49, :87, :108 */
enum raw_model_type get_raw_model(const char *modelstr);
const char *raw_model_string(const enum raw_model_type model);
int get_raw_model_stride(const enum raw_model_type model);
posterior_function_ptr get_posterior_function(const enum raw_model_type model);
/* python/build.py:34-44 (defined only in the cffi build of the reference) */
int get_raw_model_stride_from_string(const char *modelstr);

/* src/networks.c:250, :299, :348, :567 (python/pyscrap.h:11-23).  Run on the
 * process-default engine (device 0 or $SCRAPPIE_HIP_DEVICE) as a launch group
 * of one read; the returned matrix is HOST memory in the reference's padded
 * layout, released with free_scrappie_matrix.  NULL on any failure.  Weights
 * come from the model registered under the same name (scrappie_hip_register_
 * model / $SCRAPPIE_MODEL_DIR/<name>.scrm); the reference compiles them in. */
/* src/networks.c:196 (interface/scrappie.h:49): bi-GRU model raw_r94 */
scrappie_matrix nanonet_raw_posterior(const raw_table signal, float min_prob,
                                      float tempW, float tempb, bool return_log);
scrappie_matrix nanonet_rgrgr_r94_posterior(const raw_table signal, float min_prob,
                                            float tempW, float tempb, bool return_log);
scrappie_matrix nanonet_rgrgr_r941_posterior(const raw_table signal, float min_prob,
                                             float tempW, float tempb, bool return_log);
scrappie_matrix nanonet_rgrgr_r10_posterior(const raw_table signal, float min_prob,
                                            float tempW, float tempb, bool return_log);
scrappie_matrix nanonet_rnnrf_r94_transitions(const raw_table signal, float min_prob,
                                              float tempW, float tempb, bool return_log);
/* src/networks.c:146: events bi-LSTM; weights from the model registered as
 * "nanonet_events".  Features (nnfeatures.c:88, Kahan studentisation with rsqrtps) and the
 * 3-event window (layers.c:119, first column zero as in the reference) are host C. */
scrappie_matrix nanonet_posterior(const event_table events, float min_prob,
                                  float tempW, float tempb, bool return_log);

/* src/decode.c:123 -- seq has nblock+1 entries.  Returns NAN on failure. */
float decode_transducer(const_scrappie_matrix logpost, float stay_pen, float skip_pen,
                        float local_pen, int *seq, bool allow_slip);
/* src/decode.c:449 -- calloc'd string, caller frees; NULL if every entry is a stay */
char *overlapper(const int *seq, size_t n, int nkmer, int *pos);
/* src/decode.c:836, :895, :928 */
float decode_crf(const_scrappie_matrix trans, int *path);
char *crfpath_to_basecall(int const *path, size_t npos, int *pos);
scrappie_matrix posterior_crf(const_scrappie_matrix trans);
/* src/homopolymer.c:175 */
int homopolymer_path(const_scrappie_matrix post, int *viterbipath,
                     enum homopolymer_calculation pathCalculationFlag);

/* src/util.c:190, src/scrappie_common.c:5, :39 */
void medmad_normalise_array(float *x, size_t n);
raw_table trim_and_segment_raw(raw_table rt, size_t trim_start, size_t trim_end,
                               size_t varseg_chunk, float varseg_thresh);
raw_table trim_raw_by_mad(raw_table rt, size_t chunk_size, float perc);

/* src/scrappie_matrix.c:11, :69, :130 */
scrappie_matrix make_scrappie_matrix(size_t nr, size_t nc);
scrappie_matrix mat_from_array(const float *x, size_t nr, size_t nc);
scrappie_matrix free_scrappie_matrix(scrappie_matrix mat);

/* ------------------------------------------------------------------------
 * (2) Batched engine surface (additive)
 * ---------------------------------------------------------------------- */

typedef struct scrappie_hip_engine scrappie_hip_engine;

/* Decode / posterior parameters: defaults are the CLI's, src/scrappie_raw.c:98-121 */
typedef struct {
    float min_prob;      /* -m   1e-5 */
    float tempW;         /* --temperature1 1.0 */
    float tempb;         /* --temperature2 1.0 */
    float stay_pen;      /* -y   0.0 */
    float skip_pen;      /* -s   0.0 */
    float local_pen;     /* --local 2.0 */
    int use_slip;        /* --slip  0 */
    int homopolymer;     /* enum homopolymer_calculation: HOMOPOLYMER_MEAN */
    int want_pos;        /* fill scrappie_hip_call.pos */
} scrappie_hip_params;

/* One basecall, as struct _raw_basecall_info (src/scrappie_raw.c:25-34) */
typedef struct {
    float score;         /* Viterbi score; NAN if the read produced no call */
    size_t nblock;
    char *basecall;      /* malloc'd, NUL terminated; NULL if no call */
    size_t basecall_length;
    int *pos;            /* malloc'd nblock+1 if want_pos, else NULL */
} scrappie_hip_call;

/* Per-stage device time of the last launch group, milliseconds, measured with
 * HIP events on the engine's own stream (only filled when profiling is on). */
typedef struct {
    float conv_ms, affine_ms, gru_ms, ff_ms, decode_ms, backtrace_ms, total_ms;
    int n_gru_launches, n_affine_launches;
    double gru_flops, affine_flops, ff_flops;   /* algorithmic FLOPs of those launches */
    /* of which: recurrent layers run as one kernel, projection + recurrence (their time is part of
     * gru_ms, their FLOPs = projection + recurrence, counted in affine_flops / gru_flops) */
    float fused_ms;
    int n_fused_launches;
    double fused_flops;
    float stitch_ms;     /* the copy-stream kernel behind the decoder (under the next group): k_walk_stitch_out -- traceback walk + homopolymer correction +
                          * k-mer stitching + results to pinned host memory -- and backtrace_ms is 0; with SH_SPLIT_TAIL=1, host stitching or the flip-flop
                          * models: k_stitch alone, the walk in backtrace_ms */
} scrappie_hip_timing;

int scrappie_hip_device_count(void);
/* NUMA node of the socket the device hangs off (sysfs; -1 when unknown or the host has one node).  Pinned staging and result buffers of the
 * device's engine / preparer are allocated with the calling thread bound to that node's CPUs for the duration of the allocation, so that
 * loader threads write and the GPU's DMA reads socket-local memory on a two-socket node (SCRAPPIE_HIP_NUMA=0: off). */
int scrappie_hip_device_numa_node(int device);
/* last error message of the calling thread ("" if none) */
const char *scrappie_hip_last_error(void);

scrappie_hip_engine *scrappie_hip_engine_create(int device);
void scrappie_hip_engine_destroy(scrappie_hip_engine *e);
scrappie_hip_params scrappie_hip_default_params(void);

/* Load a `.scrm` weight container (scrappie_amd/model.py) and bind it to a
 * reference model name ("raw_r94", "rgrgr_r94", "rgrgr_r941", "rgrgr_r10", "rnnrf_r94").
 * Returns a model handle >= 0, or -1. */
int scrappie_hip_load_model(scrappie_hip_engine *e, const char *name, const char *path);
/* Same from memory: `blob` is the container image. */
int scrappie_hip_load_model_mem(scrappie_hip_engine *e, const char *name,
                                const void *blob, size_t nbytes);
int scrappie_hip_find_model(scrappie_hip_engine *e, const char *name);
/* bind a model to the process-default engine used by the per-read surface */
int scrappie_hip_register_model(const char *name, const char *path);
/* The per-read network functions above may be called from many host threads at once, as the reference's OpenMP loop over reads does
 * (scrappie_raw.c:355,387): calls that arrive while the device is busy, or while the leader waits for company -- windows of SCRAPPIE_HIP_COALESCE_US (default 500)
 * microseconds until three quarters of the threads seen lately have joined or a window passes with no arrival, SCRAPPIE_HIP_COALESCE_MAX_US
 * (default 10000) in all (scrappie_amd/csrc/sh_coalesce.h) -- run as ONE launch group and every caller gets its own matrix -- bit for bit the one it would get alone.  SCRAPPIE_HIP_COALESCE=0: one
 * read per launch, as in earlier rounds.  out[0..2] = launch groups run this way, reads in them, the largest group (tests, tools). */
void scrappie_hip_coalescer_stats(unsigned long long out[3]);
/* decode_transducer is coalesced the same way (one workgroup per waiting call, each the single-read form: same path, same score) */
void scrappie_hip_decode_coalescer_stats(unsigned long long out[3]);
/* ... and decode_crf (a thread per waiting call) */
void scrappie_hip_crf_coalescer_stats(unsigned long long out[3]);

/* Basecall n reads.  Each raw_table's raw[start..end) must already be trimmed
 * and normalised (as calculate_post does before the network,
 * src/scrappie_raw.c:273-277).  Reads shorter than the model's minimum
 * (scrappie_hip_min_samples) yield no call.  out[i] corresponds to reads[i].
 * Returns 0, or -1 with scrappie_hip_last_error().
 * May be called from several host threads on one engine (round 6): calls of fewer than 4096 reads that are waiting for the same
 * engine, model and parameters run as ONE engine call and share its launch groups -- the reference's `schedule(dynamic)` loop
 * (src/scrappie_raw.c:355,387) with a body that hands over 64 reads at a time keeps the device full that way; every caller gets exactly the
 * calls it would get alone.  Larger calls go in one at a time.  SCRAPPIE_HIP_COALESCE=0 or SCRAPPIE_HIP_BATCH_COALESCE=0: no sharing. */
int scrappie_hip_basecall_batch(scrappie_hip_engine *e, int model,
                                const raw_table *reads, size_t n,
                                const scrappie_hip_params *p, scrappie_hip_call *out);
/* (tests, tools) out[0..2] = engine calls made for shared small calls, caller calls in them, the most callers in one */
void scrappie_hip_batch_coalescer_stats(unsigned long long out[3]);

/* The same on SEVERAL engines (one per GPU of a node; engines[k] holds the model as models[k]).  The
 * reference's parallel axis is reads, `#pragma omp parallel for schedule(dynamic)` (src/scrappie_raw.c:355,387);
 * here reads are sorted by length, cut into launch groups, and each engine's host thread takes the next
 * group from an atomic cursor whenever it has room (two groups in flight per engine).  out[i] <-> reads[i].
 * No data moves between GPUs.  The engines must not be used by other threads meanwhile. */
int scrappie_hip_basecall_batch_multi(scrappie_hip_engine *const *engines, const int *models, size_t nengine,
                                      const raw_table *reads, size_t n, const scrappie_hip_params *p,
                                      scrappie_hip_call *out);
/* The hand-out plan of the call above (host only, no device): order[] takes the read indices sorted by
 * length, longest first; starts[] the first position (in that order) of each launch group; returns the
 * number of groups (even if > cap), -1 if one read alone exceeds max_blocks. */
long scrappie_hip_plan_dynamic(const uint32_t *lengths, size_t n, int stride, size_t nengine, size_t max_reads,
                               size_t max_blocks, uint32_t *order, size_t *starts, size_t cap);
/* Chain-bound reads (host only).  A read is a serial chain of blocks (recurrent layers of alternating direction, decoder, traceback),
 * so a launch group lasts at least as long as its longest read whatever else the device does.  scrappie_hip_basecall_batch
 * therefore runs the reads whose own chain exceeds what the whole call would take with the device full on a helper
 * engine of the same device, beside the launch groups of the others (the GPU analogue of a long read occupying one thread of
 * the reference's schedule(dynamic) loop, src/scrappie_raw.c:355-400).  is_long[n] gets 0 / 1: longest first, at most
 * max_long_blocks column blocks (0: no limit) and 15 % of the call's blocks; returns their number (0: the call is not split).
 * SCRAPPIE_HIP_TAIL=0 in the environment turns the split off. */
long scrappie_hip_plan_tail(const uint32_t *lengths, size_t n, int stride, size_t max_long_blocks, unsigned char *is_long);
/* scrappie_hip_basecall_batch that does not wait for its chain-bound reads.  Their calls are collected later, so the launch groups
 * of the NEXT calls run beside them as well, and the helper engine takes the long reads of all the calls waiting for it as ONE
 * launch group (which lasts as long as its longest read however many it holds): a stream of calls with long-tailed read lengths
 * runs at the device's rate instead of one longest-read chain per call.  deferred[n] gets 1 for the reads whose out[] entry is
 * still blank.  Returns a ticket (> 0) if any read was deferred, 0 if none, -1 on error.  The deferred reads' signals must stay
 * valid until their ticket has been collected.  One host thread per engine.  The helper engine is created by the first call that
 * has chain-bound reads and stays: from then on the engine's launch groups may take 45 % of the device's memory instead of 70 %
 * (the helper 30 % -- SCRAPPIE_HIP_TAIL_MEM, (0 .. 0.4] -- or two helpers, SCRAPPIE_HIP_TAIL=2, 15 % each), i.e. later calls are cut
 * into slightly smaller launch groups. */
long scrappie_hip_basecall_batch_deferred(scrappie_hip_engine *e, int model, const raw_table *reads, size_t n,
                                          const scrappie_hip_params *p, scrappie_hip_call *out, unsigned char *deferred);
/* ... for signals already on the device (scrappie_hip_prep_run): the chain-bound reads are copied back into host memory the ticket
 * owns, so d_signal may be reused as soon as the call returns; tickets are collected as above */
long scrappie_hip_basecall_device_deferred(scrappie_hip_engine *e, int model, const float *d_signal, const uint64_t *offsets,
                                           const uint32_t *lengths, size_t n, const scrappie_hip_params *p, scrappie_hip_call *out,
                                           unsigned char *deferred);
/* The calls of a ticket's deferred reads, in the order those reads had in their call.  wait = 0: -2 if they are not ready yet.
 * Returns their number; -1 on error (unknown ticket, cap too small, or the helper's launch group failed: the ticket is gone). */
long scrappie_hip_deferred_collect(scrappie_hip_engine *e, long ticket, scrappie_hip_call *out, size_t cap, int wait);

/* Device-resident variant (bench / pipelines that already hold signal in HBM):
 * d_signal is a DEVICE pointer to concatenated normalised samples; read i is
 * d_signal[offsets[i] .. offsets[i]+lengths[i]).  offsets/lengths are host. */
int scrappie_hip_basecall_device(scrappie_hip_engine *e, int model,
                                 const float *d_signal, const uint64_t *offsets,
                                 const uint32_t *lengths, size_t n,
                                 const scrappie_hip_params *p, scrappie_hip_call *out);

/* Signal preparation on the device (k_p0, sh_p0.h): what calculate_post does to a read before the network sees it
 * (src/scrappie_raw.c:270-277) -- trim_and_segment_raw (src/scrappie_common.c:5-73) and medmad_normalise_array
 * (src/util.c:190-205) -- for a whole batch of reads in one launch, bit-identical to the host functions above (windows
 * and samples).  A preparer belongs to one GPU and owns a stream and three slots of buffers (0, 1, 2), so that batch k+1
 * can be prepared (by another host thread) while the engine of the same GPU still reads batch k's signals -- and, with the streaming
 * call below, the last launch group of batch k-1.
 *   scrappie_hip_prep_run: reads[i].raw[0 .. n) are RAW samples (as scrappie_hip_read_raw returns them), reads[i].start /
 *   .end the window at entry.  The samples are gathered into pinned memory, copied to the device, prepared there;
 *   on return *d_signal is the slot's DEVICE buffer, read i's prepared window is d_signal[offsets[i] .. + lengths[i]),
 *   start[i] / end[i] are the window within the read as trim_and_segment_raw would have left rt.start / rt.end;
 *   lengths[i] = 0 where it returns an empty raw_table (nothing left of the read).  offsets / lengths feed
 *   scrappie_hip_basecall_device as they are.  The slot's buffer stays valid until the slot is used again.
 *   varseg_chunk = 0 (a division by zero in the reference) skips trim_raw_by_mad: only the fixed trims apply.
 *   Returns 0, -1 on error (scrappie_hip_last_error). */
typedef struct scrappie_hip_prep scrappie_hip_prep;
scrappie_hip_prep *scrappie_hip_prep_create(int device);
void scrappie_hip_prep_destroy(scrappie_hip_prep *p);
int scrappie_hip_prep_run(scrappie_hip_prep *p, int slot, const raw_table *reads, size_t n,
                          size_t trim_start, size_t trim_end, size_t varseg_chunk, float varseg_thresh,
                          const float **d_signal, uint64_t *offsets, uint32_t *lengths,
                          uint32_t *start, uint32_t *end);
/* The slot's pinned staging buffer handed out piece by piece (thread safe: an atomic cursor): scrappie_hip_prep_begin(slot,
 * capacity in samples) resets the cursor and makes room; scrappie_hip_prep_alloc (a scrappie_hip_sample_alloc; ctx = what
 * _begin returned) gives nsample floats of it, or NULL once the capacity is used up (the reader then mallocs).
 * scrappie_hip_prep_run recognises reads whose .raw lies in the slot's staging buffer and copies only the others.
 * scrappie_hip_prep_owns: whether a pointer lies in the slot's staging buffer (such samples are not the caller's to free). */
void *scrappie_hip_prep_begin(scrappie_hip_prep *p, int slot, size_t capacity_samples);
/* pinned and device buffers of a slot for batches of up to capacity_samples, ahead of time (only while nothing reads the slot;
 * scrappie_hip_prep_begin itself never touches the device side: the engine may still be reading the slot's previous batch) */
int scrappie_hip_prep_reserve(scrappie_hip_prep *p, int slot, size_t capacity_samples);
float *scrappie_hip_prep_alloc(void *ctx, size_t nsample);
int scrappie_hip_prep_owns(scrappie_hip_prep *p, int slot, const float *ptr);
/* copy count prepared samples of the slot, from sample `offset` on, to the host (tests; the CLI never needs them) */
int scrappie_hip_prep_fetch(scrappie_hip_prep *p, int slot, uint64_t offset, size_t count, float *dst);
/* milliseconds the last scrappie_hip_prep_run of the slot spent in (gather on the host, host-to-device copy, k_p0) */
void scrappie_hip_prep_timing(scrappie_hip_prep *p, int slot, double out[3]);

/* Streaming form of scrappie_hip_basecall_device: returns while the call's LAST launch group is still running; that group's calls
 * are delivered -- into the same out[] -- by the next scrappie_hip_basecall_device_stream call on the engine (behind that call's first
 * launch: the engine's two-deep pipeline never drains between batches) or by scrappie_hip_stream_flush; until then those out[] entries
 * are blank (basecall == NULL, score NAN).  out[] and d_signal of a call must stay valid until then.  Any other batched call on the
 * engine delivers the carried group first.  If a call fails, the carried group's calls are lost too (entries stay blank). */
int scrappie_hip_basecall_device_stream(scrappie_hip_engine *e, int model, const float *d_signal, const uint64_t *offsets,
                                        const uint32_t *lengths, size_t n, const scrappie_hip_params *p, scrappie_hip_call *out);
int scrappie_hip_stream_flush(scrappie_hip_engine *e);
int scrappie_hip_stream_pending(scrappie_hip_engine *e);      /* 1: the last streaming call's last launch group has not been delivered yet */
/* scrappie_hip_basecall_device_deferred whose reads that are not deferred are streamed as above, whether or not the call defers
 * any: out[] entries of the last launch group stay blank until the next streaming call or the flush delivers them; the deferred
 * reads' entries until scrappie_hip_deferred_collect */
long scrappie_hip_basecall_device_deferred_stream(scrappie_hip_engine *e, int model, const float *d_signal, const uint64_t *offsets,
                                                  const uint32_t *lengths, size_t n, const scrappie_hip_params *p, scrappie_hip_call *out,
                                                  unsigned char *deferred);

/* Make the engine's arenas (device and pinned) and code objects ready for launch groups of n reads of `samples` samples: runs one
 * such group on all-zero signals and discards it.  Optional; without it the first calls grow the arenas as they go. */
int scrappie_hip_warm_up(scrappie_hip_engine *e, int model, size_t n, size_t samples);

/* Lower-level, asynchronous: enqueue the device part for reads already in HBM (metadata upload,
 * kernels, D2H of the paths into pinned buffers) and return; scrappie_hip_collect waits for that
 * launch group and stitches it on the host.  The engine holds TWO launch groups, so group k+1 can be
 * enqueued before group k is collected (its kernels then hide the stitching of group k); collect
 * always takes the OLDEST group in flight and its n must match; a third enqueue without a collect is
 * refused.  d_signal must stay valid until the group has been collected.  Returns total blocks, < 0
 * on error.  One thread drives an engine at a time (the per-read functions of the reference surface
 * take the default engine's lock themselves). */
long scrappie_hip_run_device(scrappie_hip_engine *e, int model, const float *d_signal,
                             const uint64_t *offsets, const uint32_t *lengths, size_t n,
                             const scrappie_hip_params *p);
/* wait for the oldest launch group in flight and stitch it on the host (homopolymer correction,
 * k-mer overlap); out[i] describes read i of that group */
int scrappie_hip_collect(scrappie_hip_engine *e, const scrappie_hip_params *p,
                         scrappie_hip_call *out, size_t n);

void scrappie_hip_free_calls(scrappie_hip_call *calls, size_t n);

/* Measurement / test hook for SURVEY.md 8(d) "decode driven by HMM-simulated posteriors" (synthetic
 * weights decode to a handful of bases per read, which leaves the decode -> D2H -> homopolymer ->
 * overlapper stage almost idle).  From the next launch group on, everything downstream of S1 -- the
 * device Viterbi, the homopolymer rows, scrappie_hip_posterior -- sees, instead of the network's own
 * normalised posterior, probabilities supplied by the caller: read i of a launch group takes the
 * row-major [nblock_i][nstate] float matrix at d_prob + prob_off[i % n_prob] (DEVICE memory, nstate =
 * the model's; rows past the read's nblock are not read).  The network itself still runs in full (S1
 * writes its own buffer).  The decoder image of the matrices is built on the first launch group and
 * re-used while consecutive groups have the same shape.  d_prob = NULL switches the hook off.
 * Transducer models only; no launch group may be in flight.  The reference has no counterpart. */
int scrappie_hip_set_decoder_input(scrappie_hip_engine *e, const float *d_prob, const uint64_t *prob_off, size_t n_prob);

/* The same idea one stage earlier, so that the DEFAULT decode path (S1 inside the decoder, k_ff_viterbi) sees
 * realistic posteriors: from the next launch group on, the output layer (S1 / globalnorm) and everything
 * downstream read, instead of the trunk's own output, activations supplied by the caller: read i of a launch
 * group takes the row-major [nblock_i][S] float matrix at d_trunk + trunk_off[i % n_trunk] (DEVICE memory, S = the
 * model's state width; blocks past the matrix's read are zero).  The network above the output layer still runs
 * in full.  With an output layer built from state codes (scrappie_amd.synth.hmm_output_layer) the posteriors
 * are those of a simulated k-mer path, computed by the production kernels.  d_trunk = NULL switches the hook
 * off; no launch group may be in flight.  The reference has no counterpart. */
int scrappie_hip_set_trunk_input(scrappie_hip_engine *e, const float *d_trunk, const uint64_t *trunk_off, size_t n_trunk);

/* Test hooks (tests/test_gpu_parity.py: the whole traceback of the two decoder forms, byte for byte).
 * scrappie_hip_debug_option: "ff_separate" = S1 and the decoder as two kernels on this engine (k_ff_lds / k_ff_exp +
 * k_viterbi) even where the one-kernel forms apply; "fv_single" = S1 inside the decoder on k_ff_viterbi's eight do-everything waves
 * instead of k_ff_viterbi_teams (S1 team + decoder team; env SH_FV_SINGLE=1 does the same for a process); "dump_final" = the decoders leave every tile's final scores in the
 * hand-over buffer; "fail_run" = k: the k-th next launch group is refused (failure paths); "redo_all" = every read
 * takes the host fallback of k_stitch; "gru_tiles" = 1 / 2: tiles of 16 reads per workgroup of the recurrent layers
 * whatever the lane schedules say (0: the engine chooses).  scrappie_hip_debug_fetch copies a buffer of the most recent transducer launch group to the
 * host: "tb" (one byte per state: [column block][state quad][read of tile][state of quad]), "tb_end" (int per
 * column block and read), "final_state" / "final_score" (per read, tiled order), "final_scores" ([tile][states x
 * 16 reads | 16 start | 16 end] floats, needs dump_final), "order" (int per tiled position: index of the read in
 * the call, -1 = padding), "tile_boff" (long long per tile: its first column block), "n_redo" (unsigned long long:
 * reads whose stitching k_stitch has left to the host since the engine was created), "gru_tiles" (int: what the group's
 * recurrent layers ran with).  Returns the bytes the
 * buffer holds (min(that, nbytes) are copied; dst may be NULL), -1 on error. */
int scrappie_hip_debug_option(scrappie_hip_engine *e, const char *name, int value);
/* k_stitch (homopolymer correction + k-mer stitching / crfpath_to_basecall on the device: what the batched path
 * runs instead of src/homopolymer.c:175 + src/decode.c:449 / :895 on the host) on ONE read given on the host, for
 * the bit-exact tests against the compiled-reference fixtures.  path: nblock + 1 entries; side: [nblock][5] log-
 * posterior rows (homopolymer k-mers of A, C, G, T, then stay) or NULL (no homopolymer pass); nstate: 4^k + 1 (25
 * with crf != 0).  bases takes at most cap bytes incl. the NUL; pos (nblock + 1 ints) and redo (1: the device
 * left the posterior-mean rounding of a run to the host) may be NULL.  Returns the number of bases, -1 = no
 * call (all stays), -2 = error. */
long scrappie_hip_debug_stitch(scrappie_hip_engine *e, const int *path, const float *side, size_t nblock, int nstate, int crf,
                               char *bases, size_t cap, int *pos, int *redo);
long long scrappie_hip_debug_fetch(scrappie_hip_engine *e, const char *what, void *dst, size_t nbytes);

/* Posterior of one read on a given engine/model (what the per-read surface
 * calls): HOST matrix in reference layout. */
scrappie_matrix scrappie_hip_posterior(scrappie_hip_engine *e, int model, const raw_table signal,
                                       float min_prob, float tempW, float tempb, bool return_log);
/* Intermediate activations for layer-by-layer parity tests: layer 0 = conv +
 * activation, 1..5 = GRU layer outputs (incl. residual for rnnrf). */
scrappie_matrix scrappie_hip_trunk(scrappie_hip_engine *e, int model, const raw_table signal, int upto);

/* Events models (arch "events"): the input of a read is its windowed feature matrix,
 * row-major [nevent][12] floats.  For these models the raw_table / offsets / lengths of the
 * calls above count FLOATS of that matrix for `raw`, `start`, `end` and offsets, and EVENTS
 * for `lengths` of scrappie_hip_run_device.
 * scrappie_hip_event_features: events[start..end) -> out[(end-start) * 12]
 *   = window(nanonet_features_from_events(events, true), 3, 1)   (networks.c:155-157). */
int scrappie_hip_event_features(const event_table events, float *out);
scrappie_matrix scrappie_hip_events_posterior(scrappie_hip_engine *e, int model, const float *feature3, size_t nevent,
                                              float min_prob, float tempW, float tempb, bool return_log);

/* Lane schedule of the recurrent kernel (scrappie_amd/csrc/sh_sched.h), host only:
 * how the tiles (16 reads, tile_T[i] blocks) of a launch group are cut into
 * segments for the 2 * nwg lanes.  lane_off: 2 * ncu + 1 ints; seg: cap rows of
 * {tile, first step, end step, 0}.  Returns the number of segments. */
long scrappie_hip_gru_schedule(const int *tile_T, size_t ntile, int ncu, int *nwg, int *capacity,
                               int *lane_off, int *seg, size_t cap);
/* ... with lanes_per_wg = 1 (k_gru_proj: projection + recurrence teams, one lane per workgroup) or 2;
 * lane_off: lanes_per_wg * ncu + 1 ints */
long scrappie_hip_lane_schedule(const int *tile_T, size_t ntile, int ncu, int lanes_per_wg, int *nwg, int *capacity,
                                int *lane_off, int *seg, size_t cap);

/* Pieces of the Viterbi decoder's launch (scrappie_amd/csrc/sh_sched.h), host only:
 * seg takes cap rows of {tile, first block, end block, 0} in workgroup order. */
long scrappie_hip_decoder_pieces(const int *tile_T, size_t ntile, int ncu, int *seg, size_t cap);

size_t scrappie_hip_min_samples(scrappie_hip_engine *e, int model);
int scrappie_hip_model_stride(scrappie_hip_engine *e, int model);
void scrappie_hip_set_profiling(scrappie_hip_engine *e, int on);
int scrappie_hip_get_timing(scrappie_hip_engine *e, scrappie_hip_timing *t);
/* upper bound on reads per launch group (default 16384) */
void scrappie_hip_set_max_launch_reads(scrappie_hip_engine *e, size_t n);
/* upper bound on column blocks (16 reads x 1 block) per launch group; 0 (default) = derived from the
 * device's memory (about 70 % of it across the arena).  scrappie_hip_basecall_batch/_device cut their
 * input into launch groups under both bounds and keep two groups in flight. */
void scrappie_hip_set_max_launch_blocks(scrappie_hip_engine *e, size_t n);
/* the cut itself (host only, no device): starts[] takes the first read of each group, input order
 * kept; returns the number of groups (even if > cap), -1 if one read alone exceeds max_blocks */
long scrappie_hip_plan_groups(const uint32_t *lengths, size_t n, int stride, size_t max_reads, size_t max_blocks,
                              size_t *starts, size_t cap);
/* Host threads an engine of this process uses for gathering and stitching: min(CPUs of the affinity mask, the cgroup's
 * cpu.max quota, 32), divided by LOCAL_WORLD_SIZE when a launcher runs one process per GPU (torchrun sets it), or
 * SCRAPPIE_HIP_HOST_THREADS.  Read once per process. */
unsigned scrappie_hip_host_thread_budget(void);
/* ... and without the per-engine cap of 32: the CPUs the process may use (affinity, cgroup quota, / LOCAL_WORLD_SIZE) -- what `scrappie raw` sizes its
 * loader team from: 16 loader threads per GPU (fast5 input needs ~14 to feed one engine, profiles/r6_cli_rate.txt), as many as there are CPUs at most */
unsigned scrappie_hip_host_cpu_budget(void);
/* device memory helpers so a host with no HIP runtime of its own can stage data */
void *scrappie_hip_device_alloc(scrappie_hip_engine *e, size_t nbytes);
void scrappie_hip_device_free(scrappie_hip_engine *e, void *dptr);
int scrappie_hip_memcpy_h2d(scrappie_hip_engine *e, void *dst, const void *src, size_t nbytes);
int scrappie_hip_synchronize(scrappie_hip_engine *e);

/* read_raw (src/fast5_interface.c:130): first read of a fast5 file, optionally
 * scaled to pA; also reads headerless *.f32 / *.i16 signal files.  HDF5 is
 * resolved with dlopen at run time (scrappie_hip_have_hdf5() tells whether one
 * was found).  Caller frees .raw and .uuid; .raw == NULL on failure. */
raw_table scrappie_hip_read_raw(const char *filename, bool scale_to_pA);
/* ... with the samples placed where `alloc` says (NULL, or a NULL return: malloc as above) -- e.g. scrappie_hip_prep_alloc, so
 * that a loader thread reads a file straight into the pinned buffer the device copies from */
typedef float *(*scrappie_hip_sample_alloc)(void *ctx, size_t nsample);
raw_table scrappie_hip_read_raw_into(const char *filename, bool scale_to_pA, scrappie_hip_sample_alloc alloc, void *ctx);
int scrappie_hip_have_hdf5(void);
/* offset, range, digitisation attributes of a fast5 file; 0 on success */
int scrappie_hip_fast5_scaling(const char *filename, float out[3]);

/* FASTA / SAM record exactly as src/scrappie_raw.c:317-331 prints them.
 * Returns the number of characters written (snprintf semantics). */
int scrappie_hip_format_fasta(char *buf, size_t buflen, const char *uuid, const char *readname,
                              bool uuid_primary, const char *prefix, const scrappie_hip_call *res,
                              size_t nsample, size_t trim_start, size_t trim_end);
int scrappie_hip_format_sam(char *buf, size_t buflen, const char *uuid, const char *readname,
                            bool uuid_primary, const char *prefix, const scrappie_hip_call *res);

#ifdef __cplusplus
}
#endif
#endif /* SCRAPPIE_HIP_H */
