/* oracle/oracle.h -- CPU restatement of the `scrappie raw` hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under scrappie_amd/ (the product) may
 * include, link or execute this.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * Every function is a plain scalar C restatement (no SSE, no BLAS) of the
 * reference function cited next to it (paths relative to /root/reference).
 *
 * PARITY PINNING STATUS (see DESIGN.md "Oracle"):
 *   - signal prep (P0), homopolymer (D2), stitching helpers, vector math
 *     (exp/log/logistic/tanh/elu): pinned bit-for-bit against the reference's
 *     own compiled code (oracle/_ref/libref_pure.so) and its golden .crp files.
 *   - decode (D1, D3, D4, D5): pinned bit-for-bit against the reference's
 *     decode.c compiled from where it lies, hosted on this file's allocator
 *     (oracle/_ref/libref_decode.so; scrappie_matrix.c itself needs cblas.h,
 *     which this image lacks).
 *   - network layers (C1, L1, G1, G2, S1, S2, K1, N1, N2): PARITY UNPINNED
 *     against compiled reference code -- layers.c / scrappie_matrix.c need an
 *     external BLAS header and networks.c needs model headers that are missing
 *     blobs; they are unbuildable here without stand-ins.  Pinned only by the
 *     reference's unit-test literals (ELU, row-normalise, 1-tap convolution)
 *     and by an independent float64 numpy cross-check.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* T1: src/scrappie_matrix.h:10-16.  ABI-identical to _Mat (the __m128* arm of
 * the union is a pointer, so void* keeps the layout). */
typedef struct {
    size_t nr, nrq, nc, stride;
    union { void *v; float *f; } data;
} orc_mat;

/* T2: src/scrappie_matrix.h:18-24 */
typedef struct {
    size_t nr, nrq, nc, stride;
    union { void *v; int32_t *f; } data;
} orc_imat;

/* T3: src/scrappie_structures.h:24-30 */
typedef struct {
    char *uuid;
    size_t n, start, end;
    float *raw;
} orc_raw_table;

enum { ORC_ARCH_RGRGR = 0, ORC_ARCH_RNNRF = 1, ORC_ARCH_RAW = 2, ORC_ARCH_EVENTS = 3 };
enum { ORC_ACT_ELU = 0, ORC_ACT_TANH = 1 };

/* W: weight set of one model (names: misc/parse_rgrgr.py:79-130). */
typedef struct {
    int arch;      /* ORC_ARCH_* */
    int conv_act;  /* ORC_ACT_*  : networks.c:260 (elu) / :358 (tanh) */
    int stride;    /* conv_<tag>_stride */
    const orc_mat *conv_W, *conv_b;
    const orc_mat *gru_iW[5], *gru_sW[5], *gru_sW2[5], *gru_b[5]; /* B1 F2 B3 F4 B5 */
    const orc_mat *ff_W, *ff_b;
    /* raw_r94 only (networks.c:196-247): gru_* slots 0..3 are F1, B1, F2, B2 */
    const orc_mat *ff1_Wf, *ff1_Wb, *ff1_b, *ff2_Wf, *ff2_Wb, *ff2_b;
    /* events bi-LSTM only (networks.c:146-193): gru_iW/gru_sW/gru_b slots 0..3 hold the LSTMs'
     * iW, sW, b (F1, B1, F2, B2) and lstm_p their peepholes; ff1_*, ff2_*, ff_* as for raw_r94 */
    const orc_mat *lstm_p[4];
} orc_model;

/* bench-only: route the two BLAS call shapes to a cblas implementation */
void orc_set_blas(void *cblas_sgemv, void *cblas_sgemm);

/* ---- T1/T2 containers : scrappie_matrix.c:11,44,69,80,130,269 ---- */
orc_mat *orc_make_mat(size_t nr, size_t nc);
orc_mat *orc_remake_mat(orc_mat *M, size_t nr, size_t nc);
orc_mat *orc_free_mat(orc_mat *M);
orc_mat *orc_mat_from_array(const float *x, size_t nr, size_t nc);
float *orc_array_from_mat(const orc_mat *M);
orc_imat *orc_make_imat(size_t nr, size_t nc);
orc_imat *orc_free_imat(orc_imat *M);

/* ---- A1 vector math : util.h:170-198, sse_mathfun.h:123-301 ---- */
float orc_expf(float x);
float orc_logf(float x);
float orc_logisticf(float x);
float orc_tanhf(float x);
float orc_eluf(float x);
float orc_logsumexpf(float x, float y);  /* util.h:162 */

/* ---- P0 signal prep : util.c:69-206, scrappie_common.c:5-73 ---- */
void orc_quantilef(const float *x, size_t nx, float *p, size_t np);
float orc_medianf(const float *x, size_t n);
float orc_madf(const float *x, size_t n, const float *med);
void orc_medmad_normalise_array(float *x, size_t n);
orc_raw_table orc_trim_raw_by_mad(orc_raw_table rt, size_t chunk_size, float perc);
/* NB: unlike the reference this never frees rt.raw; on failure returns a
 * zeroed table (raw == NULL) and the caller keeps ownership. */
orc_raw_table orc_trim_and_segment_raw(orc_raw_table rt, size_t trim_start,
                                       size_t trim_end, size_t varseg_chunk,
                                       float varseg_thresh);

/* ---- F0 : nnfeatures.c:102 ---- */
orc_mat *orc_features_from_raw(orc_raw_table signal);

/* ---- layers : layers.c ---- */
orc_mat *orc_convolution(const orc_mat *X, const orc_mat *W, const orc_mat *b,
                         size_t stride, orc_mat *C);            /* :159 */
void orc_tanh_activation_inplace(orc_mat *C);                   /* :15 */
void orc_exp_activation_inplace(orc_mat *C);                    /* :30 */
void orc_elu_activation_inplace(orc_mat *C);                    /* :60 */
void orc_robustlog_activation_inplace(orc_mat *C, float min_prob); /* :79 */
orc_mat *orc_affine_map(const orc_mat *X, const orc_mat *W, const orc_mat *b,
                        orc_mat *C);              /* scrappie_matrix.c:323 */
orc_mat *orc_affine_map2(const orc_mat *Xf, const orc_mat *Xb, const orc_mat *Wf,
                         const orc_mat *Wb, const orc_mat *b, orc_mat *C); /* scrappie_matrix.c:353 */
void orc_row_normalise_inplace(orc_mat *C);       /* scrappie_matrix.c:385 */
void orc_shift_scale_matrix_inplace(orc_mat *C, float shift, float scale); /* :560 */
void orc_residual_inplace(const orc_mat *X, orc_mat *fX);       /* :303 */
void orc_gru_step(const orc_mat *x, const orc_mat *istate, const orc_mat *sW,
                  const orc_mat *sW2, orc_mat *xF, orc_mat *ostate); /* :472 */
orc_mat *orc_gru_forward(const orc_mat *X, const orc_mat *sW, const orc_mat *sW2,
                         orc_mat *ostate);                      /* :373 */
orc_mat *orc_gru_backward(const orc_mat *X, const orc_mat *sW, const orc_mat *sW2,
                          orc_mat *ostate);                     /* :422 */
orc_mat *orc_softmax_with_temperature(orc_mat *X, const orc_mat *W, const orc_mat *b,
                                      float tempW, float tempb, orc_mat *C); /* :340 */
float orc_crf_partition_function(const orc_mat *C);             /* :835 */
orc_mat *orc_globalnorm(const orc_mat *X, const orc_mat *W, const orc_mat *b,
                        orc_mat *C);                            /* :874 */

/* ---- N1/N2 networks : networks.c:250-394, :567-615 ---- */
orc_mat *orc_rgrgr_posterior(const orc_model *m, orc_raw_table signal, float min_prob,
                             float tempW, float tempb, bool return_log);
orc_mat *orc_rnnrf_transitions(const orc_model *m, orc_raw_table signal);
/* N3: networks.c:196-247; `upto` as orc_trunk: 0 conv, 1 after FF1, 2 after FF2 */
orc_mat *orc_raw_trunk(const orc_model *m, orc_raw_table signal, int upto);
orc_mat *orc_raw_posterior(const orc_model *m, orc_raw_table signal, float min_prob,
                           float tempW, float tempb, bool return_log);
/* dispatch on m->arch (get_posterior_function, networks.c:108) */
orc_mat *orc_posterior(const orc_model *m, orc_raw_table signal, float min_prob,
                       float tempW, float tempb, bool return_log);
/* intermediate capture for layer-by-layer parity: returns activation after
 * layer `upto` (0 = conv+act, 1..5 = GRU layer output incl. residual). */
orc_mat *orc_trunk(const orc_model *m, orc_raw_table signal, int upto);

/* ---- decode : decode.c ---- */
int orc_argmaxf(const float *x, size_t n);                      /* util.c:9 */
float orc_decode_transducer(const orc_mat *logpost, float stay_pen, float skip_pen,
                            float local_pen, int *seq, bool allow_slip); /* :123 */
float orc_sloika_viterbi(const orc_mat *logpost, float stay_pen, float skip_pen,
                         float local_pen, int *seq);            /* :725 */
int orc_overlap(int k1, int k2, int nkmer);                     /* :367 */
char *orc_overlapper(const int *seq, size_t n, int nkmer, int *pos); /* :449 */
float orc_decode_crf(const orc_mat *trans, int *path);          /* :836 */
char *orc_crfpath_to_basecall(const int *path, size_t npos, int *pos); /* :895 */
orc_mat *orc_posterior_crf(const orc_mat *trans);               /* :928 */

/* ---- D2 : homopolymer.c:67-235, scrappie_seq_helpers.c:115-139 ---- */
int orc_repeatblock(int b, int nrep);
int orc_kmerlength_fromnblocks(int n);
int orc_homopolymer_path(const orc_mat *post, int *viterbipath, int mean_flag);

/* ---- whole read, as scrappie_raw.c:265-315 (signal already read) ---- */
typedef struct {
    float score;
    size_t nblock;
    size_t start, end;   /* trim window */
    char *basecall;      /* malloc'd, caller frees */
    int *pos;            /* malloc'd nblock+1, caller frees */
} orc_call;
typedef struct {
    float min_prob, tempW, tempb, stay_pen, skip_pen, local_pen;
    int use_slip, homopolymer_mean;
    int trim_start, trim_end, varseg_chunk;
    float varseg_thresh;
    int do_trim;       /* 0: use [start,end) as given, signal already normalised */
} orc_params;
orc_params orc_default_params(void);               /* scrappie_raw.c:98-121 */
/* raw is copied; returns 0 on success, nonzero if no basecall */
int orc_basecall_raw(const orc_model *m, const float *raw, size_t n,
                     const orc_params *p, orc_call *out);
/* scrappie_raw.c:355,387: threads over reads (OpenMP, dynamic), for bench.py's cpu_baseline: reads are
 * walked round-robin until budget_s seconds have passed (at least min_reads); returns reads done */
long orc_basecall_many(const orc_model *m, const float *const *raws, const size_t *ns, size_t nreads,
                       const orc_params *p, int nthreads, double budget_s, size_t min_reads,
                       double *samples, double *bases, double *elapsed);

#ifdef __cplusplus
}
#endif

/* ---- (f).4 events bi-LSTM : nnfeatures.c:51-110, layers.c:119-147, :673-832, networks.c:146-193 ---- */
typedef struct { uint64_t start; float length, mean, stdv; int pos, state; } orc_event;   /* scrappie_structures.h:8-15 */
typedef struct { size_t n, start, end; orc_event *event; } orc_event_table;
void orc_studentise_features_kahan(orc_mat *features);
orc_mat *orc_features_from_events(orc_event_table et, bool normalise);
orc_mat *orc_window(const orc_mat *input, size_t w, size_t stride);
void orc_lstm_step(const orc_mat *xAffine, const orc_mat *out_prev, const orc_mat *sW, const orc_mat *peep,
                   orc_mat *xF, orc_mat *state, orc_mat *output);
orc_mat *orc_lstm_forward(const orc_mat *Xaffine, const orc_mat *sW, const orc_mat *p, orc_mat *output);
orc_mat *orc_lstm_backward(const orc_mat *Xaffine, const orc_mat *sW, const orc_mat *p, orc_mat *output);
/* feature3 = window(features_from_events(et, true), 3, 1): [12 x nevent] */
orc_mat *orc_events_trunk(const orc_model *m, const orc_mat *feature3, int upto);
orc_mat *orc_events_posterior_from_features(const orc_model *m, const orc_mat *feature3, float min_prob,
                                            float tempW, float tempb, bool return_log);
orc_mat *orc_events_posterior(const orc_model *m, orc_event_table et, float min_prob, float tempW, float tempb,
                              bool return_log);

#endif
