/* sh_host.c -- host-side C of libscrappie_hip.so.
 *
 * The reference keeps its whole path in C on the host; here only the parts that
 * are O(read length) integer/byte work or one-off signal preparation stay on the
 * host, written in C and
 * exported with the reference's own names so existing bindings keep working.
 * Everything numeric per block runs in the HIP kernels (scrappie_hip.hip).
 *
 * Citations: file:line under /root/reference/src.
 */
#define _POSIX_C_SOURCE 200809L
#include "scrappie_hip.h"
#include "sh_internal.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* matrix container (scrappie_matrix.c:11, :69, :130)                  */
/* ------------------------------------------------------------------ */
scrappie_matrix make_scrappie_matrix(size_t nr, size_t nc) {
    if (nr == 0 || nc == 0) return NULL;
    const size_t nrq = (nr + 3) / 4;
    if (nc != 0 && (nrq * 16) > ((size_t)-1) / nc) return NULL;   /* overflow */
    scrappie_matrix m = malloc(sizeof(*m));
    if (!m) return NULL;
    void *buf = NULL;
    if (posix_memalign(&buf, 16, nrq * 16 * nc) != 0) {
        free(m);
        return NULL;
    }
    memset(buf, 0, nrq * 16 * nc);
    m->nr = nr; m->nrq = nrq; m->nc = nc; m->stride = 4 * nrq;
    m->data.v = buf;
    return m;
}

scrappie_matrix mat_from_array(const float *x, size_t nr, size_t nc) {
    if (!x) return NULL;
    scrappie_matrix m = make_scrappie_matrix(nr, nc);
    if (!m) return NULL;
    for (size_t c = 0; c < nc; c++)
        memcpy(m->data.f + c * m->stride, x + c * nr, nr * sizeof(float));
    return m;
}

scrappie_matrix free_scrappie_matrix(scrappie_matrix mat) {
    if (mat) {
        free(mat->data.v);
        free(mat);
    }
    return NULL;
}

/* ------------------------------------------------------------------ */
/* P0: order statistics by selection instead of the reference's qsort  */
/* (util.c:92-130).  The two order statistics a quantile needs are the */
/* same values whichever way they are found, so results are identical. */
/* ------------------------------------------------------------------ */
static inline void swapf(float *a, float *b) { float t = *a; *a = *b; *b = t; }

/* after return v[k] is the k-th smallest and v[k+1..n) >= v[k] */
static void select_kth(float *v, size_t n, size_t k) {
    size_t lo = 0, hi = n - 1;
    while (lo < hi) {
        const size_t mid = lo + (hi - lo) / 2;
        if (v[mid] < v[lo]) swapf(&v[mid], &v[lo]);
        if (v[hi] < v[lo]) swapf(&v[hi], &v[lo]);
        if (v[hi] < v[mid]) swapf(&v[hi], &v[mid]);
        const float pivot = v[mid];
        size_t i = lo, j = hi;
        while (i <= j) {
            while (v[i] < pivot) i++;
            while (v[j] > pivot) j--;
            if (i <= j) {
                swapf(&v[i], &v[j]);
                i++;
                if (j == 0) break;
                j--;
            }
        }
        if (k <= j) hi = j;
        else if (k >= i) lo = i;
        else return;
    }
}

/* one quantile of x[0..n) using caller scratch (n floats); util.c:117-125 */
static float quantile_scratch(const float *x, size_t n, float p, float *scratch) {
    memcpy(scratch, x, n * sizeof(float));
    const size_t idx = (size_t)(p * (n - 1));
    const float remf = p * (n - 1) - idx;
    select_kth(scratch, n, idx);
    const float a = scratch[idx];
    if (idx < n - 1) {
        float b = scratch[idx + 1];
        for (size_t i = idx + 2; i < n; i++) if (scratch[i] < b) b = scratch[i];
        return (float)((1.0 - remf) * a + remf * b);
    }
    return a;
}

float sh_medianf(const float *x, size_t n, float *scratch) {
    return quantile_scratch(x, n, 0.5f, scratch);
}

/* util.c:156-180; scratch holds 2n floats */
float sh_madf(const float *x, size_t n, const float *med, float *scratch) {
    const float mad_scaling_factor = 1.4826;
    if (n == 1) return 0.0f;
    const float m = med ? *med : sh_medianf(x, n, scratch);
    float *absdiff = scratch + n;
    for (size_t i = 0; i < n; i++) absdiff[i] = fabsf(x[i] - m);
    return sh_medianf(absdiff, n, scratch) * mad_scaling_factor;
}

/* util.c:190-205 */
void medmad_normalise_array(float *x, size_t n) {
    if (!x || n == 0) return;
    if (n == 1) { x[0] = 0.0f; return; }
    float *scratch = malloc(2 * n * sizeof(float));
    if (!scratch) return;
    const float xmed = sh_medianf(x, n, scratch);
    const float xmad = sh_madf(x, n, &xmed, scratch);
    for (size_t i = 0; i < n; i++) x[i] = (x[i] - xmed) / xmad;
    free(scratch);
}

/* scrappie_common.c:39-73 */
raw_table trim_raw_by_mad(raw_table rt, size_t chunk_size, float perc) {
    const size_t nsample = rt.end - rt.start;
    const size_t nchunk = nsample / chunk_size;
    rt.end = nchunk * chunk_size;            /* relative to 0, as the reference (Q14) */
    if (nchunk == 0) return rt;
    float *madarr = malloc(nchunk * sizeof(float));
    float *scratch = malloc(2 * (chunk_size > nchunk ? chunk_size : nchunk) * sizeof(float));
    if (!madarr || !scratch) {
        free(madarr); free(scratch);
        return (raw_table){0};
    }
    for (size_t i = 0; i < nchunk; i++)
        madarr[i] = sh_madf(rt.raw + rt.start + i * chunk_size, chunk_size, NULL, scratch);
    const float thresh = quantile_scratch(madarr, nchunk, perc, scratch);
    for (size_t i = 0; i < nchunk; i++) {
        if (madarr[i] > thresh) break;
        rt.start += chunk_size;
    }
    for (size_t i = nchunk; i > 0; i--) {
        if (madarr[i - 1] > thresh) break;
        rt.end -= chunk_size;
    }
    free(scratch);
    free(madarr);
    return rt;
}

/* scrappie_common.c:5-21; like the reference, frees rt.raw when the trimmed
 * window is empty */
raw_table trim_and_segment_raw(raw_table rt, size_t trim_start, size_t trim_end,
                               size_t varseg_chunk, float varseg_thresh) {
    if (!rt.raw) return (raw_table){0};
    rt = trim_raw_by_mad(rt, varseg_chunk, varseg_thresh);
    if (!rt.raw) return (raw_table){0};
    rt.start = (rt.n - rt.start) > trim_start ? rt.start + trim_start : rt.n;
    rt.end = (rt.end > trim_end) ? rt.end - trim_end : 0;
    if (rt.start >= rt.end) {
        free(rt.raw);
        return (raw_table){0};
    }
    return rt;
}

/* ------------------------------------------------------------------ */
/* D3 k-mer stitching (decode.c:367-509)                               */
/* ------------------------------------------------------------------ */
static const char BASES[4] = { 'A', 'C', 'G', 'T' };

static inline int kmer_shift(int k1, int k2, int nkmer) {
    /* smallest s >= 1 with suffix_{k-s}(k1) == prefix_{k-s}(k2) */
    int mask = nkmer - 1, s = 0;
    do {
        mask >>= 2;
        k1 &= mask;
        k2 >>= 2;
        s++;
    } while (k1 != k2);
    return s;
}

char *overlapper(const int *seq, size_t n, int nkmer, int *pos) {
    if (!seq) return NULL;
    size_t nbit = 0;
    for (size_t x = (size_t)nkmer; x; x >>= 1) nbit++;
    const size_t klen = nbit / 2;
    size_t first = 0;
    while (first < n && seq[first] < 0) first++;
    if (first == n) return NULL;

    size_t length = klen;
    for (size_t k = first + 1, prev = (size_t)seq[first]; k < n; k++) {
        if (seq[k] < 0) continue;
        length += (size_t)kmer_shift((int)prev, seq[k], nkmer);
        prev = (size_t)seq[k];
    }
    char *bases = calloc(length + 1, 1);
    if (!bases) return NULL;
    for (size_t kmer = (size_t)seq[first], i = klen; i-- > 0; kmer >>= 2)
        bases[i] = BASES[kmer & 3];
    if (pos) pos[0] = 0;
    size_t tail = klen - 1;
    int prev = seq[first];
    for (size_t k = first + 1; k < n; k++) {
        if (seq[k] < 0) {
            if (pos) pos[k] = pos[k - 1];
            continue;
        }
        const int s = kmer_shift(prev, seq[k], nkmer);
        if (pos) pos[k] = pos[k - 1] + s;
        prev = seq[k];
        size_t kmer = (size_t)seq[k];
        for (int i = 0; i < s; i++, kmer >>= 2)
            bases[tail + (size_t)(s - i)] = BASES[kmer & 3];
        tail += (size_t)s;
    }
    return bases;
}

/* decode.c:895-918 (pos is accepted and left untouched, as there: Q11) */
char *crfpath_to_basecall(int const *path, size_t npos, int *pos) {
    if (!path || !pos) return NULL;
    size_t nb = 0;
    for (size_t i = 0; i < npos; i++) nb += (path[i] < 4);
    char *out = calloc(nb + 1, 1);
    if (!out) return NULL;
    for (size_t i = 0, j = 0; i < npos; i++)
        if (path[i] < 4) out[j++] = BASES[path[i]];
    return out;
}

/* ------------------------------------------------------------------ */
/* D2 homopolymer correction (homopolymer.c:67-235) on a 5-row side     */
/* buffer: side[t*5 + {0..3}] = log-posterior of the homopolymer k-mer  */
/* of base A,C,G,T at block t, side[t*5 + 4] = stay.  Only those five   */
/* rows are ever read by the reference (homopolymer.c:200,209-210).     */
/* ------------------------------------------------------------------ */
static inline int repeat_kmer(int b, int k) {   /* scrappie_seq_helpers.c:115 */
    int y = 0;
    for (int i = 0; i < k; i++) y = y * 4 + b;
    return y;
}

int sh_kmerlength(int nstate) {                 /* scrappie_seq_helpers.c:132 */
    return (int)(logf((float)nstate) / logf(4.0f));
}

int sh_homopolymer_side(const float *side, int *path, int nblock, int nstate) {
    const int klen = sh_kmerlength(nstate);
    const int fkm1 = 1 << (2 * (klen - 1)), fkm2 = 1 << (2 * (klen - 2));
    const int cap = nblock / 2;
    if (cap <= 0) return 0;
    int *runs = malloc(3 * (size_t)cap * sizeof(int));
    if (!runs) return -1;
    int *rstart = runs, *rlen = runs + cap, *rbase = runs + 2 * cap;
    int nrun = 0;
    /* candidate runs, base by base, in path order (homopolymer.c:95-138) */
    for (int b = 0; b < 4; b++) {
        const int hk = repeat_kmer(b, klen), hk1 = repeat_kmer(b, klen - 1), hk2 = repeat_kmer(b, klen - 2);
        for (int i = 1; i < nblock - 2; i++) {
            const int p = path[i - 1], q = path[i];
            const int q_ok = (q == -1) || (q == hk);
            if (p != -1 && p != hk && (p % fkm1) == hk1 && q_ok) {
                int e = i + 1;
                while (e < nblock && (path[e] == -1 || path[e] == hk)) e++;
                if (nrun < cap) { rstart[nrun] = i; rlen[nrun] = e - i; rbase[nrun] = b; nrun++; }
            }
            if (p != -1 && (p % fkm2) == hk2 && (p % fkm1) != hk1 && q_ok) {
                int j = i;
                while (j < nblock && path[j] == -1) j++;
                if (path[j] == hk && j < nblock - 1) {
                    int e = j + 1;
                    while (e < nblock && (path[e] == -1 || path[e] == hk)) e++;
                    if (nrun < cap) { rstart[nrun] = j; rlen[nrun] = e - j; rbase[nrun] = b; nrun++; }
                }
            }
        }
    }
    /* replace the Viterbi count of each run by the posterior mean count */
    for (int r = 0; r < nrun; r++) {
        const int hk = repeat_kmer(rbase[r], klen);
        const int from = rstart[r], to = from + rlen[r] - 1;
        int nvit = 0;
        double nmean = 0.0;
        for (int i = from; i <= to; i++) {
            const float *s = side + (size_t)(i - 1) * 5;     /* block i-1 pairs with path[i] (Q8) */
            const double ps = expf(s[4]), pr = expf(s[rbase[r]]);
            nmean += pr / (pr + ps);
            nvit += (path[i] == hk);
        }
        const int newn = (int)(nmean + 0.5);
        if (newn != nvit)
            for (int i = 0; i <= to - from; i++) path[from + i] = (i < newn) ? hk : -1;
    }
    free(runs);
    return 0;
}

/* homopolymer.c:175 on a full posterior matrix (per-read surface) */
int homopolymer_path(const_scrappie_matrix post, int *viterbipath,
                     enum homopolymer_calculation flag) {
    if (flag != HOMOPOLYMER_MEAN) return 0;
    if (!post || !viterbipath) return -1;
    const int T = (int)post->nc, ns = (int)post->nr;
    const int klen = sh_kmerlength(ns);
    float *side = malloc((size_t)T * 5 * sizeof(float));
    if (!side) return -1;
    for (int t = 0; t < T; t++) {
        const float *col = post->data.f + (size_t)t * post->stride;
        for (int b = 0; b < 4; b++) side[t * 5 + b] = col[repeat_kmer(b, klen)];
        side[t * 5 + 4] = col[ns - 1];
    }
    const int rc = sh_homopolymer_side(side, viterbipath, T, ns);
    free(side);
    return rc;
}

/* ------------------------------------------------------------------ */
/* D5 posterior_crf (decode.c:928-1012): optional per-block state       */
/* posterior for the CRF model; O(25 T) with libm, kept on the host.    */
/* ------------------------------------------------------------------ */
static inline float lse2(float x, float y) {       /* util.h:162 */
    return fmaxf(x, y) + log1pf(expf(-fabsf(x - y)));
}

scrappie_matrix posterior_crf(const_scrappie_matrix trans) {
    if (!trans) return NULL;
    const size_t ns = (size_t)roundf(sqrtf((float)trans->nr));
    const size_t T = trans->nc;
    scrappie_matrix post = make_scrappie_matrix(ns, T + 1);
    float *bwd = malloc(2 * ns * sizeof(float));
    if (!post || !bwd) { free(bwd); return free_scrappie_matrix(post); }
    /* forward messages into post columns 1..T (column 0 stays 0) */
    for (size_t t = 0; t < T; t++) {
        const float *tr = trans->data.f + t * trans->stride;
        const float *a = post->data.f + t * post->stride;
        float *c = post->data.f + (t + 1) * post->stride;
        for (size_t to = 0; to < ns; to++) {
            float acc = tr[to * ns] + a[0];
            for (size_t fr = 1; fr < ns; fr++) acc = lse2(acc, tr[to * ns + fr] + a[fr]);
            c[to] = acc;
        }
    }
    float *prev = bwd, *curr = bwd + ns;
    for (size_t s = 0; s < ns; s++) curr[s] = 0.0f;
    {   /* last column: normalise (accumulator starts at 0.0f like the reference) */
        float *last = post->data.f + T * post->stride, tot = 0.0f;
        for (size_t s = 0; s < ns; s++) tot = lse2(tot, last[s]);
        for (size_t s = 0; s < ns; s++) last[s] = expf(last[s] - tot);
    }
    for (size_t t = T; t-- > 0;) {
        const float *tr = trans->data.f + t * trans->stride;
        float *col = post->data.f + t * post->stride;
        { float *x = curr; curr = prev; prev = x; }
        for (size_t s = 0; s < ns; s++) curr[s] = tr[s] + prev[0];
        for (size_t to = 1; to < ns; to++)
            for (size_t fr = 0; fr < ns; fr++)
                curr[fr] = lse2(curr[fr], tr[to * ns + fr] + prev[to]);
        float tot = 0.0f;
        for (size_t s = 0; s < ns; s++) { col[s] += curr[s]; tot = lse2(tot, col[s]); }
        for (size_t s = 0; s < ns; s++) col[s] = expf(col[s] - tot);
    }
    free(bwd);
    return post;
}

/* ------------------------------------------------------------------ */
/* T4 model names (networks.c:17-34, :49-68)                            */
/* ------------------------------------------------------------------ */
static const char *const MODEL_NAMES[] = { "raw_r94", "rgrgr_r94", "rgrgr_r941", "rgrgr_r10", "rnnrf_r94" };

enum raw_model_type get_raw_model(const char *modelstr) {
    if (modelstr)
        for (int i = 0; i < 5; i++)
            if (0 == strcmp(modelstr, MODEL_NAMES[i])) return (enum raw_model_type)i;
    return SCRAPPIE_MODEL_INVALID;
}

const char *raw_model_string(const enum raw_model_type model) {
    if ((int)model < 0 || model >= SCRAPPIE_MODEL_INVALID) {
        /* the reference calls errx(EXIT_FAILURE, ...) here (networks.c:61-64) */
        fprintf(stderr, "Invalid scrappie model %s:%d\n", __FILE__, __LINE__);
        exit(EXIT_FAILURE);
    }
    return MODEL_NAMES[model];
}

/* ------------------------------------------------------------------ */
/* O1 output records (scrappie_raw.c:317-331)                           */
/* ------------------------------------------------------------------ */
int scrappie_hip_format_fasta(char *buf, size_t buflen, const char *uuid, const char *readname,
                              bool uuid_primary, const char *prefix, const scrappie_hip_call *res,
                              size_t nsample, size_t trim_start, size_t trim_end) {
    if (!uuid) uuid = "";
    return snprintf(buf, buflen,
                    ">%s%s  { \"filename\" : \"%s\", \"uuid\" : \"%s\", \"normalised_score\" : %f,  "
                    "\"nblock\" : %zu,  \"sequence_length\" : %zu,  \"blocks_per_base\" : %f, "
                    "\"nsample\" : %zu, \"trim\" : [ %zu, %zu ] }\n%s\n",
                    prefix ? prefix : "", uuid_primary ? uuid : readname, readname, uuid,
                    -res->score / res->nblock, res->nblock, res->basecall_length,
                    (float)res->nblock / (float)res->basecall_length,
                    nsample, trim_start, trim_end, res->basecall);
}

int scrappie_hip_format_sam(char *buf, size_t buflen, const char *uuid, const char *readname,
                            bool uuid_primary, const char *prefix, const scrappie_hip_call *res) {
    if (!uuid) uuid = "";
    return snprintf(buf, buflen, "%s%s\t4\t*\t0\t0\t*\t*\t0\t0\t%s\t*\n", prefix ? prefix : "",
                    uuid_primary ? uuid : readname, res->basecall);
}


/* ------------------------------------------------------------------ */
/* events features                                      */
/* ------------------------------------------------------------------ */
#if defined(__SSE__) || defined(__x86_64__)
#include <xmmintrin.h>
#endif

/* nanonet_features_from_events(et, true) (nnfeatures.c:51-110) followed by window(., 3, 1)
 * (layers.c:119-147).  The studentisation uses the hardware reciprocal-sqrt estimate as the
 * reference does (rsqrtps); the window leaves output column 0 zero (its loop compares an int
 * with a size_t and a negative start index ends it at once). */
int scrappie_hip_event_features(const event_table et, float *out) {
    if (!et.event || !out || et.end <= et.start) return -1;
    const size_t n = et.end - et.start;
    float *f = malloc(n * 4 * sizeof(float));
    if (!f) return -1;
    for (size_t ev = 0; ev < n; ev++) {
        const event_t *e = et.event + et.start + ev;
        f[4 * ev + 0] = e->mean;
        f[4 * ev + 1] = e->stdv;
        f[4 * ev + 2] = e->length;
        f[4 * ev + 3] = (ev + 1 < n) ? (float)fabs(e->mean - e[1].mean) : 0.0f;
    }
    float sum[4] = {0, 0, 0, 0}, sumsq[4] = {0, 0, 0, 0}, comp[4] = {0, 0, 0, 0}, compsq[4] = {0, 0, 0, 0};
    for (size_t ev = 0; ev < n; ev++)
        for (int k = 0; k < 4; k++) {
            const float x = f[4 * ev + k];
            const float d1 = x - comp[k];
            const float s1 = sum[k] + d1;
            comp[k] = (s1 - sum[k]) - d1;
            sum[k] = s1;
            const float d2 = x * x - compsq[k];
            const float s2 = sumsq[k] + d2;
            compsq[k] = (s2 - sumsq[k]) - d2;
            sumsq[k] = s2;
        }
    float scale[4], shift[4];
    for (int k = 0; k < 4; k++) {
        sum[k] /= (float)(int)n;
        sumsq[k] /= (float)(int)n;
        sumsq[k] -= sum[k] * sum[k];
    }
#if defined(__SSE__) || defined(__x86_64__)
    _mm_storeu_ps(scale, _mm_rsqrt_ps(_mm_loadu_ps(sumsq)));
#else
    for (int k = 0; k < 4; k++) scale[k] = 1.0f / sqrtf(sumsq[k]);
#endif
    for (int k = 0; k < 4; k++) shift[k] = sum[k] * scale[k];
    for (size_t ev = 0; ev < n; ev++)
        for (int k = 0; k < 4; k++) f[4 * ev + k] = scale[k] * f[4 * ev + k] - shift[k];
    memset(out, 0, n * 12 * sizeof(float));
    for (size_t col = 1; col < n; col++)                 /* column 0 stays zero */
        for (int w = 0; w < 3; w++) {
            const size_t src = col - 1 + (size_t)w;
            if (src < n) memcpy(out + col * 12 + 4 * w, f + 4 * src, 4 * sizeof(float));
        }
    free(f);
    return 0;
}
