/* oracle/oracle.c -- scalar C restatement of the `scrappie raw` hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Build with
 *   gcc -O2 -std=c99 -ffp-contract=off -fno-fast-math
 * so every float operation is a single IEEE binary32 operation, as in the
 * reference's Release build (-std=c99 implies -ffp-contract=off).
 *
 * Citations are file:line under /root/reference/src.
 */
#define _POSIX_C_SOURCE 200809L
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define BIG_FLOAT 1.e30f   /* decode.c:8 */

/* ------------------------------------------------------------------ */
/* T1 / T2 containers                                                 */
/* ------------------------------------------------------------------ */

/* scrappie_matrix.c:11-42: rows padded to a multiple of 4 floats, 16-byte
 * aligned, zero filled. */
orc_mat *orc_make_mat(size_t nr, size_t nc) {
    if (nr == 0 || nc == 0) return NULL;
    orc_mat *m = malloc(sizeof(*m));
    if (!m) return NULL;
    m->nr = nr;
    m->nrq = (nr + 3) / 4;
    m->nc = nc;
    m->stride = 4 * m->nrq;
    void *p = NULL;
    if (posix_memalign(&p, 16, m->stride * nc * sizeof(float)) != 0) {
        free(m);
        return NULL;
    }
    memset(p, 0, m->stride * nc * sizeof(float));
    m->data.v = p;
    return m;
}

/* scrappie_matrix.c:44-51: reallocate only on shape change */
orc_mat *orc_remake_mat(orc_mat *M, size_t nr, size_t nc) {
    if (M == NULL || M->nr != nr || M->nc != nc) {
        orc_free_mat(M);
        M = orc_make_mat(nr, nc);
    }
    return M;
}

/* scrappie_matrix.c:130-136 */
orc_mat *orc_free_mat(orc_mat *M) {
    if (M) {
        free(M->data.v);
        free(M);
    }
    return NULL;
}

/* scrappie_matrix.c:69-78 */
orc_mat *orc_mat_from_array(const float *x, size_t nr, size_t nc) {
    orc_mat *m = orc_make_mat(nr, nc);
    if (!m) return NULL;
    for (size_t c = 0; c < nc; c++)
        memcpy(m->data.f + c * m->stride, x + c * nr, nr * sizeof(float));
    return m;
}

/* scrappie_matrix.c:80-97 */
float *orc_array_from_mat(const orc_mat *M) {
    if (!M) return NULL;
    float *res = calloc(M->nr * M->nc, sizeof(float));
    if (!res) return NULL;
    for (size_t c = 0; c < M->nc; c++)
        for (size_t r = 0; r < M->nr; r++)
            res[c * M->nr + r] = M->data.f[c * M->stride + r];
    return res;
}

/* scrappie_matrix.c:269 (int32 twin) */
orc_imat *orc_make_imat(size_t nr, size_t nc) {
    if (nr == 0 || nc == 0) return NULL;
    orc_imat *m = malloc(sizeof(*m));
    if (!m) return NULL;
    m->nr = nr;
    m->nrq = (nr + 3) / 4;
    m->nc = nc;
    m->stride = 4 * m->nrq;
    void *p = NULL;
    if (posix_memalign(&p, 16, m->stride * nc * sizeof(int32_t)) != 0) {
        free(m);
        return NULL;
    }
    memset(p, 0, m->stride * nc * sizeof(int32_t));
    m->data.v = p;
    return m;
}

orc_imat *orc_free_imat(orc_imat *M) {
    if (M) {
        free(M->data.v);
        free(M);
    }
    return NULL;
}

/* ------------------------------------------------------------------ */
/* A1 vector math, one lane at a time                                  */
/* ------------------------------------------------------------------ */

static inline float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* sse_mathfun.h:225-301 (exp_ps): Cephes range reduction + degree-5
 * polynomial, argument clamped to +-88.3762626647949, 2^n built by integer
 * shift (so n = -127 gives exactly 0, not a denormal). */
float orc_expf(float x) {
    const float hi = 88.3762626647949f, lo = -88.3762626647949f;
    x = (x < hi) ? x : hi;   /* _mm_min_ps(x, hi) */
    x = (x > lo) ? x : lo;   /* _mm_max_ps(x, lo) */
    float fx = x * (float)1.44269504088896341;
    fx = fx + 0.5f;
    float tmp = (float)(int32_t)fx;          /* cvttps + cvtepi32 */
    if (tmp > fx) tmp = tmp - 1.0f;          /* floor */
    fx = tmp;
    tmp = fx * (float)0.693359375;
    float z = fx * (float)-2.12194440e-4;
    x = x - tmp;
    x = x - z;
    z = x * x;
    float y = (float)1.9875691500E-4;
    y = y * x; y = y + (float)1.3981999507E-3;
    y = y * x; y = y + (float)8.3334519073E-3;
    y = y * x; y = y + (float)4.1665795894E-2;
    y = y * x; y = y + (float)1.6666665459E-1;
    y = y * x; y = y + (float)5.0000001201E-1;
    y = y * z;
    y = y + x;
    y = y + 1.0f;
    int32_t n = (int32_t)fx;
    float pow2n = bits2f((uint32_t)(n + 0x7f) << 23);
    return y * pow2n;
}

/* sse_mathfun.h:123-209 (log_ps): denormals clamped up to FLT_MIN,
 * x <= 0 gives NaN. */
float orc_logf(float x) {
    const bool invalid = (x <= 0.0f);
    const float min_norm = bits2f(0x00800000u);
    x = (x > min_norm) ? x : min_norm;       /* _mm_max_ps(x, min_norm_pos) */
    uint32_t u = f2bits(x);
    int32_t emm0 = (int32_t)(u >> 23);
    u = (u & ~0x7f800000u) | f2bits(0.5f);
    x = bits2f(u);
    emm0 -= 0x7f;
    float e = (float)emm0;
    e = e + 1.0f;
    const bool small = x < (float)0.707106781186547524;
    float tmp = small ? x : 0.0f;
    x = x - 1.0f;
    if (small) e = e - 1.0f;
    x = x + tmp;
    float z = x * x;
    float y = (float)7.0376836292E-2;
    y = y * x; y = y + (float)-1.1514610310E-1;
    y = y * x; y = y + (float)1.1676998740E-1;
    y = y * x; y = y + (float)-1.2420140846E-1;
    y = y * x; y = y + (float)1.4249322787E-1;
    y = y * x; y = y + (float)-1.6668057665E-1;
    y = y * x; y = y + (float)2.0000714765E-1;
    y = y * x; y = y + (float)-2.4999993993E-1;
    y = y * x; y = y + (float)3.3333331174E-1;
    y = y * x;
    y = y * z;
    tmp = e * (float)-2.12194440e-4;
    y = y + tmp;
    tmp = z * 0.5f;
    y = y - tmp;
    tmp = e * (float)0.693359375;
    x = x + y;
    x = x + tmp;
    return invalid ? NAN : x;
}

/* util.h:180-183 */
float orc_logisticf(float x) { return 1.0f / (1.0f + orc_expf(-x)); }

/* util.h:185-188 */
float orc_tanhf(float x) {
    const float y = orc_logisticf(x + x);
    return (y + y) - 1.0f;
}

/* util.h:190-198: exact identity for x >= 0 (including -0, which compares
 * >= 0), exp(x) - 1 otherwise (not expm1). */
float orc_eluf(float x) { return (x >= 0.0f) ? x : (orc_expf(x) - 1.0f); }

/* util.h:162-164 (libm, not the Cephes vectors) */
float orc_logsumexpf(float x, float y) {
    return fmaxf(x, y) + log1pf(expf(-fabsf(x - y)));
}

/* ------------------------------------------------------------------ */
/* P0 signal preparation                                               */
/* ------------------------------------------------------------------ */

/* util.c:69-75: never returns 0 */
static int floatcmp(const void *x, const void *y) {
    float d = *(const float *)x - *(const float *)y;
    return (d > 0) ? 1 : -1;
}

/* util.c:92-130 */
void orc_quantilef(const float *x, size_t nx, float *p, size_t np) {
    if (!p) return;
    float *space = x ? malloc(nx * sizeof(float)) : NULL;
    if (!space) {
        for (size_t i = 0; i < np; i++) p[i] = NAN;
        return;
    }
    memcpy(space, x, nx * sizeof(float));
    qsort(space, nx, sizeof(float), floatcmp);
    for (size_t i = 0; i < np; i++) {
        /* p[i] * (nx - 1): float * size_t converts the integer to float */
        const size_t idx = (size_t)(p[i] * (nx - 1));
        const float remf = p[i] * (nx - 1) - idx;
        if (idx < nx - 1) {
            /* (1.0 - remf) is double arithmetic in the reference */
            p[i] = (float)((1.0 - remf) * space[idx] + remf * space[idx + 1]);
        } else {
            p[i] = space[idx];
        }
    }
    free(space);
}

/* util.c:142-146 */
float orc_medianf(const float *x, size_t n) {
    float p = 0.5f;
    orc_quantilef(x, n, &p, 1);
    return p;
}

/* util.c:156-180 */
float orc_madf(const float *x, size_t n, const float *med) {
    const float mad_scaling_factor = 1.4826;
    if (!x) return NAN;
    if (n == 1) return 0.0f;
    float *absdiff = malloc(n * sizeof(float));
    if (!absdiff) return NAN;
    const float m = med ? *med : orc_medianf(x, n);
    for (size_t i = 0; i < n; i++) absdiff[i] = fabsf(x[i] - m);
    const float mad = orc_medianf(absdiff, n);
    free(absdiff);
    return mad * mad_scaling_factor;
}

/* util.c:190-205 */
void orc_medmad_normalise_array(float *x, size_t n) {
    if (!x) return;
    if (n == 1) { x[0] = 0.0f; return; }
    const float xmed = orc_medianf(x, n);
    const float xmad = orc_madf(x, n, &xmed);
    for (size_t i = 0; i < n; i++) x[i] = (x[i] - xmed) / xmad;
}

/* scrappie_common.c:39-73.  Quirk Q14: end = nchunk*chunk is relative to 0. */
orc_raw_table orc_trim_raw_by_mad(orc_raw_table rt, size_t chunk_size, float perc) {
    const size_t nsample = rt.end - rt.start;
    const size_t nchunk = nsample / chunk_size;
    rt.end = nchunk * chunk_size;
    float *madarr = malloc((nchunk ? nchunk : 1) * sizeof(float));
    if (!madarr) return (orc_raw_table){0};
    for (size_t i = 0; i < nchunk; i++)
        madarr[i] = orc_madf(rt.raw + rt.start + i * chunk_size, chunk_size, NULL);
    orc_quantilef(madarr, nchunk, &perc, 1);
    const float thresh = perc;
    for (size_t i = 0; i < nchunk; i++) {
        if (madarr[i] > thresh) break;
        rt.start += chunk_size;
    }
    for (size_t i = nchunk; i > 0; i--) {
        if (madarr[i - 1] > thresh) break;
        rt.end -= chunk_size;
    }
    free(madarr);
    return rt;
}

/* scrappie_common.c:5-21 */
orc_raw_table orc_trim_and_segment_raw(orc_raw_table rt, size_t trim_start,
                                       size_t trim_end, size_t varseg_chunk,
                                       float varseg_thresh) {
    if (!rt.raw) return (orc_raw_table){0};
    rt = orc_trim_raw_by_mad(rt, varseg_chunk, varseg_thresh);
    if (!rt.raw) return (orc_raw_table){0};
    rt.start = (rt.n - rt.start) > trim_start ? rt.start + trim_start : rt.n;
    rt.end = (rt.end > trim_end) ? rt.end - trim_end : 0;
    if (rt.start >= rt.end) return (orc_raw_table){0};
    return rt;
}

/* ------------------------------------------------------------------ */
/* F0                                                                  */
/* ------------------------------------------------------------------ */

/* nnfeatures.c:102-116: one sample per padded 4-float column */
orc_mat *orc_features_from_raw(orc_raw_table signal) {
    if (signal.n == 0 || !signal.raw) return NULL;
    const size_t nsample = signal.end - signal.start;
    orc_mat *m = orc_make_mat(1, nsample);
    if (!m) return NULL;
    for (size_t i = 0; i < nsample; i++)
        m->data.f[i * 4] = signal.raw[i + signal.start];
    return m;
}

/* ------------------------------------------------------------------ */
/* BLAS stand-ins used ONLY inside the oracle (the two call shapes the */
/* reference uses: layers.c:193,220,237,505,517; scrappie_matrix.c:346)*/
/* Summation in index order, binary32 accumulator.                     */
/* ------------------------------------------------------------------ */

/* Optional BLAS back-end, used ONLY by bench.py's cpu_baseline leg so that the
 * timed CPU path is, like the reference, "threads over reads + single-threaded
 * OpenBLAS" (README.md:68-71) rather than these naive loops.  The checker never
 * sets it.  Signatures are the two cblas entry points the reference calls. */
typedef void (*orc_sgemv_fn)(int order, int trans, int M, int N, float alpha, const float *A, int lda,
                             const float *X, int incX, float beta, float *Y, int incY);
typedef void (*orc_sgemm_fn)(int order, int transA, int transB, int M, int N, int K, float alpha,
                             const float *A, int lda, const float *B, int ldb, float beta, float *C, int ldc);
static orc_sgemv_fn blas_sgemv = NULL;
static orc_sgemm_fn blas_sgemm = NULL;
void orc_set_blas(void *sgemv, void *sgemm) {
    blas_sgemv = (orc_sgemv_fn)sgemv;
    blas_sgemm = (orc_sgemm_fn)sgemm;
}

/* y[j] += sum_{i<M} A[i + j*lda] * x[i]   (ColMajor, Trans, alpha=beta=1) */
static void sgemv_t(size_t M, size_t N, const float *A, size_t lda,
                    const float *x, float *y) {
    if (blas_sgemv) { blas_sgemv(102, 112, (int)M, (int)N, 1.0f, A, (int)lda, x, 1, 1.0f, y, 1); return; }
#ifdef ORC_FAST_LOOPS
    /* bench.py's cpu_baseline build only (never the checker): the same dot products with 16 partial sums,
     * an order the compiler can keep in vector registers without -ffast-math */
    for (size_t j = 0; j < N; j++) {
        const float *a = A + j * lda;
        float part[16] = {0};
        size_t i = 0;
        for (; i + 16 <= M; i += 16)
            for (int k = 0; k < 16; k++) part[k] += a[i + k] * x[i + k];
        float acc = 0.0f;
        for (; i < M; i++) acc += a[i] * x[i];
        for (int k = 0; k < 16; k++) acc += part[k];
        y[j] += acc;
    }
    return;
#endif
    for (size_t j = 0; j < N; j++) {
        float acc = 0.0f;
        const float *a = A + j * lda;
        for (size_t i = 0; i < M; i++) acc += a[i] * x[i];
        y[j] += acc;
    }
}

/* C[m + n*ldc] += sum_{k<K} A[k + m*lda] * B[k + n*ldb]  (Trans, NoTrans) */
static void sgemm_tn(size_t M, size_t N, size_t K, const float *A, size_t lda,
                     const float *B, size_t ldb, float *C, size_t ldc) {
    if (blas_sgemm) { blas_sgemm(102, 112, 111, (int)M, (int)N, (int)K, 1.0f, A, (int)lda, B, (int)ldb, 1.0f, C, (int)ldc); return; }
#ifdef ORC_FAST_LOOPS
    for (size_t n = 0; n < N; n++)
        for (size_t m = 0; m < M; m++) {
            const float *a = A + m * lda, *b = B + n * ldb;
            float part[16] = {0};
            size_t k = 0;
            for (; k + 16 <= K; k += 16)
                for (int j = 0; j < 16; j++) part[j] += a[k + j] * b[k + j];
            float acc = 0.0f;
            for (; k < K; k++) acc += a[k] * b[k];
            for (int j = 0; j < 16; j++) acc += part[j];
            C[m + n * ldc] += acc;
        }
    return;
#endif
    for (size_t n = 0; n < N; n++)
        for (size_t m = 0; m < M; m++) {
            float acc = 0.0f;
            const float *a = A + m * lda, *b = B + n * ldb;
            for (size_t k = 0; k < K; k++) acc += a[k] * b[k];
            C[m + n * ldc] += acc;
        }
}

static inline size_t iceil_(size_t x, size_t y) { return (x + y - 1) / y; }

/* ------------------------------------------------------------------ */
/* C1 convolution -- layers.c:159-246, index for index (quirk Q1)      */
/* ------------------------------------------------------------------ */
orc_mat *orc_convolution(const orc_mat *X, const orc_mat *W, const orc_mat *b,
                         size_t stride, orc_mat *C) {
    if (!X) return NULL;
    const size_t winlen = W->nrq / X->nrq;               /* :169 */
    const size_t nfilter = W->nc;
    const size_t padL = (winlen - 1) / 2;                /* :172 */
    const size_t padR = winlen / 2;                      /* :173 */
    const size_t ncolC = iceil_(X->nc, stride);          /* :174 */
    C = orc_remake_mat(C, nfilter, ncolC);
    if (!C) return NULL;
    const size_t ldC = C->stride, ldW = W->stride, ldX = X->stride;
    const size_t ldFeature = ldX;

    for (size_t i = 0; i < C->nc; i++)                   /* :185 bias */
        memcpy(C->data.f + i * ldC, b->data.f, ldC * sizeof(float));

    /* left edge :190-196 */
    for (size_t w = 0; w < padL; w += stride) {
        const size_t offsetW = ldFeature * (padL - w);
        const size_t ncol = w / stride;
        sgemv_t(W->nr - offsetW, W->nc, W->data.f + offsetW, ldW,
                X->data.f, C->data.f + ldC * ncol);
    }

    const size_t ncolsL_complete = iceil_(padL, stride); /* :199 */
    const size_t offsetC_L = ldC * ncolsL_complete;
    const size_t shiftX_L = ncolsL_complete * stride - padL;
    const size_t offsetX_L = shiftX_L * ldX;
    const size_t nstepC = iceil_(winlen, stride);        /* :206 */
    const size_t nstepX = stride * nstepC;

    /* interleaved strided GEMMs :209-224; ifloor() takes ints */
    for (size_t w = 0; w < winlen; w += stride) {
        const int ncol_processed = (int)(X->nc - shiftX_L - w) / (int)nstepX;
        const size_t initial_col = w / stride;
        if (ncol_processed > 0)
            sgemm_tn(W->nc, (size_t)ncol_processed, W->nr, W->data.f, ldW,
                     X->data.f + ldX * w + offsetX_L, ldX * nstepX,
                     C->data.f + ldC * initial_col + offsetC_L, ldC * nstepC);
    }

    /* right edge :227-241 */
    const size_t maxCol_reshape = (size_t)((int)(X->nc - shiftX_L) / (int)nstepX);
    const size_t remainder_reshape = (X->nc - shiftX_L) % nstepX;
    const size_t offsetC_R = offsetC_L + ldC * nstepC * (maxCol_reshape - 1)
                           + ldC * (remainder_reshape / stride) + ldC;
    const size_t offsetX_R = (X->nc - winlen + 1) * ldX;
    const int startR = (int)(stride - (padL + X->nc - winlen) % stride - 1);
    for (size_t w = (size_t)startR; w < padR; w += stride) {
        const size_t offsetW = ldFeature * (w + 1);
        const size_t col_off = offsetC_R + ldC * (w / stride);
        if (col_off / ldC >= C->nc) continue;  /* reference would write out of bounds */
        sgemv_t(W->nr - offsetW, W->nc, W->data.f, ldW,
                X->data.f + offsetX_R + ldX * w, C->data.f + col_off);
    }
    return C;
}

/* ------------------------------------------------------------------ */
/* activations (applied to padding lanes too: quirk Q4)                */
/* ------------------------------------------------------------------ */
void orc_tanh_activation_inplace(orc_mat *C) {           /* layers.c:15 */
    if (!C) return;
    for (size_t i = 0; i < C->stride * C->nc; i++) C->data.f[i] = orc_tanhf(C->data.f[i]);
}
void orc_exp_activation_inplace(orc_mat *C) {            /* layers.c:30 */
    if (!C) return;
    for (size_t i = 0; i < C->stride * C->nc; i++) C->data.f[i] = orc_expf(C->data.f[i]);
}
void orc_elu_activation_inplace(orc_mat *C) {            /* layers.c:60 */
    if (!C) return;
    for (size_t i = 0; i < C->stride * C->nc; i++) C->data.f[i] = orc_eluf(C->data.f[i]);
}
/* layers.c:79-94: log(min_prob + (1 - min_prob) * p)  -- the code, not the
 * doc comment (quirk Q3) */
void orc_robustlog_activation_inplace(orc_mat *C, float min_prob) {
    if (!C) return;
    const float mpm1 = 1.0f - min_prob;
    for (size_t i = 0; i < C->stride * C->nc; i++) {
        const float t = mpm1 * C->data.f[i];
        C->data.f[i] = orc_logf(min_prob + t);
    }
}

/* scrappie_matrix.c:323-351 */
orc_mat *orc_affine_map(const orc_mat *X, const orc_mat *W, const orc_mat *b, orc_mat *C) {
    if (!X) return NULL;
    C = orc_remake_mat(C, W->nc, X->nc);
    if (!C) return NULL;
    for (size_t c = 0; c < C->nc; c++)
        memcpy(C->data.f + c * C->stride, b->data.f, C->stride * sizeof(float));
    sgemm_tn(W->nc, X->nc, W->nr, W->data.f, W->stride, X->data.f, X->stride,
             C->data.f, C->stride);
    return C;
}

/* scrappie_matrix.c:353-383: bias, then two sgemm's accumulated into C */
orc_mat *orc_affine_map2(const orc_mat *Xf, const orc_mat *Xb, const orc_mat *Wf,
                         const orc_mat *Wb, const orc_mat *b, orc_mat *C) {
    if (!Xf || !Xb) return NULL;
    C = orc_remake_mat(C, Wf->nc, Xf->nc);
    if (!C) return NULL;
    for (size_t c = 0; c < C->nc; c++)
        memcpy(C->data.f + c * C->stride, b->data.f, C->stride * sizeof(float));
    sgemm_tn(Wf->nc, Xf->nc, Wf->nr, Wf->data.f, Wf->stride, Xf->data.f, Xf->stride, C->data.f, C->stride);
    sgemm_tn(Wb->nc, Xb->nc, Wb->nr, Wb->data.f, Wb->stride, Xb->data.f, Xb->stride, C->data.f, C->stride);
    return C;
}

/* scrappie_matrix.c:385-407: lane-wise partial sums, pad lanes of the last
 * vector subtracted, two horizontal adds, multiply by reciprocal (pads too) */
void orc_row_normalise_inplace(orc_mat *C) {
    if (!C) return;
    const size_t npad = C->stride - C->nr;
    for (size_t col = 0; col < C->nc; col++) {
        float *v = C->data.f + col * C->stride;
        float s[4] = { v[0], v[1], v[2], v[3] };
        for (size_t row = 1; row < C->nrq; row++)
            for (int l = 0; l < 4; l++) s[l] += v[4 * row + l];
        const float *last = v + 4 * (C->nrq - 1);
        for (int l = 0; l < 4; l++) {
            /* mask lanes: lane3 if npad>=1, lane2 if npad>=2, lane1 if npad>=3 */
            const bool masked = (l >= 1) && ((size_t)(4 - l) <= npad);
            s[l] -= masked ? last[l] : 0.0f;
        }
        const float tsum = (s[0] + s[1]) + (s[2] + s[3]);
        const float recip = 1.0f / tsum;
        for (size_t i = 0; i < C->stride; i++) v[i] *= recip;
    }
}

/* scrappie_matrix.c:560-568: division, real rows only */
void orc_shift_scale_matrix_inplace(orc_mat *C, float shift, float scale) {
    if (!C) return;
    for (size_t c = 0; c < C->nc; c++)
        for (size_t r = 0; r < C->nr; r++) {
            float *p = C->data.f + c * C->stride + r;
            *p = (*p - shift) / scale;
        }
}

/* layers.c:303-319 */
void orc_residual_inplace(const orc_mat *X, orc_mat *fX) {
    if (!X || !fX) return;
    for (size_t i = 0; i < X->stride * X->nc; i++) fX->data.f[i] += X->data.f[i];
}

/* ------------------------------------------------------------------ */
/* G2 gru_step -- layers.c:472-527.  Gate order in xF: [z | r | hbar]. */
/* ------------------------------------------------------------------ */
void orc_gru_step(const orc_mat *x, const orc_mat *istate, const orc_mat *sW,
                  const orc_mat *sW2, orc_mat *xF, orc_mat *ostate) {
    const size_t size = istate->nr;
    memcpy(xF->data.f, x->data.f, x->stride * sizeof(float));          /* :501 */
    sgemv_t(sW->nr, sW->nc, sW->data.f, sW->stride, istate->data.f, xF->data.f); /* :505 */
    for (size_t i = 0; i < 2 * size; i++) xF->data.f[i] = orc_logisticf(xF->data.f[i]);
    const float *z = xF->data.f;
    float *r = xF->data.f + size;
    float *hbar = xF->data.f + 2 * size;
    for (size_t i = 0; i < size; i++) r[i] *= istate->data.f[i];       /* :515 */
    sgemv_t(sW2->nr, sW2->nc, sW2->data.f, sW2->stride, r, hbar);      /* :517 */
    for (size_t i = 0; i < size; i++) hbar[i] = orc_tanhf(hbar[i]);
    for (size_t i = 0; i < size; i++) {                                /* :525 */
        const float a = z[i] * istate->data.f[i];
        const float b = (1.0f - z[i]) * hbar[i];
        ostate->data.f[i] = a + b;
    }
}

static orc_mat colview(const orc_mat *M, size_t col) {
    orc_mat v = *M;
    v.nc = 1;
    v.data.f = M->data.f + col * M->stride;
    return v;
}

/* layers.c:373-420: zero state parked in output column 1 (quirk Q6) */
orc_mat *orc_gru_forward(const orc_mat *X, const orc_mat *sW, const orc_mat *sW2,
                         orc_mat *ostate) {
    if (!X) return NULL;
    const size_t bsize = X->nc, size = sW2->nc;
    if (bsize < 2) return NULL;
    ostate = orc_remake_mat(ostate, size, bsize);
    orc_mat *tmp = orc_make_mat(3 * size, 1);
    if (!ostate || !tmp) { orc_free_mat(tmp); return NULL; }
    memset(ostate->data.f + ostate->stride, 0, ostate->stride * sizeof(float));
    orc_mat xCol = colview(X, 0), s1 = colview(ostate, 1), s2 = colview(ostate, 0);
    orc_gru_step(&xCol, &s1, sW, sW2, tmp, &s2);
    for (size_t i = 1; i < bsize; i++) {
        xCol = colview(X, i); s1 = colview(ostate, i - 1); s2 = colview(ostate, i);
        orc_gru_step(&xCol, &s1, sW, sW2, tmp, &s2);
    }
    orc_free_mat(tmp);
    return ostate;
}

/* layers.c:422-470: zero state parked in output column 0 */
orc_mat *orc_gru_backward(const orc_mat *X, const orc_mat *sW, const orc_mat *sW2,
                          orc_mat *ostate) {
    if (!X) return NULL;
    const size_t bsize = X->nc, size = sW2->nc;
    if (bsize < 2) return NULL;
    ostate = orc_remake_mat(ostate, size, bsize);
    orc_mat *tmp = orc_make_mat(3 * size, 1);
    if (!ostate || !tmp) { orc_free_mat(tmp); return NULL; }
    memset(ostate->data.f, 0, ostate->stride * sizeof(float));
    orc_mat xCol = colview(X, bsize - 1), s1 = colview(ostate, 0),
            s2 = colview(ostate, bsize - 1);
    orc_gru_step(&xCol, &s1, sW, sW2, tmp, &s2);
    for (size_t i = 1; i < bsize; i++) {
        const size_t index = bsize - i - 1;
        xCol = colview(X, index); s1 = colview(ostate, index + 1); s2 = colview(ostate, index);
        orc_gru_step(&xCol, &s1, sW, sW2, tmp, &s2);
    }
    orc_free_mat(tmp);
    return ostate;
}

/* layers.c:340-357 (quirks Q2, Q5): input scaled in place by division, no
 * max subtraction, exp clamped, pads normalised too. */
orc_mat *orc_softmax_with_temperature(orc_mat *X, const orc_mat *W, const orc_mat *b,
                                      float tempW, float tempb, orc_mat *C) {
    if (!X) return NULL;
    orc_shift_scale_matrix_inplace(X, 0.0f, tempW / tempb);
    C = orc_affine_map(X, W, b, C);
    if (!C) return NULL;
    orc_shift_scale_matrix_inplace(C, 0.0f, tempb);
    orc_exp_activation_inplace(C);
    orc_row_normalise_inplace(C);
    return C;
}

/* layers.c:835-871 */
float orc_crf_partition_function(const orc_mat *C) {
    if (!C) return NAN;
    const size_t nstate = (size_t)roundf(sqrtf((float)C->nr));
    float *mem = calloc(2 * nstate, sizeof(float));
    if (!mem) return NAN;
    float *curr = mem, *prev = mem + nstate;
    for (size_t c = 0; c < C->nc; c++) {
        const size_t offset = c * C->stride;
        float *t = curr; curr = prev; prev = t;
        for (size_t st1 = 0; st1 < nstate; st1++) {
            const size_t offsetS = offset + st1 * nstate;
            curr[st1] = C->data.f[offsetS + 0] + prev[0];
            for (size_t st2 = 1; st2 < nstate; st2++)
                curr[st1] = orc_logsumexpf(curr[st1], C->data.f[offsetS + st2] + prev[st2]);
        }
    }
    float logZ = curr[0];
    for (size_t st = 1; st < nstate; st++) logZ = orc_logsumexpf(logZ, curr[st]);
    free(mem);
    return logZ;
}

/* layers.c:874-889 */
orc_mat *orc_globalnorm(const orc_mat *X, const orc_mat *W, const orc_mat *b, orc_mat *C) {
    C = orc_affine_map(X, W, b, C);
    if (!C) return NULL;
    const float logZ = orc_crf_partition_function(C) / (float)C->nc;
    for (size_t c = 0; c < C->nc; c++)
        for (size_t r = 0; r < C->nr; r++) C->data.f[c * C->stride + r] -= logZ;
    return C;
}

/* ------------------------------------------------------------------ */
/* N1 / N2 networks                                                    */
/* ------------------------------------------------------------------ */

/* shared trunk: F0 -> C1 -> act -> 5 x (L1 -> G1 alternating B,F,B,F,B)
 * networks.c:257-286 (rgrgr) and :575-611 (rnnrf, with residual_inplace of
 * the layer input onto the GRU output). */
orc_mat *orc_raw_trunk(const orc_model *m, orc_raw_table signal, int upto);
orc_mat *orc_trunk(const orc_model *m, orc_raw_table signal, int upto) {
    if (m->arch == ORC_ARCH_RAW) return orc_raw_trunk(m, signal, upto);
    if (signal.n == 0 || !signal.raw) return NULL;
    orc_mat *raw_mat = orc_features_from_raw(signal);
    orc_mat *act = orc_convolution(raw_mat, m->conv_W, m->conv_b, (size_t)m->stride, NULL);
    orc_free_mat(raw_mat);
    if (!act) return NULL;
    if (m->conv_act == ORC_ACT_TANH) orc_tanh_activation_inplace(act);
    else orc_elu_activation_inplace(act);
    for (int l = 0; l < 5 && l < upto; l++) {
        orc_mat *gin = orc_affine_map(act, m->gru_iW[l], m->gru_b[l], NULL);
        orc_mat *g = (l % 2 == 0) ? orc_gru_backward(gin, m->gru_sW[l], m->gru_sW2[l], NULL)
                                  : orc_gru_forward(gin, m->gru_sW[l], m->gru_sW2[l], NULL);
        orc_free_mat(gin);
        if (!g) { orc_free_mat(act); return NULL; }
        if (m->arch == ORC_ARCH_RNNRF) orc_residual_inplace(act, g);
        orc_free_mat(act);
        act = g;
    }
    return act;
}

/* networks.c:250-296 (r94), :299-345 (r941), :348-394 (r10: tanh after conv) */
orc_mat *orc_rgrgr_posterior(const orc_model *m, orc_raw_table signal, float min_prob,
                             float tempW, float tempb, bool return_log) {
    orc_mat *top = orc_trunk(m, signal, 5);
    if (!top) return NULL;
    orc_mat *post = orc_softmax_with_temperature(top, m->ff_W, m->ff_b, tempW, tempb, NULL);
    orc_free_mat(top);
    if (post && return_log) orc_robustlog_activation_inplace(post, min_prob);
    return post;
}

/* networks.c:567-615 */
orc_mat *orc_rnnrf_transitions(const orc_model *m, orc_raw_table signal) {
    orc_mat *top = orc_trunk(m, signal, 5);
    if (!top) return NULL;
    orc_mat *trans = orc_globalnorm(top, m->ff_W, m->ff_b, NULL);
    orc_free_mat(top);
    return trans;
}

/* N3 networks.c:196-247: conv -> tanh -> {two affine maps -> gru_forward,
 * gru_backward -> feedforward2_tanh (layers.c:359)} x 2 */
orc_mat *orc_raw_trunk(const orc_model *m, orc_raw_table signal, int upto) {
    if (signal.n == 0 || !signal.raw) return NULL;
    orc_mat *raw_mat = orc_features_from_raw(signal);
    orc_mat *act = orc_convolution(raw_mat, m->conv_W, m->conv_b, (size_t)m->stride, NULL);
    orc_free_mat(raw_mat);
    if (!act) return NULL;
    if (m->conv_act == ORC_ACT_TANH) orc_tanh_activation_inplace(act);
    else orc_elu_activation_inplace(act);
    for (int l = 0; l < 2 && l < upto; l++) {
        orc_mat *fin = orc_affine_map(act, m->gru_iW[2 * l], m->gru_b[2 * l], NULL);
        orc_mat *bin = orc_affine_map(act, m->gru_iW[2 * l + 1], m->gru_b[2 * l + 1], NULL);
        orc_free_mat(act);
        orc_mat *gf = orc_gru_forward(fin, m->gru_sW[2 * l], m->gru_sW2[2 * l], NULL);
        orc_mat *gb = orc_gru_backward(bin, m->gru_sW[2 * l + 1], m->gru_sW2[2 * l + 1], NULL);
        orc_free_mat(fin); orc_free_mat(bin);
        act = orc_affine_map2(gf, gb, l ? m->ff2_Wf : m->ff1_Wf, l ? m->ff2_Wb : m->ff1_Wb,
                              l ? m->ff2_b : m->ff1_b, NULL);
        orc_free_mat(gf); orc_free_mat(gb);
        if (!act) return NULL;
        orc_tanh_activation_inplace(act);
    }
    return act;
}

orc_mat *orc_raw_posterior(const orc_model *m, orc_raw_table signal, float min_prob,
                           float tempW, float tempb, bool return_log) {
    orc_mat *top = orc_raw_trunk(m, signal, 2);
    if (!top) return NULL;
    orc_mat *post = orc_softmax_with_temperature(top, m->ff_W, m->ff_b, tempW, tempb, NULL);
    orc_free_mat(top);
    if (post && return_log) orc_robustlog_activation_inplace(post, min_prob);
    return post;
}

orc_mat *orc_posterior(const orc_model *m, orc_raw_table signal, float min_prob,
                       float tempW, float tempb, bool return_log) {
    if (m->arch == ORC_ARCH_RNNRF) return orc_rnnrf_transitions(m, signal);
    if (m->arch == ORC_ARCH_RAW) return orc_raw_posterior(m, signal, min_prob, tempW, tempb, return_log);
    return orc_rgrgr_posterior(m, signal, min_prob, tempW, tempb, return_log);
}

/* ------------------------------------------------------------------ */
/* (f).4  events bi-LSTM                                                */
/* ------------------------------------------------------------------ */

/* nnfeatures.c:51-86: Kahan sums per feature row, variance, then the SSE
 * APPROXIMATE reciprocal square root (rsqrtps, ~12 bits) -- the reference's
 * result depends on that instruction, so the restatement uses it too. */
#include <xmmintrin.h>
void orc_studentise_features_kahan(orc_mat *features) {
    const int nevent = (int)features->nc;
    float sum[4] = {0, 0, 0, 0}, sumsq[4] = {0, 0, 0, 0}, comp[4] = {0, 0, 0, 0}, compsq[4] = {0, 0, 0, 0};
    for (int ev = 0; ev < nevent; ev++) {
        const float *f = features->data.f + (size_t)ev * features->stride;
        for (int k = 0; k < 4; k++) {
            const float d1 = f[k] - comp[k];
            const float sum_tmp = sum[k] + d1;
            comp[k] = (sum_tmp - sum[k]) - d1;
            sum[k] = sum_tmp;
            const float d2 = f[k] * f[k] - compsq[k];
            const float sumsq_tmp = sumsq[k] + d2;
            compsq[k] = (sumsq_tmp - sumsq[k]) - d2;
            sumsq[k] = sumsq_tmp;
        }
    }
    float scale[4], shift[4];
    for (int k = 0; k < 4; k++) {
        sum[k] /= (float)nevent;
        sumsq[k] /= (float)nevent;
        sumsq[k] -= sum[k] * sum[k];
    }
    _mm_storeu_ps(scale, _mm_rsqrt_ps(_mm_loadu_ps(sumsq)));
    for (int k = 0; k < 4; k++) shift[k] = sum[k] * scale[k];
    for (int ev = 0; ev < nevent; ev++) {
        float *f = features->data.f + (size_t)ev * features->stride;
        for (int k = 0; k < 4; k++) f[k] = scale[k] * f[k] - shift[k];
    }
}

/* nnfeatures.c:88-110: (mean, stdv, length, |mean - next mean|), last event's delta 0 */
orc_mat *orc_features_from_events(orc_event_table et, bool normalise) {
    if (!et.event) return NULL;
    const size_t nevent = et.end - et.start, offset = et.start;
    orc_mat *features = orc_make_mat(4, nevent);
    if (!features) return NULL;
    for (size_t ev = 0; ev + 1 < nevent; ev++) {
        float *f = features->data.f + ev * features->stride;
        f[0] = et.event[ev + offset].mean;
        f[1] = et.event[ev + offset].stdv;
        f[2] = et.event[ev + offset].length;
        f[3] = (float)fabs(et.event[ev + offset].mean - et.event[ev + offset + 1].mean);
    }
    float *f = features->data.f + (nevent - 1) * features->stride;
    f[0] = et.event[et.end - 1].mean;
    f[1] = et.event[et.end - 1].stdv;
    f[2] = et.event[et.end - 1].length;
    f[3] = 0.0f;
    if (normalise) orc_studentise_features_kahan(features);
    return features;
}

/* layers.c:119-147, types as in the reference: the window covers w1 = icol-wh+1 .. icol+wh
 * (four positions for w = 3; the fourth lands in the next column's first rows and is
 * overwritten when that column is built), and because `w1 <= icol + wh` compares an int
 * with a size_t, a negative w1 ends the loop at once: output column 0 stays zero (Q17). */
orc_mat *orc_window(const orc_mat *input, size_t w, size_t stride) {
    if (!input) return NULL;
    const size_t wh = (w + 1) / 2;
    orc_mat *output = orc_make_mat(input->nr * w, (size_t)ceilf(input->nc / (float)stride));
    if (!output) return NULL;
    for (size_t col = 0; col < output->nc; col++) {
        const size_t out_offset = col * output->stride;
        const int icol = (int)(col * stride);
        for (int i = 0, w1 = (icol - wh + 1); w1 <= icol + wh; w1++) {
            if (w1 < 0 || w1 >= input->nc) {
                i += input->nr;
                continue;
            }
            const size_t in_offset = w1 * input->stride;
            for (size_t row = 0; row < input->nr; row++, i++) {
                /* the reference writes past the column (and, for the last columns, would write
                 * past the matrix were those positions not skipped); stay inside the buffer */
                if (out_offset + i < output->nc * output->stride) output->data.f[out_offset + i] = input->data.f[in_offset + row];
            }
        }
    }
    return output;
}

/* layers.c:772-832: xF = [input | update | forget | output] pre-activations, peep = [update | forget | output] */
void orc_lstm_step(const orc_mat *xAffine, const orc_mat *out_prev, const orc_mat *sW, const orc_mat *peep,
                   orc_mat *xF, orc_mat *state, orc_mat *output) {
    const size_t size = state->nr;
    memcpy(xF->data.f, xAffine->data.f, xAffine->stride * sizeof(float));                     /* :798 */
    sgemv_t(sW->nr, sW->nc, sW->data.f, sW->stride, out_prev->data.f, xF->data.f);           /* :800 */
    for (size_t i = 0; i < size; i++) {
        const float st = state->data.f[i];
        const float forget = orc_logisticf(xF->data.f[2 * size + i] + st * peep->data.f[size + i]) * st;
        const float update = orc_logisticf(xF->data.f[size + i] + st * peep->data.f[i]) * orc_tanhf(xF->data.f[i]);
        const float ns = forget + update;
        state->data.f[i] = ns;
        output->data.f[i] = orc_logisticf(xF->data.f[3 * size + i] + ns * peep->data.f[2 * size + i]) * orc_tanhf(ns);
    }
}

/* layers.c:673-721: zero output parked in column 1, state zero */
orc_mat *orc_lstm_forward(const orc_mat *Xaffine, const orc_mat *sW, const orc_mat *p, orc_mat *output) {
    if (!Xaffine) return NULL;
    const size_t size = sW->nr, bsize = Xaffine->nc;
    if (bsize < 2) return NULL;
    output = orc_remake_mat(output, size, bsize);
    orc_mat *tmp = orc_make_mat(4 * size, 1), *state = orc_make_mat(size, 1);
    if (!output || !tmp || !state) { orc_free_mat(tmp); orc_free_mat(state); return NULL; }
    memset(output->data.f + output->stride, 0, output->stride * sizeof(float));
    orc_mat xCol = colview(Xaffine, 0), s1 = colview(output, 1), s2 = colview(output, 0);
    orc_lstm_step(&xCol, &s1, sW, p, tmp, state, &s2);
    for (size_t i = 1; i < bsize; i++) {
        xCol = colview(Xaffine, i); s1 = colview(output, i - 1); s2 = colview(output, i);
        orc_lstm_step(&xCol, &s1, sW, p, tmp, state, &s2);
    }
    orc_free_mat(state); orc_free_mat(tmp);
    return output;
}

/* layers.c:723-770 */
orc_mat *orc_lstm_backward(const orc_mat *Xaffine, const orc_mat *sW, const orc_mat *p, orc_mat *output) {
    if (!Xaffine) return NULL;
    const size_t size = sW->nr, bsize = Xaffine->nc;
    if (bsize < 2) return NULL;
    output = orc_remake_mat(output, size, bsize);
    orc_mat *tmp = orc_make_mat(4 * size, 1), *state = orc_make_mat(size, 1);
    if (!output || !tmp || !state) { orc_free_mat(tmp); orc_free_mat(state); return NULL; }
    memset(output->data.f, 0, output->stride * sizeof(float));
    orc_mat xCol = colview(Xaffine, bsize - 1), s1 = colview(output, 0), s2 = colview(output, bsize - 1);
    orc_lstm_step(&xCol, &s1, sW, p, tmp, state, &s2);
    for (size_t i = 1; i < bsize; i++) {
        const size_t index = bsize - i - 1;
        xCol = colview(Xaffine, index); s1 = colview(output, index + 1); s2 = colview(output, index);
        orc_lstm_step(&xCol, &s1, sW, p, tmp, state, &s2);
    }
    orc_free_mat(state); orc_free_mat(tmp);
    return output;
}

/* networks.c:155-181 from the windowed features on */
orc_mat *orc_events_trunk(const orc_model *m, const orc_mat *feature3, int upto) {
    if (!feature3) return NULL;
    orc_mat *act = orc_make_mat(feature3->nr, feature3->nc);
    if (!act) return NULL;
    memcpy(act->data.f, feature3->data.f, feature3->nc * feature3->stride * sizeof(float));
    for (int l = 0; l < 2 && l < upto; l++) {
        orc_mat *fin = orc_affine_map(act, m->gru_iW[2 * l], m->gru_b[2 * l], NULL);
        orc_mat *bin = orc_affine_map(act, m->gru_iW[2 * l + 1], m->gru_b[2 * l + 1], NULL);
        orc_free_mat(act);
        orc_mat *lf = orc_lstm_forward(fin, m->gru_sW[2 * l], m->lstm_p[2 * l], NULL);
        orc_mat *lb = orc_lstm_backward(bin, m->gru_sW[2 * l + 1], m->lstm_p[2 * l + 1], NULL);
        orc_free_mat(fin); orc_free_mat(bin);
        act = orc_affine_map2(lf, lb, l ? m->ff2_Wf : m->ff1_Wf, l ? m->ff2_Wb : m->ff1_Wb,
                              l ? m->ff2_b : m->ff1_b, NULL);
        orc_free_mat(lf); orc_free_mat(lb);
        if (!act) return NULL;
        orc_tanh_activation_inplace(act);
    }
    return act;
}

orc_mat *orc_events_posterior_from_features(const orc_model *m, const orc_mat *feature3, float min_prob,
                                            float tempW, float tempb, bool return_log) {
    orc_mat *top = orc_events_trunk(m, feature3, 2);
    if (!top) return NULL;
    orc_mat *post = orc_softmax_with_temperature(top, m->ff_W, m->ff_b, tempW, tempb, NULL);
    orc_free_mat(top);
    if (post && return_log) orc_robustlog_activation_inplace(post, min_prob);
    return post;
}

/* networks.c:146-193 */
orc_mat *orc_events_posterior(const orc_model *m, orc_event_table et, float min_prob, float tempW, float tempb,
                              bool return_log) {
    if (et.n == 0 || !et.event) return NULL;
    orc_mat *features = orc_features_from_events(et, true);
    orc_mat *feature3 = orc_window(features, 3, 1);
    orc_free_mat(features);
    orc_mat *post = orc_events_posterior_from_features(m, feature3, min_prob, tempW, tempb, return_log);
    orc_free_mat(feature3);
    return post;
}

/* ------------------------------------------------------------------ */
/* D1 transducer Viterbi                                               */
/* ------------------------------------------------------------------ */

/* util.c:9-23: first maximum wins */
int orc_argmaxf(const float *x, size_t n) {
    if (!x) return -1;
    size_t imax = 0;
    float vmax = x[0];
    for (size_t i = 1; i < n; i++)
        if (x[i] > vmax) { vmax = x[i]; imax = i; }
    return (int)imax;
}

/* decode.c:58-98 */
static float local_backtrace(const float *score, size_t n, const orc_imat *tb, int *seq) {
    const size_t nblock = tb->nc;
    for (size_t i = 0; i <= nblock; i++) seq[i] = -1;
    int last_state = orc_argmaxf(score, n + 2);
    const float logscore = score[last_state];
    for (size_t i = 0; i < nblock; i++) {
        const size_t ri = nblock - i - 1;
        const int state = tb->data.f[ri * tb->stride + last_state];
        if (state >= 0) {
            seq[ri + 1] = last_state;
            last_state = state;
        }
    }
    seq[0] = last_state;
    for (size_t i = 0; i < nblock; i++) {          /* start -> stay */
        if (seq[i] == (int)n) seq[i] = -1; else break;
    }
    for (int i = (int)nblock; i >= 0; i--) {       /* end -> stay */
        if (seq[i] == (int)n + 1) seq[i] = -1; else break;
    }
    return logscore;
}

/* Suffix maximum with the reference's tie rule (decode.c:186-210, :228-251,
 * :276-302): candidate r=0 first, a later r replaces only if strictly greater.
 * nsuf = number of suffixes, nr = number of prefixes; candidate index written
 * is r*nsuf + j (the previous state itself). */
static void suffix_max(const float *prev, size_t nsuf, size_t nr, float *val, int *idx) {
    for (size_t j = 0; j < nsuf; j++) { val[j] = prev[j]; idx[j] = 0; }
    for (size_t r = 1; r < nr; r++)
        for (size_t j = 0; j < nsuf; j++) {
            const float cand = prev[r * nsuf + j];
            if (val[j] < cand) { val[j] = cand; idx[j] = (int)r; }
        }
    for (size_t j = 0; j < nsuf; j++) idx[j] = idx[j] * (int)nsuf + (int)j;
}

/* decode.c:123-365.  Moves applied in the order stay, step, skip, (slip),
 * leave-start; every comparison strict so the earlier move wins ties (Q7). */
float orc_decode_transducer(const orc_mat *logpost, float stay_pen, float skip_pen,
                            float local_pen, int *seq, bool allow_slip) {
    if (!logpost || !seq) return NAN;
    const size_t nblock = logpost->nc;
    const size_t nh = logpost->nr - 1;
    if (nh % 64 != 0 || (allow_slip && nh % 256 != 0)) return NAN;
    const size_t n4 = nh / 4, n16 = nh / 16, n64 = nh / 64;
    float logscore = NAN;
    float *score = malloc((nh + 2) * sizeof(float));
    float *prev = malloc((nh + 2) * sizeof(float));
    float *tmp = malloc(n4 * sizeof(float));
    int *itmp = malloc(n4 * sizeof(int));
    orc_imat *tb = orc_make_imat(nh + 2, nblock);
    if (!score || !prev || !tmp || !itmp || !tb) goto cleanup;

    for (size_t i = 0; i < nh; i++) score[i] = -BIG_FLOAT;      /* :155-159 */
    score[nh] = 0.0f;
    score[nh + 1] = -BIG_FLOAT;

    for (size_t blk = 0; blk < nblock; blk++) {
        const float *post = logpost->data.f + blk * logpost->stride;
        int32_t *t = tb->data.f + blk * tb->stride;
        { float *s = score; score = prev; prev = s; }

        /* stay :175-182 */
        const float stay = post[nh] - stay_pen;
        for (size_t i = 0; i < nh; i++) { score[i] = prev[i] + stay; t[i] = -1; }

        /* step :186-224 */
        suffix_max(prev, n4, 4, tmp, itmp);
        for (size_t pref = 0; pref < n4; pref++)
            for (size_t e = 0; e < 4; e++) {
                const size_t s = 4 * pref + e;
                const float step_score = post[s] + tmp[pref];
                if (score[s] < step_score) { score[s] = step_score; t[s] = itmp[pref]; }
            }

        /* skip :227-270 */
        suffix_max(prev, n16, 16, tmp, itmp);
        for (size_t pref = 0; pref < n16; pref++)
            for (size_t e = 0; e < 16; e++) {
                const size_t s = 16 * pref + e;
                const float skip_score = (post[s] + tmp[pref]) - skip_pen;
                if (score[s] < skip_score) { score[s] = skip_score; t[s] = itmp[pref]; }
            }

        /* slip :273-323; penalty is (float)(2.0 * skip_pen) */
        if (allow_slip) {
            const float slip_pen = (float)(2.0 * skip_pen);
            suffix_max(prev, n64, 64, tmp, itmp);
            for (size_t pref = 0; pref < n64; pref++)
                for (size_t e = 0; e < 64; e++) {
                    const size_t s = 64 * pref + e;
                    const float slip_score = (post[s] + tmp[pref]) - slip_pen;
                    if (score[s] < slip_score) { score[s] = slip_score; t[s] = itmp[pref]; }
                }
        }

        /* remain in / leave start :326-336 */
        score[nh] = prev[nh] + fmaxf(-local_pen, post[nh] - stay_pen);
        t[nh] = (int32_t)nh;
        for (size_t hst = 0; hst < nh; hst++) {
            const float sc = prev[nh] + post[hst];
            if (sc > score[hst]) { score[hst] = sc; t[hst] = (int32_t)nh; }
        }

        /* remain in / enter end :339-349 */
        score[nh + 1] = (float)(prev[nh + 1] + fmax(-local_pen, post[nh] - stay_pen));
        t[nh + 1] = (int32_t)(nh + 1);
        for (size_t hst = 0; hst < nh; hst++) {
            const float sc = prev[hst] - local_pen;
            if (sc > score[nh + 1]) { score[nh + 1] = sc; t[nh + 1] = (int32_t)hst; }
        }
    }
    logscore = local_backtrace(score, nh, tb, seq);

cleanup:
    orc_free_imat(tb);
    free(itmp); free(tmp); free(prev); free(score);
    return logscore;
}

/* decode.c:725-834 -- the reference's scalar twin, used by its own unit test
 * (test_scrappie_decoding.c:33-67) as the cross-check for decode_transducer. */
float orc_sloika_viterbi(const orc_mat *logpost, float stay_pen, float skip_pen,
                         float local_pen, int *seq) {
    if (!logpost || !seq) return NAN;
    const size_t nblock = logpost->nc, nhst = logpost->nr - 1;
    const size_t nstep = 4, nskip = 16;
    const size_t step_rem = nhst / nstep, skip_rem = nhst / nskip;
    float logscore = NAN;
    float *cscore = calloc(nhst + 2, sizeof(float));
    float *pscore = calloc(nhst + 2, sizeof(float));
    int *step_idx = calloc(step_rem, sizeof(int));
    int *skip_idx = calloc(skip_rem, sizeof(int));
    orc_imat *tb = orc_make_imat(nhst + 2, nblock);
    if (cscore && pscore && step_idx && skip_idx && tb) {
        for (size_t i = 0; i < nhst + 2; i++) cscore[i] = -BIG_FLOAT;
        cscore[nhst] = 0.0f;
        for (size_t i = 0; i < nblock; i++) {
            const float *lp = logpost->data.f + i * logpost->stride;
            int32_t *t = tb->data.f + i * tb->stride;
            { float *s = pscore; pscore = cscore; cscore = s; }
            /* colmaxf (decode.c:694-723): argmax over rows of a [nr x nc]
             * row-major view, first max wins */
            for (size_t c = 0; c < step_rem; c++) {
                int im = 0; float vm = pscore[c];
                for (size_t r = 1; r < nstep; r++)
                    if (pscore[r * step_rem + c] > vm) { vm = pscore[r * step_rem + c]; im = (int)r; }
                step_idx[c] = im;
            }
            for (size_t c = 0; c < skip_rem; c++) {
                int im = 0; float vm = pscore[c];
                for (size_t r = 1; r < nskip; r++)
                    if (pscore[r * skip_rem + c] > vm) { vm = pscore[r * skip_rem + c]; im = (int)r; }
                skip_idx[c] = im;
            }
            for (size_t hst = 0; hst < nhst; hst++) {
                const size_t sp = hst / nstep, kp = hst / nskip;
                const size_t step_hst = sp + (size_t)step_idx[sp] * step_rem;
                const size_t skip_hst = kp + (size_t)skip_idx[kp] * skip_rem;
                const float step_score = pscore[step_hst];
                const float skip_score = pscore[skip_hst] - skip_pen;
                if (step_score > skip_score) { cscore[hst] = step_score; t[hst] = (int32_t)step_hst; }
                else { cscore[hst] = skip_score; t[hst] = (int32_t)skip_hst; }
                cscore[hst] += lp[hst];
            }
            for (size_t hst = 0; hst < nhst; hst++) {
                const float sc = pscore[hst] + lp[nhst] - stay_pen;
                if (sc > cscore[hst]) { cscore[hst] = sc; t[hst] = -1; }
            }
            cscore[nhst] = pscore[nhst] + fmaxf(-local_pen, lp[nhst] - stay_pen);
            t[nhst] = (int32_t)nhst;
            for (size_t hst = 0; hst < nhst; hst++) {
                const float sc = pscore[nhst] + lp[hst];
                if (sc > cscore[hst]) { cscore[hst] = sc; t[hst] = (int32_t)nhst; }
            }
            cscore[nhst + 1] = pscore[nhst + 1] + fmaxf(-local_pen, lp[nhst] - stay_pen);
            t[nhst + 1] = (int32_t)(nhst + 1);
            for (size_t hst = 0; hst < nhst; hst++) {
                const float sc = pscore[hst] - local_pen;
                if (sc > cscore[nhst + 1]) { cscore[nhst + 1] = sc; t[nhst + 1] = (int32_t)hst; }
            }
        }
        logscore = local_backtrace(cscore, nhst, tb, seq);
    }
    orc_free_imat(tb);
    free(skip_idx); free(step_idx); free(pscore); free(cscore);
    return logscore;
}

/* ------------------------------------------------------------------ */
/* D3 stitching                                                        */
/* ------------------------------------------------------------------ */

/* decode.c:367-382 */
int orc_overlap(int k1, int k2, int nkmer) {
    int kmer_mask = nkmer - 1, ov = 0;
    do {
        kmer_mask >>= 2;
        k1 &= kmer_mask;
        k2 >>= 2;
        ov += 1;
    } while (k1 != k2);
    return ov;
}

static const char base_lookup[4] = { 'A', 'C', 'G', 'T' };   /* decode.c:410 */

/* decode.c:449-509 */
char *orc_overlapper(const int *seq, size_t n, int nkmer, int *pos) {
    if (!seq) return NULL;
    size_t hb = 0;                                 /* position_highest_bit :384 */
    for (size_t x = (size_t)nkmer; x != 0; hb++, x >>= 1) ;
    const size_t kmer_len = hb / 2;
    size_t length = kmer_len;
    size_t st = 0;
    while (st < n && seq[st] < 0) st++;            /* first_nonnegative :390 */
    if (st == n) return NULL;
    int kprev = seq[st];
    for (size_t k = st + 1; k < n; k++) {
        if (seq[k] < 0) continue;
        length += (size_t)orc_overlap(kprev, seq[k], nkmer);
        kprev = seq[k];
    }
    char *bases = calloc(length + 1, sizeof(char));
    if (!bases) return NULL;
    {
        size_t kmer = (size_t)seq[st];
        for (size_t k = 1; k <= kmer_len; k++) {
            bases[kmer_len - k] = base_lookup[kmer & 3];
            kmer >>= 2;
        }
    }
    if (pos) pos[0] = 0;
    size_t last_idx = kmer_len - 1;
    kprev = seq[st];
    for (size_t k = st + 1; k < n; k++) {
        if (seq[k] < 0) {
            if (pos) pos[k] = pos[k - 1];
            continue;
        }
        const int ol = orc_overlap(kprev, seq[k], nkmer);
        if (pos) pos[k] = pos[k - 1] + ol;
        kprev = seq[k];
        size_t kmer = (size_t)seq[k];
        for (int i = 0; i < ol; i++) {
            bases[last_idx + (size_t)ol - (size_t)i] = base_lookup[kmer & 3];
            kmer >>= 2;
        }
        last_idx += (size_t)ol;
    }
    return bases;
}

/* ------------------------------------------------------------------ */
/* D4 / D5 CRF                                                         */
/* ------------------------------------------------------------------ */

/* decode.c:836-893 */
float orc_decode_crf(const orc_mat *trans, int *path) {
    if (!trans || !path) return NAN;
    const size_t nblk = trans->nc;
    const size_t nstate = (size_t)roundf(sqrtf((float)trans->nr));
    float *mem = calloc(2 * nstate, sizeof(float));
    orc_imat *tb = orc_make_imat(nstate, nblk);
    if (!mem || !tb) { orc_free_imat(tb); free(mem); return NAN; }
    float *curr = mem, *prev = mem + nstate;
    for (size_t blk = 0; blk < nblk; blk++) {
        const float *tr = trans->data.f + blk * trans->stride;
        int32_t *t = tb->data.f + blk * tb->stride;
        { float *s = curr; curr = prev; prev = s; }
        for (size_t st1 = 0; st1 < nstate; st1++) {
            curr[st1] = tr[st1 * nstate + 0] + prev[0];
            t[st1] = 0;
            for (size_t st2 = 1; st2 < nstate; st2++) {
                const float sc = tr[st1 * nstate + st2] + prev[st2];
                if (sc > curr[st1]) { curr[st1] = sc; t[st1] = (int32_t)st2; }
            }
        }
    }
    float score = curr[0];
    for (size_t i = 1; i < nstate; i++) if (curr[i] > score) score = curr[i];
    path[nblk] = orc_argmaxf(curr, nstate);
    for (size_t blk = nblk; blk > 0; blk--)
        path[blk - 1] = tb->data.f[(blk - 1) * tb->stride + (size_t)path[blk]];
    orc_free_imat(tb);
    free(mem);
    return score;
}

/* decode.c:895-918; `pos` is never written (quirk Q11) */
char *orc_crfpath_to_basecall(const int *path, size_t npos, int *pos) {
    if (!path || !pos) return NULL;
    size_t nbase = 0;
    for (size_t i = 0; i < npos; i++) if (path[i] < 4) nbase++;
    char *bc = calloc(nbase + 1, sizeof(char));
    if (!bc) return NULL;
    for (size_t i = 0, b = 0; i < npos; i++)
        if (path[i] < 4) bc[b++] = base_lookup[path[i]];
    return bc;
}

/* decode.c:928-1012 */
orc_mat *orc_posterior_crf(const orc_mat *trans) {
    if (!trans) return NULL;
    const size_t nstate = (size_t)roundf(sqrtf((float)trans->nr));
    const size_t nblk = trans->nc;
    orc_mat *post = orc_make_mat(nstate, nblk + 1);
    if (!post) return NULL;
    for (size_t blk = 0; blk < nblk; blk++) {
        const float *tr = trans->data.f + blk * trans->stride;
        const float *prev = post->data.f + blk * post->stride;
        float *curr = post->data.f + (blk + 1) * post->stride;
        for (size_t st1 = 0; st1 < nstate; st1++) {
            curr[st1] = tr[st1 * nstate + 0] + prev[0];
            for (size_t st2 = 1; st2 < nstate; st2++)
                curr[st1] = orc_logsumexpf(curr[st1], tr[st1 * nstate + st2] + prev[st2]);
        }
    }
    float *tmpmem = malloc(2 * nstate * sizeof(float));
    if (!tmpmem) return orc_free_mat(post);
    float *prev = tmpmem, *curr = tmpmem + nstate;
    for (size_t st = 0; st < nstate; st++) curr[st] = 0.0f;
    float *lastp = post->data.f + nblk * post->stride;
    float tot = 0.0f;   /* NB: starts from 0.0, not -inf, as the reference does */
    for (size_t st = 0; st < nstate; st++) tot = orc_logsumexpf(tot, lastp[st]);
    for (size_t st = 0; st < nstate; st++) lastp[st] = expf(lastp[st] - tot);
    for (size_t blk = nblk; blk > 0; blk--) {
        const size_t blkm1 = blk - 1;
        const float *tr = trans->data.f + blkm1 * trans->stride;
        float *pp = post->data.f + blkm1 * post->stride;
        { float *s = curr; curr = prev; prev = s; }
        for (size_t st = 0; st < nstate; st++) curr[st] = tr[st] + prev[0];
        for (size_t st1 = 1; st1 < nstate; st1++)
            for (size_t st2 = 0; st2 < nstate; st2++)
                curr[st2] = orc_logsumexpf(curr[st2], tr[st1 * nstate + st2] + prev[st1]);
        float t2 = 0.0f;
        for (size_t st = 0; st < nstate; st++) {
            pp[st] += curr[st];
            t2 = orc_logsumexpf(t2, pp[st]);
        }
        for (size_t st = 0; st < nstate; st++) pp[st] = expf(pp[st] - t2);
    }
    free(tmpmem);
    return post;
}

/* ------------------------------------------------------------------ */
/* D2 homopolymer correction                                           */
/* ------------------------------------------------------------------ */

/* scrappie_seq_helpers.c:115-120 */
int orc_repeatblock(int b, int nrep) {
    int y = 0;
    for (int n = 0; n < nrep; n++) y = y * 4 + b;
    return y;
}

/* scrappie_seq_helpers.c:132-134 */
int orc_kmerlength_fromnblocks(int n) {
    return (int)(logf((float)n) / logf(4.0f));
}

/* homopolymer.c:67-157.  Returns run count, -1 on failure.  The reference
 * writes at most pathlength/2 runs into its arrays without a bound check; the
 * same capacity is kept and runs beyond it are dropped (the reference would
 * have corrupted the heap). */
static int find_runs(const int *path, int *runstarts, int *runlengths, int *runbases,
                     int cap, int pathlength, int kmerlength) {
    const int fkm1 = 1 << (2 * (kmerlength - 1));
    const int fkm2 = 1 << (2 * (kmerlength - 2));
    int runcount = 0;
    for (int base = 0; base < 4; base++) {
        const int repeatk = orc_repeatblock(base, kmerlength);
        const int repeatkm1 = orc_repeatblock(base, kmerlength - 1);
        const int repeatkm2 = orc_repeatblock(base, kmerlength - 2);
        for (int i = 1; i < pathlength - 2; i++) {
            const int p = path[i - 1], q = path[i];
            if ((p % fkm1 == repeatkm1) && (p != repeatk) && (p != -1)
                && ((-1 == q) || (q == repeatk))) {                 /* :103-113 */
                int e = i + 1;
                while (e < pathlength && (-1 == path[e] || path[e] == repeatk)) e++;
                if (runcount < cap) {
                    runstarts[runcount] = i; runlengths[runcount] = e - i; runbases[runcount] = base;
                    runcount++;
                }
            }
            if ((p % fkm2 == repeatkm2) && (p % fkm1 != repeatkm1) && (-1 != p)
                && ((-1 == q) || (q == repeatk))) {                 /* :116-136 */
                int j = i;
                while (j < pathlength && -1 == path[j]) j++;
                if (path[j] == repeatk && j < pathlength - 1) {
                    int e = j + 1;
                    while (e < pathlength && (path[e] == -1 || path[e] == repeatk)) e++;
                    if (runcount < cap) {
                        runstarts[runcount] = j; runlengths[runcount] = e - j; runbases[runcount] = base;
                        runcount++;
                    }
                }
            }
        }
    }
    return runcount;
}

/* homopolymer.c:175-235.  pathlength is post->nc (= T, not T+1: quirk Q9);
 * post column (i-1) pairs with path[i] (quirk Q8); libm expf, double mean. */
int orc_homopolymer_path(const orc_mat *post, int *viterbipath, int mean_flag) {
    if (!mean_flag) return 0;
    if (!post || !viterbipath) return -1;
    const int nsamples = (int)post->nc;
    const int staystate = (int)(post->nr - 1);
    const int kmerlength = orc_kmerlength_fromnblocks((int)post->nr);
    const int cap = nsamples / 2;
    int *rs = calloc((size_t)(cap ? cap : 1), sizeof(int));
    int *rl = calloc((size_t)(cap ? cap : 1), sizeof(int));
    int *rb = calloc((size_t)(cap ? cap : 1), sizeof(int));
    if (!rs || !rl || !rb) { free(rs); free(rl); free(rb); return -1; }
    const int runcount = find_runs(viterbipath, rs, rl, rb, cap, nsamples, kmerlength);
    for (int nrun = 0; nrun < runcount; nrun++) {
        int nviterbi = 0;
        double nmean = 0.0;
        const int runstate = orc_repeatblock(rb[nrun], kmerlength);
        const int ambigfrom = rs[nrun];
        const int ambigto = ambigfrom + rl[nrun] - 1;
        for (int i = ambigfrom; i <= ambigto; i++) {
            const double psu = expf(post->data.f[(size_t)(i - 1) * post->stride + (size_t)staystate]);
            const double pru = expf(post->data.f[(size_t)(i - 1) * post->stride + (size_t)runstate]);
            const double pr = pru / (pru + psu);
            nmean = nmean + pr;
            if (viterbipath[i] == runstate) nviterbi++;
        }
        const int newn = (int)(nmean + 0.5);
        if (newn != nviterbi)
            for (int i = 0; i <= ambigto - ambigfrom; i++)
                viterbipath[i + ambigfrom] = (i < newn) ? runstate : -1;
    }
    free(rs); free(rl); free(rb);
    return 0;
}

/* ------------------------------------------------------------------ */
/* whole-read driver, as calculate_post() scrappie_raw.c:265-315       */
/* ------------------------------------------------------------------ */
orc_params orc_default_params(void) {   /* scrappie_raw.c:98-121 */
    orc_params p = { .min_prob = 1e-5f, .tempW = 1.0f, .tempb = 1.0f,
                     .stay_pen = 0.0f, .skip_pen = 0.0f, .local_pen = 2.0f,
                     .use_slip = 0, .homopolymer_mean = 1,
                     .trim_start = 200, .trim_end = 10, .varseg_chunk = 100,
                     .varseg_thresh = 0.0f, .do_trim = 1 };
    return p;
}

int orc_basecall_raw(const orc_model *m, const float *raw, size_t n,
                     const orc_params *p, orc_call *out) {
    memset(out, 0, sizeof(*out));
    float *buf = malloc((n ? n : 1) * sizeof(float));
    if (!buf) return -1;
    memcpy(buf, raw, n * sizeof(float));
    orc_raw_table rt = { NULL, n, 0, n, buf };
    if (p->do_trim) {
        rt = orc_trim_and_segment_raw(rt, (size_t)p->trim_start, (size_t)p->trim_end,
                                      (size_t)p->varseg_chunk, p->varseg_thresh);
        if (!rt.raw) { free(buf); return 1; }
        orc_medmad_normalise_array(rt.raw + rt.start, rt.end - rt.start);
    }
    orc_mat *post = orc_posterior(m, rt, p->min_prob, p->tempW, p->tempb, true);
    if (!post) { free(buf); return 2; }
    const size_t nblock = post->nc;
    int *path = calloc(nblock + 1, sizeof(int));
    int *pos = calloc(nblock + 1, sizeof(int));
    char *basecall = NULL;
    float score;
    if (m->arch != ORC_ARCH_RNNRF) {
        score = orc_decode_transducer(post, p->stay_pen, p->skip_pen, p->local_pen,
                                      path, p->use_slip != 0);
        orc_homopolymer_path(post, path, p->homopolymer_mean);
        basecall = orc_overlapper(path, nblock + 1, (int)post->nr - 1, pos);
    } else {
        score = orc_decode_crf(post, path);
        basecall = orc_crfpath_to_basecall(path, nblock, pos);
    }
    free(path);
    orc_free_mat(post);
    out->score = score;
    out->nblock = nblock;
    out->start = rt.start;
    out->end = rt.end;
    out->basecall = basecall;
    out->pos = pos;
    free(buf);
    if (!basecall) { free(pos); out->pos = NULL; return 3; }
    return 0;
}

/* ------------------------------------------------------------------ */
/* many reads, as the reference's driver loops run them:                */
/* `#pragma omp parallel for schedule(dynamic)` over reads              */
/* (scrappie_raw.c:355,387), BLAS single-threaded (README.md:68-71).    */
/* Used by bench.py's cpu_baseline leg only: the reads are walked        */
/* round-robin for `budget_s` seconds of wall time (a thread looks at    */
/* the clock before it takes the next read), so the sample is bounded.   */
/* Without -fopenmp this is the same loop on one thread.                 */
/* ------------------------------------------------------------------ */
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif
static double orc_now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

long orc_basecall_many(const orc_model *m, const float *const *raws, const size_t *ns, size_t nreads,
                       const orc_params *p, int nthreads, double budget_s, size_t min_reads,
                       double *samples, double *bases, double *elapsed) {
    long done = 0, cursor = 0;
    double nsamp = 0.0, nbase = 0.0;
    const double t0 = orc_now();
    if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
    {
        long my_done = 0;
        double my_samp = 0.0, my_base = 0.0;
        for (;;) {
            long i;
#ifdef _OPENMP
#pragma omp atomic capture
#endif
            i = cursor++;
            if ((size_t)i >= min_reads && orc_now() - t0 >= budget_s) break;
            const size_t r = (size_t)i % nreads;
            orc_call c;
            if (0 == orc_basecall_raw(m, raws[r], ns[r], p, &c)) {
                my_base += (double)strlen(c.basecall);
                free(c.basecall); free(c.pos);
            }
            my_samp += (double)ns[r];
            my_done++;
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        { done += my_done; nsamp += my_samp; nbase += my_base; }
    }
    if (samples) *samples = nsamp;
    if (bases) *bases = nbase;
    if (elapsed) *elapsed = orc_now() - t0;
    return done;
}
