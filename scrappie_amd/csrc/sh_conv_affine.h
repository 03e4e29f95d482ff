/* sh_conv_affine.h -- part of sh_kernels.h (included from there, in this order): convolution + activation, affine maps (projection, joining layer).
 * Device code for gfx950 only; see sh_kernels.h for conventions (layouts, split products, citations). */
#ifndef SH_CONV_AFFINE_H
#define SH_CONV_AFFINE_H

/* ------------------------------------------------------------------ */
/* C1 + A1: strided convolution + ELU/tanh  (layers.c:159-246, :60, :15) */
/* One thread per (block t, 4 filters, read).  The reference builds the  */
/* result from edge sgemv's and strided sgemm's; which windows exist at  */
/* the right edge follows its index arithmetic exactly (quirk Q1).       */
/* ------------------------------------------------------------------ */
struct ShConvGeom {
    int WL, st, F, padL, padR, c0, shiftX, nstepC, nstepX;
};

__device__ __forceinline__ bool conv_main_included(const ShConvGeom &g, int N, int t) {
    /* layers.c:209-224: column c0+i+k*nstepC exists iff k < (N-shiftX-i*st)/nstepX */
    const int i = (t - g.c0) % g.nstepC, k = (t - g.c0) / g.nstepC;
    const int avail = N - g.shiftX - i * g.st;
    return avail > 0 && k < avail / g.nstepX;
}

template <int ACT>   /* 0 elu, 1 tanh */
__device__ __forceinline__ void conv_act_body(const float *__restrict__ sig, const ShMeta &md,
                                              const float *__restrict__ W /*[WL][F]*/,
                                              const float *__restrict__ bias, const ShConvGeom &g,
                                              float *__restrict__ out, int tchunk, unsigned *__restrict__ bad /*[npad]*/) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    bool out_of_range = false;
    float *sW = smem;                 /* WL*F */
    float *sB = smem + g.WL * g.F;    /* F */
    float *sX = sB + g.F;             /* 16 reads x span samples of this block's windows */
    const int span = (tchunk - 1) * g.st + g.WL;
    const int tile = blockIdx.x;
    const int Tt = md.tile_T[tile];
    if ((int)blockIdx.y * tchunk >= Tt) return;
    for (int i = threadIdx.x; i < g.WL * g.F; i += 256) sW[i] = W[i];
    for (int i = threadIdx.x; i < g.F; i += 256) sB[i] = bias[i];
    const int nchunk = g.F / 16;
    const int l = threadIdx.x & 63, b = l & 15, q = l >> 4;
    const int rd = tile * 16 + b;
    const int N = md.rN[rd], T = md.rT[rd];
    /* where this read's right-edge partial windows fall (layers.c:227-241) */
    const int maxCol = (N - g.shiftX) / g.nstepX;
    const int rem = (N - g.shiftX) % g.nstepX;
    const int colR = g.c0 + g.nstepC * (maxCol - 1) + rem / g.st + 1;
    const int startR = g.st - (g.padL + N - g.WL) % g.st - 1;
    /* grid-stride over the block chunks of the tile: grid.y is clamped to the 65535 limit, so a read of
     * any length the launch-group planner accepts is covered */
    for (int t0 = blockIdx.y * tchunk; t0 < Tt; t0 += (int)gridDim.y * tchunk) {
    const int t1 = min(Tt, t0 + tchunk);
    __syncthreads();      /* previous chunk's windows are no longer being read */
    /* stage the samples the regular windows of blocks t0..t1-1 touch, zero outside [0, N) */
    const int x0 = t0 * g.st - g.padL;
    for (int i = threadIdx.x; i < 16 * span; i += 256) {
        const int bb = i / span, k = i - bb * span;
        const int rr = tile * 16 + bb, xi = x0 + k;
        sX[i] = (xi >= 0 && xi < md.rN[rr]) ? sig[md.sig_off[rr] + xi] : 0.0f;
    }
    __syncthreads();
    const long long boff = md.tile_boff[tile];
    const int items = (t1 - t0) * nchunk * 64;
    /* item = (block t, chunk c of 16 filters, lane): a thread's lane -- hence its read and everything that depends
     * on the read's length only -- is the same for all its items; (t, c) advance by 4 chunks per item without
     * divisions */
    int c = (threadIdx.x >> 6) % nchunk, t = t0 + (threadIdx.x >> 6) / nchunk;
    /* t - c0 = kk * nstepC + ii, kept by increments (ii < 0 while t < c0) */
    int ii = t - g.c0, kk = 0;
    if (ii >= 0) { kk = ii / g.nstepC; ii -= kk * g.nstepC; }
    for (int it = threadIdx.x; it < items; it += 256, c += 4) {
        while (c >= nchunk) { c -= nchunk; t++; if (++ii == g.nstepC) { ii = 0; kk++; } }
        const int f0 = 16 * c + 4 * q;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (t < T) {
            acc = *(const f32x4 *)(sB + f0);
            /* regular window starting at t*st - padL (left edge: layers.c:190-196);
             * samples left of 0 are staged as zeros, which adds exact zeros */
            /* layers.c:209-224: column c0 + ii + kk * nstepC exists iff kk < (N - shiftX - ii * st) / nstepX, i.e.
             * iff (kk + 1) * nstepX <= N - shiftX - ii * st (conv_main_included without its divisions) */
            const bool regular = (t < g.c0) || (kk + 1) * g.nstepX <= N - g.shiftX - ii * g.st;
            if (regular) {
                const float *xw = sX + b * span + (t - t0) * g.st;
                for (int w = 0; w < g.WL; w++) {
                    const f32x4 wv = *(const f32x4 *)(sW + w * g.F + f0);
                    acc += wv * xw[w];
                }
            }
            /* right-edge partial windows (layers.c:227-241), straight from HBM: rare */
            for (int w = startR, cw = colR + startR / g.st; w < g.padR; w += g.st, cw++) {
                if (cw != t) continue;
                const float *x = sig + md.sig_off[rd];
                const int s = N - g.WL + 1 + w;
                for (int tap = 0; tap < g.WL - w - 1; tap++) {
                    const f32x4 wv = *(const f32x4 *)(sW + tap * g.F + f0);
                    acc += wv * x[s + tap];
                }
            }
            /* Operand range of the fp16 split products downstream (sh_kernels.h): |activation| < 1023.  layers.c:159-246 has
             * no such limit, so a value outside +-SH_ACT_LIMIT -- an un-normalised or non-finite signal; a med/MAD-
             * normalised one stays below ~50 -- is REPORTED: the read is flagged, the host gives it no call and says
             * why.  (The value is still bounded so that the read's tile-mates see finite numbers.) */
            for (int r = 0; r < 4; r++) {
                const float v = ACT ? d_tanh(acc[r]) : d_elu(acc[r]);
                /* (the pre-activation is tested too: d_exp's clamp turns a NaN into a finite value) */
                out_of_range |= !(__builtin_fabsf(v) < SH_ACT_LIMIT) | !(__builtin_fabsf(acc[r]) <= 3.0e38f);
                acc[r] = (v == v) ? __builtin_amdgcn_fmed3f(v, -SH_ACT_LIMIT, SH_ACT_LIMIT) : 0.0f;
            }
        }
        *(f32x4 *)(out + ((boff + t) * nchunk + c) * 256 + l * 4) = acc;
    }
    }
    if (out_of_range && bad) bad[rd] = 1u;
}

/* The same convolution on the matrix pipe (round 3).  A block of a tile is a [F x WL] . [WL x 16 reads] product: NCH chunks
 * of 16 filters x KST k steps of 4 taps, exact-fp32 MFMAs (v_mfma_f32_16x16x4_f32: fp32 products, fp32 accumulation -- a raw
 * signal has no operand range to respect), whose result layout IS the chunk layout the next kernel reads.  Per block and
 * wave: KST LDS reads of samples (shared by the chunks) + NCH * KST MFMAs, where the VALU form issues 22 LDS reads and 44
 * multiplies / additions per chunk; what is left is the activation.  The filter taps stay in registers (AREG: NCH * KST
 * <= 18 values) or come from LDS in MFMA lane order.  Which windows exist follows layers.c:209-241 exactly as in the VALU
 * form: a read whose regular window does not exist at a column contributes zeros to the B operand, the right edge's
 * partial windows (at most a few columns per read) are added by the lanes concerned afterwards, in the same order.
 * Why: run beside the recurrent layers of the previous launch group the VALU form took 3.0 ms and cost them 1.0 ms; a
 * plain 3 GB fill in its place costs them 0.08 ms (SH_CONV_FAKE): the interference was instruction issue, not memory. */
#ifndef SH_CONV_NT
#define SH_CONV_NT 1        /* 1 (measured -0.1 ms per step): the output (read much later, by the next group's first layer) as non-temporal stores */
#endif
#ifndef SH_CONV_ABL
#define SH_CONV_ABL 0       /* timing ablations of k_conv_mfma (1: no activation, 2: no MFMAs, 4: no output store); results invalid unless 0 */
#endif
template <int ACT, int NCH, int KST, bool AREG>
__device__ __forceinline__ void conv_act_mfma_body(const float *__restrict__ sig, const ShMeta &md,
                                                   const float *__restrict__ W /*[WL][F]*/,
                                                   const float *__restrict__ bias, const ShConvGeom &g,
                                                   float *__restrict__ out, int tchunk, unsigned *__restrict__ bad /*[npad]*/) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    bool out_of_range = false;
    float *sW = smem;                 /* WL*F: the partial windows' taps */
    float *sB = smem + g.WL * g.F;    /* F */
    float *sA = sB + g.F;             /* [NCH][KST][64]: taps in A-operand lane order (!AREG) */
    float *sX = sA + (AREG ? 0 : NCH * KST * 64);     /* 16 reads x span samples of this pass's windows */
    const int span = (tchunk - 1) * g.st + g.WL;
    const int tile = blockIdx.x;
    const int Tt = md.tile_T[tile];
    if ((int)blockIdx.y * tchunk >= Tt) return;
    for (int i = threadIdx.x; i < g.WL * g.F; i += 256) sW[i] = W[i];
    for (int i = threadIdx.x; i < g.F; i += 256) sB[i] = bias[i];
    const int l = threadIdx.x & 63, b = l & 15, q = l >> 4, wave = threadIdx.x >> 6;
    /* A operand of chunk c, k step s: lane (k = l / 16, row = l % 16) holds tap 4 s + k of filter 16 c + row */
    float a[AREG ? NCH : 1][AREG ? KST : 1];
    if (AREG) {
#pragma unroll
        for (int c = 0; c < NCH; c++)
#pragma unroll
            for (int ks = 0; ks < KST; ks++) a[c][ks] = (4 * ks + q < g.WL) ? W[(4 * ks + q) * g.F + 16 * c + b] : 0.0f;
    } else {
        for (int i = threadIdx.x; i < NCH * KST * 64; i += 256) {
            const int ll = i & 63, cs = i >> 6, c = cs / KST, ks = cs - c * KST, tap = 4 * ks + (ll >> 4);
            sA[i] = (tap < g.WL) ? W[tap * g.F + 16 * c + (ll & 15)] : 0.0f;
        }
    }
    const int rd = tile * 16 + b;
    const int N = md.rN[rd], T = md.rT[rd];
    /* where this read's right-edge partial windows fall (layers.c:227-241) */
    const int maxCol = (N - g.shiftX) / g.nstepX;
    const int rem = (N - g.shiftX) % g.nstepX;
    const int colR = g.c0 + g.nstepC * (maxCol - 1) + rem / g.st + 1;
    const int startR = g.st - (g.padL + N - g.WL) % g.st - 1;
    const long long boff = md.tile_boff[tile];
    for (int t0 = blockIdx.y * tchunk; t0 < Tt; t0 += (int)gridDim.y * tchunk) {
        const int t1 = min(Tt, t0 + tchunk);
        __syncthreads();      /* previous pass's windows are no longer being read */
        const int x0 = t0 * g.st - g.padL;
        for (int i = threadIdx.x; i < 16 * span; i += 256) {
            const int bb = i / span, k = i - bb * span;
            const int rr = tile * 16 + bb, xi = x0 + k;
            sX[i] = (xi >= 0 && xi < md.rN[rr]) ? sig[md.sig_off[rr] + xi] : 0.0f;
        }
        __syncthreads();
        constexpr int NB = AREG ? 1 : 2;     /* blocks in flight per wave: the rolled chunk loop of the LDS-tap build interleaves two chains */
        for (int tb = t0 + wave; tb < t1; tb += 4 * NB) {
            bool live[NB];
            float xb[NB][KST];
            int wpart[NB];
#pragma unroll
            for (int n = 0; n < NB; n++) {
                const int t = tb + 4 * n;
                live[n] = t < T && t < t1;
                /* layers.c:209-224: column c0 + ii + kk * nstepC exists iff (kk + 1) * nstepX <= N - shiftX - ii * st */
                bool regular = live[n];
                if (t >= g.c0) {
                    const int kk = (t - g.c0) / g.nstepC, ii = (t - g.c0) - kk * g.nstepC;
                    regular = live[n] && (kk + 1) * g.nstepX <= N - g.shiftX - ii * g.st;
                }
                /* B operand of k step s: lane (k = l / 16, read = l % 16) holds sample 4 s + k of the read's window;
                 * samples outside [0, N) are staged as zeros, taps past WL meet zero weights */
                const float *xw = sX + b * span + (min(t, t1 - 1) - t0) * g.st;
#pragma unroll
                for (int ks = 0; ks < KST; ks++) xb[n][ks] = (regular && 4 * ks + q < g.WL) ? xw[4 * ks + q] : 0.0f;
                /* is column t one of this read's right-edge partial windows (layers.c:227-241)?  w identifies it */
                wpart[n] = -1;
                for (int w = startR, cw = colR + startR / g.st; w < g.padR; w += g.st, cw++) if (cw == t && live[n]) wpart[n] = w;
            }
            auto chunk = [&](int c) {
                const int f0 = 16 * c + 4 * q;
                f32x4 acc[NB];
#pragma unroll
                for (int n = 0; n < NB; n++) acc[n] = *(const f32x4 *)(sB + f0);
#pragma unroll
                for (int ks = 0; ks < KST; ks++) {
                    const float av = AREG ? a[AREG ? c : 0][AREG ? ks : 0] : sA[(c * KST + ks) * 64 + l];
#pragma unroll
                    for (int n = 0; n < NB; n++) if (!(SH_CONV_ABL & 2)) acc[n] = mfma4(av, xb[n][ks], acc[n]);
                }
#pragma unroll
                for (int n = 0; n < NB; n++) {
                    const int t = tb + 4 * n;
                    if (t >= t1) continue;                 /* wave-uniform */
                    if (wpart[n] >= 0) {       /* rare: the lanes of reads that end here */
                        const float *x = sig + md.sig_off[rd];
                        const int s = N - g.WL + 1 + wpart[n];
                        for (int tap = 0; tap < g.WL - wpart[n] - 1; tap++) {
                            const f32x4 wv = *(const f32x4 *)(sW + tap * g.F + f0);
                            acc[n] += wv * x[s + tap];
                        }
                    }
                    if (!live[n]) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
                    else if (!(SH_CONV_ABL & 1)) {
#pragma unroll
                        for (int r = 0; r < 4; r++) {      /* as the VALU form: the read is flagged, the value bounded */
                            const float v = ACT ? d_tanh(acc[n][r]) : d_elu(acc[n][r]);
                            out_of_range |= !(__builtin_fabsf(v) < SH_ACT_LIMIT) | !(__builtin_fabsf(acc[n][r]) <= 3.0e38f);
                            acc[n][r] = (v == v) ? __builtin_amdgcn_fmed3f(v, -SH_ACT_LIMIT, SH_ACT_LIMIT) : 0.0f;
                        }
                    }
                    if (!(SH_CONV_ABL & 4) || acc[n][0] == 123.456f) {
                        f32x4 *dst = (f32x4 *)(out + ((boff + t) * NCH + c) * 256 + l * 4);
                        if (SH_CONV_NT) __builtin_nontemporal_store(acc[n], dst); else *dst = acc[n];
                    }
                }
            };
            if constexpr (AREG) {
#pragma unroll
                for (int c = 0; c < NCH; c++) chunk(c);
            } else {            /* taps from LDS: a rolled loop over the chunks keeps the build that runs beside k_gru_proj within 56 VGPRs */
#pragma unroll 1
                for (int c = 0; c < NCH; c++) chunk(c);
            }
        }
    }
    if (out_of_range && bad) bad[rd] = 1u;
}

/* Two builds of the same body.  k_conv_act: the compiler's choice of registers (48), for a launch group that has the GPU to
 * itself.  k_conv_act_bg: at most 48 VGPRs guaranteed (amdgpu_num_vgpr counts pairs): a wave of it fits beside three
 * k_gru_proj waves on a SIMD (144 each), so the convolution of the NEXT launch group runs on the prologue stream under
 * the recurrent layers of the current one.  (With k_gru_proj at 160 and this build at 32 VGPRs + 72 bytes of scratch:
 * 2.1 instead of 1.0 ms alone, and 0.45 ms more per step.) */
template <int ACT>
__global__ __launch_bounds__(256) void k_conv_act(const float *__restrict__ sig, ShMeta md, const float *__restrict__ W,
                                                  const float *__restrict__ bias, ShConvGeom g, float *__restrict__ out, int tchunk,
                                                  unsigned *__restrict__ bad) {
    conv_act_body<ACT>(sig, md, W, bias, g, out, tchunk, bad);
}
#ifndef SH_CONV_BG_VGPR_HALF
#define SH_CONV_BG_VGPR_HALF 24
#endif
template <int ACT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(SH_CONV_BG_VGPR_HALF))) void k_conv_act_bg(const float *__restrict__ sig, ShMeta md, const float *__restrict__ W,
                                                                                            const float *__restrict__ bias, ShConvGeom g, float *__restrict__ out,
                                                                                            int tchunk, unsigned *__restrict__ bad) {
    conv_act_body<ACT>(sig, md, W, bias, g, out, tchunk, bad);
}

template <int ACT, int NCH, int KST, bool AREG>
__global__ __launch_bounds__(256) void k_conv_mfma(const float *__restrict__ sig, ShMeta md, const float *__restrict__ W,
                                                   const float *__restrict__ bias, ShConvGeom g, float *__restrict__ out, int tchunk,
                                                   unsigned *__restrict__ bad) {
    conv_act_mfma_body<ACT, NCH, KST, AREG>(sig, md, W, bias, g, out, tchunk, bad);
}
#ifndef SH_CONVM_BG_VGPR_HALF
#define SH_CONVM_BG_VGPR_HALF 32      /* 64 VGPRs (the attribute counts pairs): beside three k_gru_proj waves of 144 a SIMD has 80 left */
#endif
template <int ACT, int NCH, int KST, bool AREG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(SH_CONVM_BG_VGPR_HALF))) void k_conv_mfma_bg(const float *__restrict__ sig, ShMeta md, const float *__restrict__ W,
                                                                                             const float *__restrict__ bias, ShConvGeom g, float *__restrict__ out,
                                                                                             int tchunk, unsigned *__restrict__ bad) {
    conv_act_mfma_body<ACT, NCH, KST, AREG>(sig, md, W, bias, g, out, tchunk, bad);
}

/* ------------------------------------------------------------------ */
/* L1: affine map  C = W^T X + b   (scrappie_matrix.c:323-351)           */
/* Weight-stationary: each wave keeps the A fragments of MT m-tiles in   */
/* registers and streams column blocks; no LDS, no barriers.             */
/* ------------------------------------------------------------------ */
template <int KQ, int MT, bool F32 = false>   /* F32: exact-fp32 MFMAs whatever K (weights outside the split products' range) */
__global__ __launch_bounds__(256) void k_affine(const float *__restrict__ in, float *__restrict__ out,
                                                const float *__restrict__ wfrag, const unsigned *__restrict__ wpiece,
                                                const float *__restrict__ bfrag, long long ncb,
                                                int mtiles_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mt0 = blockIdx.y * MT;
    constexpr bool SPLIT = (KQ % 2 == 0) && !F32;               /* odd K/16: exact-fp32 MFMA on the fp32 fragments */
    constexpr int KS = KQ / 2;
    float a[SPLIT ? 1 : MT][SPLIT ? 1 : KQ * 4];
    ShSplit ap[SPLIT ? MT : 1][SPLIT ? KS : 1];
    f32x4 bias[MT];
#pragma unroll
    for (int m = 0; m < MT; m++) {
        if constexpr (SPLIT) {
#pragma unroll
            for (int ks = 0; ks < KS; ks++) ap[m][ks] = load_pieces(wpiece + ((long long)(mt0 + m) * KS + ks) * 512, lane);
        } else {
#pragma unroll
            for (int r = 0; r < KQ * 4; r++) a[m][r] = wfrag[((long long)(mt0 + m) * (KQ * 4) + r) * 64 + lane];
        }
        bias[m] = *(const f32x4 *)(bfrag + ((mt0 + m) * 64 + lane) * 4);
    }
    const long long stride = (long long)gridDim.x * 4;
    long long cb = (long long)blockIdx.x * 4 + wave;
    if (cb >= ncb) return;
    f32x4 bcur[KQ], bnext[KQ];
#pragma unroll
    for (int mm = 0; mm < KQ; mm++) bcur[mm] = *(const f32x4 *)(in + (cb * KQ + mm) * 256 + lane * 4);
    for (; cb < ncb; cb += stride) {
        const long long nb = cb + stride;
        if (nb < ncb) {
#pragma unroll
            for (int mm = 0; mm < KQ; mm++)
                bnext[mm] = *(const f32x4 *)(in + (nb * KQ + mm) * 256 + lane * 4);
        }
        if constexpr (SPLIT) {
            ShSplit bp[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ks++) bp[ks] = split8(bcur[2 * ks], bcur[2 * ks + 1]);
#pragma unroll
            for (int m = 0; m < MT; m++)
                *(f32x4 *)(out + (cb * mtiles_total + mt0 + m) * 256 + lane * 4) = split_dot<KS>(ap[m], bp, bias[m]) * SH_OINV;
        } else {
#pragma unroll
            for (int m = 0; m < MT; m++) {
                f32x4 acc = bias[m];
#pragma unroll
                for (int mm = 0; mm < KQ; mm++) {
#pragma unroll
                    for (int s = 0; s < 4; s++) acc = mfma4(a[m][mm * 4 + s], bcur[mm][s], acc);
                }
                *(f32x4 *)(out + (cb * mtiles_total + mt0 + m) * 256 + lane * 4) = acc;
            }
        }
#pragma unroll
        for (int mm = 0; mm < KQ; mm++) bcur[mm] = bnext[mm];
    }
}

/* L1, LDS-resident weights: the register-stationary k_affine above needs
 * M/96 passes over the input (each wave can hold only 6 m-tiles of A
 * fragments), and PMC shows the 3 m-groups of a 288-row layer each re-fetch the
 * 3 GB input through the fabric: 18.4 GB per launch at 4.4 TB/s, i.e. it sits
 * on the HBM roof, not the MFMA one.  Here the whole fragment set (110 KiB for
 * 288 x 96, as fp16 pieces the size of the fp32 matrix) lives in LDS, one workgroup
 * per CU; a wave keeps NB column blocks as B pieces and walks ALL m-tiles, reading
 * the A pieces of a k step with two ds_read_b128.  Input is read once. */
template <int KQ, int NB, int NTH>
__global__ __launch_bounds__(NTH) void k_affine_lds(const float *__restrict__ in, float *__restrict__ out,
                                                    const float *__restrict__ wfrag, const unsigned *__restrict__ wpiece,
                                                    const float *__restrict__ bfrag, long long ncb,
                                                    int mtiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sA = smem;                                   /* [mtiles][KQ][64][4] fp32, or [mtiles][KS][2 pieces][64][4] words */
    float *sBias = smem + (size_t)mtiles * KQ * 256;    /* [mtiles][64][4] */
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NWV = NTH / 64;
    constexpr bool SPLIT = (KQ % 2 == 0);
    if constexpr (SPLIT) {
        unsigned *sP = (unsigned *)sA;
        for (int i = threadIdx.x; i < mtiles * KQ * 64; i += NTH) ((u32x4 *)sP)[i] = ((const u32x4 *)wpiece)[i];
    } else {
        /* regroup [mt][r = 4 mm + s][lane] -> [mt][mm][lane][s] */
        for (int i = threadIdx.x; i < mtiles * KQ * 256; i += NTH) {
            const int sidx = i & 3, l = (i >> 2) & 63, mm = (i >> 8) % KQ, mt = (i >> 8) / KQ;
            sA[i] = wfrag[((long long)mt * (KQ * 4) + mm * 4 + sidx) * 64 + l];
        }
    }
    for (int i = threadIdx.x; i < mtiles * 256; i += NTH) sBias[i] = bfrag[i];
    __syncthreads();
    /* column groups by fixed striding (the dynamic hand-out k_ff_lds uses measured 4 % slower here): workgroup w
     * owns groups w, w + gridDim.x, ..., its waves take them in turn */
    for (int j = wave;; j += NWV) {
        const long long cb0 = ((long long)j * gridDim.x + blockIdx.x) * NB;
        if (cb0 >= ncb) break;
        if constexpr (SPLIT) {
            constexpr int KS = KQ / 2;
            const unsigned *sP = (const unsigned *)sA;
            /* the columns are cut into pieces once per column group */
            ShSplit bp[KS][NB];
#pragma unroll
            for (int n = 0; n < NB; n++) {
                const long long cb = min(cb0 + n, ncb - 1);
#pragma unroll
                for (int ks = 0; ks < KS; ks++)
                    bp[ks][n] = split8(*(const f32x4 *)(in + (cb * KQ + 2 * ks) * 256 + lane * 4),
                                       *(const f32x4 *)(in + (cb * KQ + 2 * ks + 1) * 256 + lane * 4));
            }
            for (int mt = 0; mt < mtiles; mt++) {
                f32x4 acc[NB];
                const f32x4 bias = *(const f32x4 *)(sBias + (mt * 64 + lane) * 4);
#pragma unroll
                for (int n = 0; n < NB; n++) acc[n] = bias;
                ShSplit ap[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ks++) ap[ks] = load_pieces(sP + (mt * KS + ks) * 512, lane);
#pragma unroll
                for (int ks = 0; ks < KS; ks++) split_step<NB, 0>(ap[ks], bp[ks], acc);
#pragma unroll
                for (int ks = 0; ks < KS; ks++) split_step<NB, 1>(ap[ks], bp[ks], acc);
#pragma unroll
                for (int ks = 0; ks < KS; ks++) split_step<NB, 2>(ap[ks], bp[ks], acc);
#pragma unroll
                for (int n = 0; n < NB; n++)
                    if (cb0 + n < ncb) *(f32x4 *)(out + ((cb0 + n) * mtiles + mt) * 256 + lane * 4) = acc[n] * SH_OINV;
            }
            continue;
        }
        f32x4 b[NB][KQ];
#pragma unroll
        for (int n = 0; n < NB; n++) {
            const long long cb = min(cb0 + n, ncb - 1);
#pragma unroll
            for (int mm = 0; mm < KQ; mm++) b[n][mm] = *(const f32x4 *)(in + (cb * KQ + mm) * 256 + lane * 4);
        }
        for (int mt = 0; mt < mtiles; mt++) {
            f32x4 acc[NB];
            const f32x4 bias = *(const f32x4 *)(sBias + (mt * 64 + lane) * 4);
#pragma unroll
            for (int n = 0; n < NB; n++) acc[n] = bias;
#pragma unroll
            for (int mm = 0; mm < KQ; mm++) {
                const f32x4 a4 = *(const f32x4 *)(sA + ((mt * KQ + mm) * 64 + lane) * 4);
#pragma unroll
                for (int sidx = 0; sidx < 4; sidx++)
#pragma unroll
                    for (int n = 0; n < NB; n++) acc[n] = mfma4(a4[sidx], b[n][mm][sidx], acc[n]);
            }
#pragma unroll
            for (int n = 0; n < NB; n++)
                if (cb0 + n < ncb) *(f32x4 *)(out + ((cb0 + n) * mtiles + mt) * 256 + lane * 4) = acc[n];
        }
    }
}

/* feedforward2_tanh (layers.c:359 -> affine_map2, scrappie_matrix.c:353):
 * C = tanh(Wf^T Xf + Wb^T Xb + b), the layer that joins the two directions of
 * raw_r94's bi-GRU (networks.c:219,233) and of the events bi-LSTM.  Weight-stationary: a wave keeps MT m-tiles of
 * both matrices as fp16 pieces and streams column blocks; the two contractions are split products on one
 * accumulator (forward input first), tanh with the 2^-14 folded into its exponent.  (Round 1 / first half of
 * round 2: 96 exact-fp32 MFMAs of 32 cycles per m-tile and column block; now 18 of 16 -- the kernel sits on
 * its 9.2 GB of HBM traffic.) */
template <int KQ, int MT>
__global__ __launch_bounds__(512) void k_affine2_tanh(const float *__restrict__ inF, const float *__restrict__ inB,
                                                      float *__restrict__ out,
                                                      const unsigned *__restrict__ wpF,
                                                      const unsigned *__restrict__ wpB,
                                                      const float *__restrict__ bfrag, long long ncb,
                                                      int mtiles_total) {
    static_assert(KQ % 2 == 0, "k steps of 32");
    constexpr int KS = KQ / 2;
    /* the groups of MT m-tiles are spread over the wave quartets of one workgroup (not over blockIdx.y): the
     * quartets read the same column blocks at about the same time, so the inputs come from HBM once */
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3;
    const int mt0 = (threadIdx.x >> 8) * MT;
    ShSplit af[MT][KS], ab[MT][KS];
    f32x4 bias[MT];
#pragma unroll
    for (int m = 0; m < MT; m++) {
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            af[m][ks] = load_pieces(wpF + ((long long)(mt0 + m) * KS + ks) * 512, lane);
            ab[m][ks] = load_pieces(wpB + ((long long)(mt0 + m) * KS + ks) * 512, lane);
        }
        bias[m] = *(const f32x4 *)(bfrag + ((mt0 + m) * 64 + lane) * 4);
    }
    const long long stride = (long long)gridDim.x * 4;
    for (long long cb = (long long)blockIdx.x * 4 + wave; cb < ncb; cb += stride) {
        ShSplit xf[KS], xb[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            xf[ks] = split8(*(const f32x4 *)(inF + (cb * KQ + 2 * ks) * 256 + lane * 4), *(const f32x4 *)(inF + (cb * KQ + 2 * ks + 1) * 256 + lane * 4));
            xb[ks] = split8(*(const f32x4 *)(inB + (cb * KQ + 2 * ks) * 256 + lane * 4), *(const f32x4 *)(inB + (cb * KQ + 2 * ks + 1) * 256 + lane * 4));
        }
#pragma unroll
        for (int m = 0; m < MT; m++) {
            f32x4 acc = split_dot<KS>(af[m], xf, bias[m]);
            acc = split_dot<KS>(ab[m], xb, acc);
            *(f32x4 *)(out + (cb * mtiles_total + mt0 + m) * 256 + lane * 4) = d_tanh4_acc(acc);
        }
    }
}

#endif /* SH_CONV_AFFINE_H */
