/* sh_s1.h -- part of sh_kernels.h (included from there, in this order): S1: softmax_with_temperature up to exp + row sums; the finalisation shared by its consumers.
 * Device code for gfx950 only; see sh_kernels.h for conventions (layouts, split products, citations). */
#ifndef SH_S1_H
#define SH_S1_H

/* ------------------------------------------------------------------ */
/* S1 (first half): softmax_with_temperature up to exp + row sums       */
/* (layers.c:340-357).  E = exp((W^T (X / (tempW/tempb)) + b) / tempb),  */
/* sums[cb][b] = sum over the NS real states.  Normalisation and the     */
/* robust log (S2, layers.c:79) are applied by the consumers with the    */
/* same operations (multiply by 1/sum; log(mp + (1-mp) p)), so the       */
/* 3.3 MB/read posterior is written once and read once.                  */
/* Each wave takes NB column blocks and streams all m-tiles' fragments.  */
/* ------------------------------------------------------------------ */
/* ------------------------------------------------------------------ */
#define SH_SUM_GROUP 8     /* m-tiles per row-sum group: the tiles one wave of k_ff_viterbi owns */
template <int KQ, int NB, bool DIV>
__global__ __launch_bounds__(256) void k_ff_exp(const float *__restrict__ in, float *__restrict__ E,
                                                float *__restrict__ sums,
                                                const unsigned *__restrict__ wpiece,
                                                const float *__restrict__ bfrag, long long ncb,
                                                int mtiles, int mtp, int NS, float in_div, float out_div) {
    constexpr int KS = KQ / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long cb0 = ((long long)blockIdx.x * 4 + wave) * NB;
    if (cb0 >= ncb) return;
    f32x4 b[NB][KQ];
#pragma unroll
    for (int n = 0; n < NB; n++) {
        const long long cb = min(cb0 + n, ncb - 1);
#pragma unroll
        for (int mm = 0; mm < KQ; mm++) {
            f32x4 v = *(const f32x4 *)(in + (cb * KQ + mm) * 256 + lane * 4);
            if (in_div != 1.0f) v = v / in_div;          /* shift_scale_matrix_inplace: division (Q5) */
            b[n][mm] = v;
        }
    }
    /* row sums are formed per group of SH_SUM_GROUP consecutive m-tiles and the groups added in order:
     * the same association in k_ff_exp, k_ff_lds and k_ff_viterbi, so all three give identical bits */
    float part[NB], tot[NB];
#pragma unroll
    for (int n = 0; n < NB; n++) { part[n] = 0.0f; tot[n] = 0.0f; }
    /* the contraction runs as split products (split8 / split_step), the same sequence per accumulator as k_ff_lds */
    ShSplit bp[KQ / 2][NB];
#pragma unroll
    for (int ks = 0; ks < KQ / 2; ks++)
#pragma unroll
        for (int n = 0; n < NB; n++) bp[ks][n] = split8(b[n][2 * ks], b[n][2 * ks + 1]);
    ShSplit a[KS], an[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) a[ks] = load_pieces(wpiece + (long long)ks * 512, lane);
    const int q = lane >> 4;
    for (int mt = 0; mt < mtiles; mt++) {
        if (mt + 1 < mtiles) {
#pragma unroll
            for (int ks = 0; ks < KS; ks++) an[ks] = load_pieces(wpiece + ((long long)(mt + 1) * KS + ks) * 512, lane);
        }
        const f32x4 bias = *(const f32x4 *)(bfrag + (mt * 64 + lane) * 4);
        const int row0 = mt * 16 + 4 * q;
        f32x4 acc[NB];
#pragma unroll
        for (int n = 0; n < NB; n++) acc[n] = bias;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) split_step<NB, 0>(a[ks], bp[ks], acc);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) split_step<NB, 1>(a[ks], bp[ks], acc);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) split_step<NB, 2>(a[ks], bp[ks], acc);
#pragma unroll
        for (int n = 0; n < NB; n++) {
            f32x4 e;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float v = DIV ? d_exp((acc[n][r] * SH_OINV) / out_div) : d_exp_acc(acc[n][r]);     /* no max subtraction (Q2) */
                e[r] = (row0 + r < NS) ? v : 0.0f;
            }
            part[n] += (e[0] + e[1]) + (e[2] + e[3]);
            if (cb0 + n < ncb) *(f32x4 *)(E + ((cb0 + n) * mtiles + mt) * 256 + lane * 4) = e;
        }
#pragma unroll
        for (int ks = 0; ks < KS; ks++) a[ks] = an[ks];
        if ((mt + 1) % SH_SUM_GROUP == 0 || mt + 1 == mtiles) {
#pragma unroll
            for (int n = 0; n < NB; n++) {
                float v = part[n];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                tot[n] += v;
                part[n] = 0.0f;
            }
        }
    }
#pragma unroll
    for (int n = 0; n < NB; n++) {
        if (lane < 16 && cb0 + n < ncb) sums[(cb0 + n) * 16 + lane] = tot[n];
    }
}

/* ------------------------------------------------------------------ */
/* S1 for large batches: the same arithmetic as k_ff_exp with the weight  */
/* fragments resident in LDS.  The 1040 x 96 matrix (400 KB) does not     */
/* fit, so the state rows are cut into `nparts` groups of `mtp` m-tiles;  */
/* every workgroup walks the parts in order, refilling LDS once per part, */
/* and inside a part sweeps its share of the column blocks.  A wave meets */
/* the same column blocks in every part, so the row sums are carried from */
/* part to part through `sums` without atomics, always added in the same  */
/* order.  Each A fragment read (ds_read_b128 = 4 k-steps) feeds 4 x NB   */
/* MFMAs on NB independent accumulators.                                  */
/* ------------------------------------------------------------------ */
template <int KQ, int NB, int NTH, bool DIV>   /* DIV: tempb != 1, a true division per result (the compiler would otherwise
                                                  if-convert the test into an unconditional IEEE division + select) */
__global__ __launch_bounds__(NTH) void k_ff_lds(const float *__restrict__ in, float *__restrict__ E,
                                                float *__restrict__ sums,
                                                const unsigned *__restrict__ wpiece,
                                                const float *__restrict__ bfrag, long long ncb,
                                                int mtiles, int mtp, int NS, float in_div, float out_div, unsigned long long *dbg = nullptr) {
    constexpr int KS = KQ / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned *sA = (unsigned *)smem;                    /* [mtp][KS][2 pieces][64][4] words: the rows as fp16 pieces (cut on the host) */
    float *sBias = smem + (size_t)mtp * KQ * 256;       /* [mtp][64][4] */
    int *sNext = (int *)(sBias + (size_t)mtp * 256);    /* next column group of this workgroup */
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NV = NB * 4;                          /* results per lane per m-tile */
    constexpr int VPS = (NV + KQ - 1) / KQ;             /* ... finished per k-chunk of the next m-tile */
    const int q = lane >> 4;
    unsigned long long c_fill = 0, c_b = 0, c_loop = 0, c_sum = 0, c_t, c_n; long long c_tiles = 0;
#define FSTAMP(acc) do { if (dbg) { c_n = __builtin_readcyclecounter(); acc += c_n - c_t; c_t = c_n; } } while (0)
    if (dbg) c_t = __builtin_readcyclecounter();
    for (int mt0 = 0; mt0 < mtiles; mt0 += mtp) {
        const int nmt = min(mtp, mtiles - mt0);
        __syncthreads();                                /* previous part's readers are done */
        for (int i = threadIdx.x; i < nmt * KQ * 64; i += NTH) ((u32x4 *)sA)[i] = ((const u32x4 *)wpiece)[(long long)mt0 * KQ * 64 + i];
        for (int i = threadIdx.x; i < nmt * 256; i += NTH) sBias[i] = bfrag[(long long)mt0 * 256 + i];
        if (threadIdx.x == 0) *sNext = 0;
        __syncthreads();
        FSTAMP(c_fill);
        /* Column groups are handed out dynamically: the waves of a SIMD do not progress at
         * the same rate (the older one wins the matrix pipe), and with a fixed split the
         * faster half idles at the part barrier while the slower runs alone.  Workgroup w
         * owns groups w, w + gridDim.x, ...; a wave takes the next one when it is free. */
        for (;;) {
            int j = 0;
            if (lane == 0) j = atomicAdd(sNext, 1);
            j = __builtin_amdgcn_readfirstlane(j);
            const long long cb0 = ((long long)j * gridDim.x + blockIdx.x) * NB;
            if (cb0 >= ncb) break;
            f32x4 b[NB][KQ];
#pragma unroll
            for (int n = 0; n < NB; n++) {
                const long long cb = min(cb0 + n, ncb - 1);
#pragma unroll
                for (int mm = 0; mm < KQ; mm++) {
                    f32x4 v = *(const f32x4 *)(in + (cb * KQ + mm) * 256 + lane * 4);
                    if (in_div != 1.0f) v = v / in_div;          /* shift_scale_matrix_inplace: division (Q5) */
                    b[n][mm] = v;
                }
            }
            ShSplit bp[KS][NB];                                   /* the columns as fp16 pieces, reused by every m-tile */
#pragma unroll
            for (int ks = 0; ks < KS; ks++)
#pragma unroll
                for (int n = 0; n < NB; n++) bp[ks][n] = split8(b[n][2 * ks], b[n][2 * ks + 1]);
            FSTAMP(c_b);
            /* row sums: per group of SH_SUM_GROUP m-tiles (a part holds whole groups: mtp is a multiple), the groups
             * added in order; the running total crosses parts through `sums` */
            float part[NB], tot[NB];
#pragma unroll
            for (int n = 0; n < NB; n++) {
                part[n] = 0.0f;
                tot[n] = (mt0 == 0) ? 0.0f : sums[min(cb0 + n, ncb - 1) * 16 + (lane & 15)];
            }
            auto flush = [&]() {
#pragma unroll
                for (int n = 0; n < NB; n++) {
                    float v = part[n];
                    v += __shfl_xor(v, 16);
                    v += __shfl_xor(v, 32);
                    tot[n] += v;
                    part[n] = 0.0f;
                }
            };
            /* Software pipeline over the m-tiles: while the MFMAs of tile mt run, the exp /
             * row-sum / store of tile mt-1 is issued in KQ slices between the MFMA groups, so a
             * wave's VALU work sits under its own matrix instructions.  (The waves of a SIMD
             * share the matrix pipe evenly and otherwise fall into step: MFMA phases together
             * at a fraction of the rate each, then all epilogues with the pipe idle.)  Two sets of
             * accumulators alternate, so a tile's MFMAs never wait for the previous tile's results. */
            /* Columns past the end are clamped to the last one: such duplicates compute and
             * store the same values to the same place, which keeps the loop free of branches. */
            long long eoff[NB];
#pragma unroll
            for (int n = 0; n < NB; n++) eoff[n] = (min(cb0 + n, ncb - 1) * mtiles + mt0) * 256 + lane * 4;
            f32x4 acc0[NB], acc1[NB];
            f32x4 ex[NB];
            auto finish_slice = [&](const f32x4 (&ap)[NB], int mm, int ptile, bool lastrow) {      /* slice mm of the pending tile's epilogue */
#pragma unroll
                for (int v = mm * VPS; v < (mm + 1) * VPS && v < NV; v++) {
                    const int n = v >> 2, r = v & 3;
                    ex[n][r] = DIV ? d_exp((ap[n][r] * SH_OINV) / out_div) : d_exp_acc(ap[n][r]);   /* no max subtraction (Q2) */
                    if (r == 3) {
                        if (lastrow) {                                     /* rows >= NS are padding */
                            const int row0 = (mt0 + ptile) * 16 + 4 * q;
#pragma unroll
                            for (int rr = 0; rr < 4; rr++) ex[n][rr] = (row0 + rr < NS) ? ex[n][rr] : 0.0f;
                        }
                        part[n] += (ex[n][0] + ex[n][1]) + (ex[n][2] + ex[n][3]);
                        *(f32x4 *)(E + eoff[n] + (long long)ptile * 256) = ex[n];
                    }
                }
            };
            /* A pieces and bias of tile mt+1 are read from LDS while tile mt multiplies:
             * two register sets used alternately (the loop is unrolled by two), the reads
             * pinned to the top of the tile so their latency sits under the MFMAs */
            ShSplit A0[KS], A1[KS];
            f32x4 bias0, bias1;
            auto load_tile = [&](ShSplit (&A)[KS], f32x4 &bias, int mt) {
                bias = *(const f32x4 *)(sBias + (mt * 64 + lane) * 4);
#pragma unroll
                for (int ks = 0; ks < KS; ks++) A[ks] = load_pieces(sA + (mt * KS + ks) * 512, lane);
            };
            auto tile = [&](f32x4 (&acc)[NB], const f32x4 (&accp)[NB], ShSplit (&Au)[KS], f32x4 &bu, ShSplit (&Af)[KS], f32x4 &bf, int mt, bool pend) {
#pragma unroll
                for (int n = 0; n < NB; n++) acc[n] = bu;
                load_tile(Af, bf, min(mt + 1, nmt - 1));
                __builtin_amdgcn_sched_barrier(0);
                /* three passes over the k steps (a1 b2, a2 b1, a1 b1); the pending tile's epilogue in KQ slices between them */
                constexpr int NG = 3 * KS;
#pragma unroll
                for (int g = 0; g < NG; g++) {
                    const int ks = g % KS;
                    if (g < KS) split_step<NB, 0>(Au[ks], bp[ks], acc);
                    else if (g < 2 * KS) split_step<NB, 1>(Au[ks], bp[ks], acc);
                    else split_step<NB, 2>(Au[ks], bp[ks], acc);
                    if (pend) {
#pragma unroll
                        for (int mm = (g * KQ) / NG; mm < ((g + 1) * KQ) / NG; mm++) finish_slice(accp, mm, mt - 1, false);
                    }
                }
                if (pend && (mt % SH_SUM_GROUP) == 0) flush();          /* tile mt - 1 closed a group */
            };
            load_tile(A0, bias0, 0);
            tile(acc0, acc1, A0, bias0, A1, bias1, 0, false);
            int mt = 1;
            for (; mt + 1 < nmt; mt += 2) {                                 /* steady state: straight-line bodies */
                tile(acc1, acc0, A1, bias1, A0, bias0, mt, true);
                tile(acc0, acc1, A0, bias0, A1, bias1, mt + 1, true);
            }
            const bool lastrow = (mt0 + nmt == mtiles);
            if (mt < nmt) {
                tile(acc1, acc0, A1, bias1, A0, bias0, mt, true);
#pragma unroll
                for (int mm = 0; mm < KQ; mm++) finish_slice(acc1, mm, nmt - 1, lastrow);
            } else {
#pragma unroll
                for (int mm = 0; mm < KQ; mm++) finish_slice(acc0, mm, nmt - 1, lastrow);
            }
            FSTAMP(c_loop); c_tiles += nmt;
            flush();                                                       /* the part's last group */
#pragma unroll
            for (int n = 0; n < NB; n++) {
                if (lane < 16 && cb0 + n < ncb) sums[(cb0 + n) * 16 + lane] = tot[n];      /* real columns only */
            }
            FSTAMP(c_sum);
        }
    }
    if (dbg && lane == 0 && blockIdx.x == 100) { unsigned long long *d = dbg + wave * 8; d[0] = c_fill; d[1] = c_b; d[2] = c_loop; d[3] = c_sum; d[4] = (unsigned long long)c_tiles; }
}

/* finalisation shared by every consumer of E: row_normalise_inplace
 * (scrappie_matrix.c:385: multiply by reciprocal of the sum) followed by
 * robustlog_activation_inplace (layers.c:90-91) */
__device__ __forceinline__ float d_log(float x) {
#if SH_FAST_MATH
    /* raw v_log_f32 (log2) times ln 2; arguments here are >= min_prob, never denormal */
    return __builtin_amdgcn_logf(x) * 0.69314718055994530942f;
#else
    return logf(x);
#endif
}
/* rm = (1 / sum) * (1 - min_prob): log(min_prob + (1 - min_prob) e / sum) as one fused multiply-add, one v_log_f32 and
 * one multiply (the reference rounds e / sum and the product separately: a difference of an ulp of the probability,
 * far inside the posterior tolerance; every consumer of E goes through here, so they all see the same bits) */
__device__ __forceinline__ float fin_log(float e, float rm, float mp) { return d_log(__builtin_fmaf(e, rm, mp)); }
/* fin_log of four values with the multiply-add and the ln 2 scaling as packed f32 (v_pk_fma_f32 / v_pk_mul_f32: the same IEEE
 * operations, two values per instruction) */
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 fin_log4_pk(f32x4 e, float rm, float mp) {
#if SH_FAST_MATH
    const f32x2 rm2 = {rm, rm}, mp2 = {mp, mp};
    const f32x2 lo = __builtin_elementwise_fma((f32x2){e[0], e[1]}, rm2, mp2), hi = __builtin_elementwise_fma((f32x2){e[2], e[3]}, rm2, mp2);
    f32x2 g0 = {__builtin_amdgcn_logf(lo[0]), __builtin_amdgcn_logf(lo[1])}, g1 = {__builtin_amdgcn_logf(hi[0]), __builtin_amdgcn_logf(hi[1])};
    g0 = g0 * 0.69314718055994530942f;
    g1 = g1 * 0.69314718055994530942f;
    return (f32x4){g0[0], g0[1], g1[0], g1[1]};
#else
    return (f32x4){fin_log(e[0], rm, mp), fin_log(e[1], rm, mp), fin_log(e[2], rm, mp), fin_log(e[3], rm, mp)};
#endif
}
__device__ __forceinline__ float fin_post(float e, float recip, float mp, float mpm1, int want_log) {
    return want_log ? fin_log(e, recip * mpm1, mp) : e * recip;
}

#endif /* SH_S1_H */
