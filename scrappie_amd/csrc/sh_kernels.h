/* sh_kernels.h -- CDNA4 (gfx950) kernels of the raw basecalling hot path.
 *
 * DATA LAYOUT IN HBM (everything between the signal and the decoded path):
 * reads are grouped into TILES of 16 (sorted by length); a tile advances one
 * block (= one conv output column / time step t) at a time.  For tile T and
 * block t the activations of U units form a contiguous "column block" of
 * U*16 floats laid out as CHUNKS of 256 floats, one chunk per 16 units:
 *
 *     element (unit m, read b)  ->  chunk m>>4, float (((m>>2)&3)*16 + b)*4 + (m&3)
 *
 * i.e. a chunk is exactly the D (and, K-permuted, the B) operand image of one
 * v_mfma_f32_16x16x4_f32 tile: lane l = q*16 + b holds the 4 consecutive units
 * 4q..4q+3 of read b as one 16-byte vector, so every wave-level load/store of a
 * chunk is one lane-linear, fully coalesced 1 KiB access, and the same bytes
 * feed the next layer's MFMA B operand without any shuffle.
 *
 * MFMA use.  The layouts are those of v_mfma_f32_16x16x4_f32 (exact f32, A and B one VGPR per lane:
 * A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D[row=4*(l>>4)+r][col=l&15]).  With B
 * loaded as the 16-byte vector above, the four MFMAs of one 16-wide K group
 * consume k = 16*mm + 4*q + s (s = 0..3); weights are pre-permuted to match
 * ("fragments": [m-tile][K/4 regs][64 lanes]).  The hot contractions (projection, GRU / LSTM recurrence,
 * S1) run instead as SPLIT PRODUCTS on v_mfma_f32_16x16x32_f16 (split_pair / split_dot below): weights x 256
 * and activations x 64 are each cut into two fp16 pieces and the product accumulated in fp32, in units of
 * 2^-14, from three partial products (cross terms first); a lane holds the same 8 values of k per 32-wide step
 * as it holds in two consecutive fp32 chunks, so nothing above changes.
 * The exact-fp32 MFMA remains in the small-shape kernels (k_gru, k_affine with K odd).
 *
 * Reference rows each kernel replaces are cited inline;
 * file:line under /root/reference/src.
 */
#ifndef SH_KERNELS_H
#define SH_KERNELS_H
#include <type_traits>

#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SH_BIG 1.e30f          /* decode.c:8 */
#define SH_TB_STAY 0u
#define SH_TB_STEP 1u          /* + r, r < 4  */
#define SH_TB_SKIP 5u          /* + r, r < 16 */
#define SH_TB_SLIP 21u         /* + r, r < 64 */
#define SH_TB_START 85u

/* ------------------------------------------------------------------ */
/* device math: same algebraic forms as util.h:170-198                  */
/* ------------------------------------------------------------------ */
/* SH_FAST_MATH=1 (default): exp/log/reciprocal through the hardware
 * transcendental unit (v_exp_f32, v_log_f32, v_rcp_f32; ~1 ulp each).  Every
 * GPU parity test passes at the stated tolerances with it.  Build with
 * EXTRA_HIPFLAGS=-DSH_FAST_MATH=0 for the libm-accurate variants. */
#ifndef SH_FAST_MATH
#define SH_FAST_MATH 1
#endif

__device__ __forceinline__ float d_exp(float x) {
    /* exp_ps clamps its argument (sse_mathfun.h:233-234) */
#if SH_FAST_MATH
    x = __builtin_amdgcn_fmed3f(x, -88.3762626647949f, 88.3762626647949f);   /* one v_med3_f32 */
    return __builtin_amdgcn_exp2f(x * 1.44269504088896341f);   /* raw v_exp_f32; |x| <= 88.4 after the clamp */
#else
    x = fminf(x, 88.3762626647949f);
    x = fmaxf(x, -88.3762626647949f);
    return expf(x);
#endif
}
__device__ __forceinline__ float d_rcp(float x) {
#if SH_FAST_MATH
    return __builtin_amdgcn_rcpf(x);     /* raw v_rcp_f32 (1 ulp); __frcp_rn expands to a full IEEE division */
#else
    return 1.0f / x;
#endif
}
__device__ __forceinline__ float d_logistic(float x) { return d_rcp(1.0f + d_exp(-x)); }
__device__ __forceinline__ float d_tanh(float x) {
    const float y = d_logistic(x + x);
    return (y + y) - 1.0f;
}
/* the same operations on four values as vector arithmetic, which the compiler lowers to packed f32 VALU
 * (v_pk_mul_f32 / v_pk_add_f32: two lanes per instruction, identical IEEE results) whether or not the SLP
 * vectoriser is on.  The recurrent kernels run their gate activations with the matrix pipe idle, so there
 * the halved instruction count pays; next to MFMAs packed f32 is slow (see the Makefile). */
__device__ __forceinline__ f32x4 d_exp4(f32x4 x) {
#if SH_FAST_MATH
#pragma unroll
    for (int k = 0; k < 4; k++) x[k] = __builtin_amdgcn_fmed3f(x[k], -88.3762626647949f, 88.3762626647949f);
    x = x * 1.44269504088896341f;
#pragma unroll
    for (int k = 0; k < 4; k++) x[k] = __builtin_amdgcn_exp2f(x[k]);
    return x;
#else
#pragma unroll
    for (int k = 0; k < 4; k++) x[k] = d_exp(x[k]);
    return x;
#endif
}
__device__ __forceinline__ f32x4 d_logistic4(f32x4 x) {
    f32x4 y = 1.0f + d_exp4(-x);
#pragma unroll
    for (int k = 0; k < 4; k++) y[k] = d_rcp(y[k]);
    return y;
}
__device__ __forceinline__ f32x4 d_tanh4(f32x4 x) {
    const f32x4 y = d_logistic4(x + x);
    return (y + y) - 1.0f;
}
__device__ __forceinline__ float d_elu(float x) { return (x >= 0.0f) ? x : (d_exp(x) - 1.0f); }
__device__ __forceinline__ float d_lse(float x, float y) {   /* util.h:162 */
#if SH_FAST_MATH
    /* v_exp_f32 / v_log_f32: log(1 + e) taken literally is within 6e-8 ABSOLUTE of log1p(e) (1 + e rounds to an
     * ulp of 1), against partition functions of ~2e3 whose own ulp is 1.2e-4; the library log1pf(expf()) is a
     * dependent chain of ~60 instructions, four of them per block on k_crf's critical path */
    const float e = __builtin_amdgcn_exp2f(-fabsf(x - y) * 1.44269504088896341f);
    return fmaxf(x, y) + __builtin_amdgcn_logf(1.0f + e) * 0.69314718055994530942f;
#else
    return fmaxf(x, y) + log1pf(expf(-fabsf(x - y)));
#endif
}

/* Workgroup barrier that orders LDS traffic only: waits for this wave's LDS
 * operations (lgkmcnt) and not for its global loads/stores, so prefetches and
 * result stores stay in flight across the barrier (__syncthreads() drains vmcnt too). */
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

/* Wait until *flag >= need (relaxed agent-scope polls, one every ~1 us).  The wait is bounded by WALL time
 * (s_memrealtime, 100 MHz), not by a poll count: a producer workgroup that is merely late (shared or
 * pre-empted device, skewed dispatch) is waited for; after SH_HANDOVER_TIMEOUT_S seconds the caller raises
 * the launch group's error word and the host re-runs the group on whole tiles (scrappie_hip_collect). */
#ifndef SH_HANDOVER_TIMEOUT_S
#define SH_HANDOVER_TIMEOUT_S 20ull
#endif
__device__ __forceinline__ bool sh_wait_flag(const unsigned *flag, unsigned need) {
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) return true;
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        __builtin_amdgcn_s_sleep(32);
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) return true;
        if (wall_clock64() - t0 > SH_HANDOVER_TIMEOUT_S * 100000000ull) return false;
    }
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

/* fp32 contraction on the 16-bit matrix pipe without leaving fp32 accuracy.  Both operands are brought to a
 * common power-of-two scale -- weights x 256 (on the host, once), activations x 64 -- and cut into two fp16
 * pieces x' = p1 + p2: p1 = fp16(x') (round to nearest even), p2 = fp16(x' - p1); the residual has at most 13
 * significant bits and is exact in fp32, and the up-scaling keeps it out of fp16's subnormal range for every
 * |weight| > 5e-4 and |activation| > 2e-3 (below that the absolute error is < 5e-10).  p1 + p2 carries 22 bits
 * of x.  A product a . b is accumulated in fp32 by v_mfma_f32_16x16x32_f16 on ONE accumulator, in 2^14 units
 * (it starts from 2^14 x bias, or from the gate input in the same units), cross terms first:
 *     acc += sum_k a1 b2;   acc += sum_k a2 b1;   acc += sum_k a1 b1          (a2 b2 < 2^-22 of the product)
 * i.e. 3 x 16 cycles per 32-wide k step against 8 x 32 cycles of v_mfma_f32_16x16x4_f32.  The consumers take
 * the 2^-14 into a multiplication they perform anyway (the log2(e) of exp, the output scaling): power-of-two
 * scalings are exact, so nothing is rounded twice.  Every kernel below uses exactly this form, so kernels that
 * compute the same thing agree bit for bit.  tools/split_probe.hip on [288 x 96] . [96 x 16] against float64:
 * rms error 5.3e-8 (max 4.3e-7) against 1.0e-7 (1.2e-6) for the exact-fp32 MFMA, and no worse than it for
 * operands scaled from 1e-2 to 10 (profiles/r2_split_probe.txt).  Operand range: |weight| < 255, |activation| <
 * 1023 (fp16's largest finite value is 65504): activations here are gate outputs in (-1, 1), residual sums of
 * them, and convolution outputs of med/MAD-normalised signal (k_conv_act clamps at +-1000).  A lane holds the
 * same 8 values of k per 32-wide step as it holds in two consecutive fp32 chunks, so the fp32 layouts carry
 * over unchanged. */
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#ifndef SH_WSCALE
#define SH_WSCALE 256.0f            /* weights (host: make_piece_frags) */
#define SH_ASCALE 64.0f             /* activations (split_pair) */
#endif
#define SH_OSCALE (SH_WSCALE * SH_ASCALE)          /* accumulators: 2^14 */
#define SH_OINV (1.0f / SH_OSCALE)
struct ShSplit { f16x8 p1, p2; };
__device__ __forceinline__ void split_pair(float x, float y, unsigned &w1, unsigned &w2) {
    const float xs = x * SH_ASCALE, ys = y * SH_ASCALE;
    const f16x2 h = __builtin_convertvector((f32x2){xs, ys}, f16x2);               /* round to nearest even */
    w1 = __builtin_bit_cast(unsigned, h);
    w2 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){xs - (float)h[0], ys - (float)h[1]}, f16x2));
}
__device__ __forceinline__ ShSplit split8(f32x4 lo, f32x4 hi) {
    unsigned w1[4], w2[4];
    split_pair(lo[0], lo[1], w1[0], w2[0]);
    split_pair(lo[2], lo[3], w1[1], w2[1]);
    split_pair(hi[0], hi[1], w1[2], w2[2]);
    split_pair(hi[2], hi[3], w1[3], w2[3]);
    ShSplit s;
    s.p1 = __builtin_bit_cast(f16x8, (u32x4){w1[0], w1[1], w1[2], w1[3]});
    s.p2 = __builtin_bit_cast(f16x8, (u32x4){w2[0], w2[1], w2[2], w2[3]});
    return s;
}
/* pieces of one 32-wide k step as they lie in memory (weights cut on the host, activations published through LDS):
 * [piece][64 lanes][4 words] */
__device__ __forceinline__ ShSplit load_pieces(const unsigned *p, int lane) {
    ShSplit s;
    s.p1 = __builtin_bit_cast(f16x8, *(const u32x4 *)(p + lane * 4));
    s.p2 = __builtin_bit_cast(f16x8, *(const u32x4 *)(p + 256 + lane * 4));
    return s;
}
__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
/* the three passes over the k steps, on NB independent column blocks.  PASS 0: a1 b2, 1: a2 b1, 2: a1 b1 */
template <int NB, int PASS>
__device__ __forceinline__ void split_step(const ShSplit &a, const ShSplit (&b)[NB], f32x4 (&acc)[NB]) {
#pragma unroll
    for (int n = 0; n < NB; n++)
        acc[n] = (PASS == 0) ? mfma16(a.p1, b[n].p2, acc[n]) : (PASS == 1) ? mfma16(a.p2, b[n].p1, acc[n]) : mfma16(a.p1, b[n].p1, acc[n]);
}
/* a whole contraction of KS k steps on one column block */
template <int KS>
__device__ __forceinline__ f32x4 split_dot(const ShSplit (&a)[KS], const ShSplit (&b)[KS], f32x4 acc) {
#pragma unroll
    for (int ks = 0; ks < KS; ks++) acc = mfma16(a[ks].p1, b[ks].p2, acc);
#pragma unroll
    for (int ks = 0; ks < KS; ks++) acc = mfma16(a[ks].p2, b[ks].p1, acc);
#pragma unroll
    for (int ks = 0; ks < KS; ks++) acc = mfma16(a[ks].p1, b[ks].p1, acc);
    return acc;
}
/* ... and two of them on the same column block, interleaved (two dependency chains) */
template <int KS>
__device__ __forceinline__ void split_dot2(const ShSplit (&a0)[KS], const ShSplit (&a1)[KS], const ShSplit (&b)[KS], f32x4 &acc0, f32x4 &acc1) {
#pragma unroll
    for (int ks = 0; ks < KS; ks++) { acc0 = mfma16(a0[ks].p1, b[ks].p2, acc0); acc1 = mfma16(a1[ks].p1, b[ks].p2, acc1); }
#pragma unroll
    for (int ks = 0; ks < KS; ks++) { acc0 = mfma16(a0[ks].p2, b[ks].p1, acc0); acc1 = mfma16(a1[ks].p2, b[ks].p1, acc1); }
#pragma unroll
    for (int ks = 0; ks < KS; ks++) { acc0 = mfma16(a0[ks].p1, b[ks].p1, acc0); acc1 = mfma16(a1[ks].p1, b[ks].p1, acc1); }
}
/* gate activations of an accumulator in 2^14 units: the same algebraic forms as d_logistic / d_tanh (util.h:180-188)
 * with the unit folded into the log2(e) factor of the exponential -- exact, a power of two.  (No clamp of the
 * exponent: past +-88.4 the result of 1 / (1 + e) is 0 or 1 to within 1e-38 either way.) */
__device__ __forceinline__ f32x4 d_logistic4_acc(f32x4 a) {
#if SH_FAST_MATH
    f32x4 t = a * (-1.44269504088896341f * SH_OINV);
#pragma unroll
    for (int k = 0; k < 4; k++) t[k] = __builtin_amdgcn_exp2f(t[k]);
    t = 1.0f + t;
#pragma unroll
    for (int k = 0; k < 4; k++) t[k] = d_rcp(t[k]);
    return t;
#else
    return d_logistic4(a * SH_OINV);
#endif
}
/* exp of an accumulator in 2^14 units with exp_ps's clamp (sse_mathfun.h:233-234): as d_exp, the unit folded in */
__device__ __forceinline__ float d_exp_acc(float a) {
#if SH_FAST_MATH
    a = __builtin_amdgcn_fmed3f(a, -88.3762626647949f * SH_OSCALE, 88.3762626647949f * SH_OSCALE);
    return __builtin_amdgcn_exp2f(a * (1.44269504088896341f * SH_OINV));
#else
    return d_exp(a * SH_OINV);
#endif
}
__device__ __forceinline__ f32x4 d_tanh4_acc(f32x4 a) {
#if SH_FAST_MATH
    f32x4 t = a * (-2.0f * 1.44269504088896341f * SH_OINV);
#pragma unroll
    for (int k = 0; k < 4; k++) t[k] = __builtin_amdgcn_exp2f(t[k]);
    t = 1.0f + t;
#pragma unroll
    for (int k = 0; k < 4; k++) t[k] = d_rcp(t[k]);
    return (t + t) - 1.0f;
#else
    return d_tanh4(a * SH_OINV);
#endif
}

/* ------------------------------------------------------------------ */
/* per-launch-group metadata (device arrays, tiled read order)          */
/* ------------------------------------------------------------------ */
struct ShMeta {
    const unsigned long long *sig_off;   /* [npad] offset of the read's first sample */
    const int *rN;                       /* [npad] samples (0 = padding read) */
    const int *rT;                       /* [npad] blocks */
    const int *tile_T;                   /* [ntile] max blocks in tile */
    const long long *tile_boff;          /* [ntile] first column block of tile */
};

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
__global__ __launch_bounds__(256) void k_conv_act(const float *__restrict__ sig, ShMeta md,
                                                  const float *__restrict__ W /*[WL][F]*/,
                                                  const float *__restrict__ bias, ShConvGeom g,
                                                  float *__restrict__ out, int tchunk) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
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
            /* (the clamp: operand range of the fp16 split products downstream; never reached by a normalised signal) */
            for (int r = 0; r < 4; r++) acc[r] = __builtin_amdgcn_fmed3f(ACT ? d_tanh(acc[r]) : d_elu(acc[r]), -1000.0f, 1000.0f);
        }
        *(f32x4 *)(out + ((boff + t) * nchunk + c) * 256 + l * 4) = acc;
    }
    }
}

/* ------------------------------------------------------------------ */
/* L1: affine map  C = W^T X + b   (scrappie_matrix.c:323-351)           */
/* Weight-stationary: each wave keeps the A fragments of MT m-tiles in   */
/* registers and streams column blocks; no LDS, no barriers.             */
/* ------------------------------------------------------------------ */
template <int KQ, int MT>
__global__ __launch_bounds__(256) void k_affine(const float *__restrict__ in, float *__restrict__ out,
                                                const float *__restrict__ wfrag, const unsigned *__restrict__ wpiece,
                                                const float *__restrict__ bfrag, long long ncb,
                                                int mtiles_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mt0 = blockIdx.y * MT;
    constexpr bool SPLIT = (KQ % 2 == 0);                       /* odd K/16: exact-fp32 MFMA on the fp32 fragments */
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

/* ------------------------------------------------------------------ */
/* G1/G2 (+R1): one GRU layer, whole sequence, one tile of 16 reads per  */
/* workgroup (layers.c:373-527, :303).  NU = S/16 waves; wave u owns     */
/* units 16u..16u+15: the z, r and candidate rows of those units stay in */
/* its registers as MFMA A fragments for all T steps, the 16-read state  */
/* is exchanged through a 16*S float LDS image in B-operand layout.      */
/* ------------------------------------------------------------------ */
template <int NU>
__global__ __launch_bounds__(64 * NU) void k_gru(const float *__restrict__ xaff, float *__restrict__ out,
                                                 const float *__restrict__ resid,
                                                 const float *__restrict__ sWfrag /*[2NU][4NU][64]*/,
                                                 const float *__restrict__ sW2frag /*[NU][4NU][64]*/,
                                                 ShMeta md, int backward, unsigned long long *dbgbuf) {
    constexpr int KR = NU * 4;                 /* A regs per m-tile = S/4 */
    const int dbg = backward >> 8;             /* experiment switch (0 in production) */
    backward &= 1;
    __shared__ __attribute__((aligned(16))) float lds[2 * NU * 256];
    float *lds_h = lds, *lds_rh = lds + NU * 256;
    const int lane = threadIdx.x & 63, u = threadIdx.x >> 6;
    const int tile = blockIdx.x;
    const int Tt = md.tile_T[tile];
    const long long boff = md.tile_boff[tile];
    const int myT = md.rT[tile * 16 + (lane & 15)];

    float wz[KR], wr[KR], wh[KR];
#pragma unroll
    for (int r = 0; r < KR; r++) {
        wz[r] = sWfrag[((long long)u * KR + r) * 64 + lane];
        wr[r] = sWfrag[((long long)(NU + u) * KR + r) * 64 + lane];
        wh[r] = sW2frag[((long long)u * KR + r) * 64 + lane];
    }
    f32x4 h = {0.f, 0.f, 0.f, 0.f};
    *(f32x4 *)(lds_h + u * 256 + lane * 4) = h;
    if (dbg == 4) { const unsigned ph = ((unsigned)blockIdx.x * 2654435761u) >> 28; for (unsigned i = 0; i < ph; i++) __builtin_amdgcn_s_sleep(8); }
    if (dbg == 5 && ((blockIdx.x >> 3) & 1)) { for (int i = 0; i < 8; i++) __builtin_amdgcn_s_sleep(8); }
    __syncthreads();

    const long long xstride = 3LL * NU * 256;     /* floats per column block of xaff */
    auto xptr = [&](int t, int chunk) { return xaff + (boff + t) * xstride + chunk * 256 + lane * 4; };
    int t = backward ? Tt - 1 : 0;
    const int dt = backward ? -1 : 1;
    /* gate inputs are fetched two steps ahead (HBM latency > one step) */
    f32x4 xz0, xr0, xh0, xz1, xr1, xh1;
    xz0 = xr0 = xh0 = xz1 = xr1 = xh1 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (Tt > 0) { xz0 = *(const f32x4 *)xptr(t, u); xr0 = *(const f32x4 *)xptr(t, NU + u); xh0 = *(const f32x4 *)xptr(t, 2 * NU + u); }
    if (Tt > 1) { xz1 = *(const f32x4 *)xptr(t + dt, u); xr1 = *(const f32x4 *)xptr(t + dt, NU + u); xh1 = *(const f32x4 *)xptr(t + dt, 2 * NU + u); }
    unsigned long long tA = 0, tB = 0, tC = 0, tD = 0, tE = 0, ts0 = 0, ts1;
#define STAMP(acc) do { if (dbgbuf) { ts1 = __builtin_readcyclecounter(); acc += ts1 - ts0; ts0 = ts1; } } while (0)
    unsigned long long wall0 = 0;
    if (dbgbuf) { ts0 = __builtin_readcyclecounter(); wall0 = wall_clock64(); }
    for (int step = 0; step < Tt; step++, t += dt) {
        f32x4 accz = xz0, accr = xr0, acch = xh0;
        xz0 = xz1; xr0 = xr1; xh0 = xh1;
        if (step + 2 < Tt) {
            xz1 = *(const f32x4 *)xptr(t + 2 * dt, u);
            xr1 = *(const f32x4 *)xptr(t + 2 * dt, NU + u);
            xh1 = *(const f32x4 *)xptr(t + 2 * dt, 2 * NU + u);
        }
        /* Reset gate first: only r is needed before the barrier (layers.c:505,
         * :511-516).  The update-gate GEMM and its logistic are issued after the
         * r*h image is written, so they fill the barrier / LDS round trip. */
        f32x4 hb[NU];
#pragma unroll
        for (int mm = 0; mm < NU; mm++) hb[mm] = *(const f32x4 *)(lds_h + mm * 256 + lane * 4);
        f32x4 accr2 = {0.f, 0.f, 0.f, 0.f};
        if (dbg != 1)
#pragma unroll
        for (int mm = 0; mm < NU; mm++) {
            accr = mfma4(wr[mm * 4 + 0], hb[mm][0], accr);
            accr2 = mfma4(wr[mm * 4 + 1], hb[mm][1], accr2);
            accr = mfma4(wr[mm * 4 + 2], hb[mm][2], accr);
            accr2 = mfma4(wr[mm * 4 + 3], hb[mm][3], accr2);
        }
        accr += accr2;
        if (dbgbuf) asm volatile("" :: "v"(accr[0]));
        STAMP(tA);
        f32x4 rh;
#pragma unroll
        for (int i = 0; i < 4; i++) rh[i] = d_logistic(accr[i]) * h[i];          /* layers.c:515 */
        *(f32x4 *)(lds_rh + u * 256 + lane * 4) = rh;
        if (dbgbuf) asm volatile("" :: "v"(rh[0]));
        STAMP(tB);
        __syncthreads();
        STAMP(tC);
        /* update-gate GEMM (needs only h, still in hb) runs while the r*h image
         * comes back from LDS */
        f32x4 rb[NU];
#pragma unroll
        for (int mm = 0; mm < NU; mm++) rb[mm] = *(const f32x4 *)(lds_rh + mm * 256 + lane * 4);
        f32x4 accz2 = {0.f, 0.f, 0.f, 0.f};
        if (dbg != 1)
#pragma unroll
        for (int mm = 0; mm < NU; mm++) {
            accz = mfma4(wz[mm * 4 + 0], hb[mm][0], accz);
            accz2 = mfma4(wz[mm * 4 + 1], hb[mm][1], accz2);
            accz = mfma4(wz[mm * 4 + 2], hb[mm][2], accz);
            accz2 = mfma4(wz[mm * 4 + 3], hb[mm][3], accz2);
        }
        /* xF[2S:3S] += sW2^T (r*h)   (layers.c:517) */
        f32x4 acch2 = {0.f, 0.f, 0.f, 0.f};
        if (dbg != 1)
#pragma unroll
        for (int mm = 0; mm < NU; mm++) {
            acch = mfma4(wh[mm * 4 + 0], rb[mm][0], acch);
            acch2 = mfma4(wh[mm * 4 + 1], rb[mm][1], acch2);
            acch = mfma4(wh[mm * 4 + 2], rb[mm][2], acch);
            acch2 = mfma4(wh[mm * 4 + 3], rb[mm][3], acch2);
        }
        accz += accz2;
        f32x4 z;
#pragma unroll
        for (int i = 0; i < 4; i++) z[i] = d_logistic(accz[i]);   /* VALU work in the shadow of the MFMAs above */
        acch += acch2;
        const bool active = t < myT;
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float hbar = d_tanh(acch[i]);
            const float hn = z[i] * h[i] + (1.0f - z[i]) * hbar;   /* layers.c:525 */
            h[i] = active ? hn : 0.0f;
            o[i] = h[i];
        }
        *(f32x4 *)(lds_h + u * 256 + lane * 4) = h;
        if (resid) {   /* residual_inplace(layer input, gru output): networks.c:583 */
            const f32x4 rv = *(const f32x4 *)(resid + ((boff + t) * NU + u) * 256 + lane * 4);
            o += rv;
        }
        if (dbg != 2) *(f32x4 *)(out + ((boff + t) * NU + u) * 256 + lane * 4) = o;
        STAMP(tD);
        __syncthreads();
        STAMP(tE);
    }
    if (dbgbuf && lane == 0) { unsigned long long *d = dbgbuf + ((long long)blockIdx.x * NU + u) * 8; d[0] = tA; d[1] = tB; d[2] = tC; d[3] = tD; d[4] = tE; d[5] = Tt; d[6] = wall0; d[7] = wall_clock64(); }
    if (dbg == 2) *(f32x4 *)(out + (boff * NU + u) * 256 + lane * 4) = h;
}


/* ------------------------------------------------------------------ */
/* R1, lane-scheduled, exact-fp32 MFMA (v_mfma_f32_16x16x4_f32): the      */
/* reference the split-product kernels below were measured against        */
/* (SH_GRU_F32=1).  Two lanes per workgroup (wave                          */
/* groups of NU waves, one tile each, SIMD load (3,3,3,3)); every lane    */
/* walks a list of segments = steps [s0,s1) of a tile (sh_sched.h), so    */
/* 625 tiles keep all 512 lanes of 256 CUs busy for 1.22 tile-times       */
/* instead of 3 tiles on some CUs and 2 on others.  A tile cut between two */
/* lanes hands its state over through HBM (agent-scope stores, arrival     */
/* counter); the consumer polls with a bounded spin.                      */
/* Step anatomy: the reset-gate GEMM alone sits in front of the first      */
/* barrier; the update-gate GEMM runs after it on the h fragments still    */
/* in registers, covering the LDS latency of the r*h exchange, and its     */
/* logistic issues under the candidate GEMM's MFMAs.                       */
/* ------------------------------------------------------------------ */
struct ShGruSegD { int tile, s0, s1, pad; };
struct ShGruLanes {
    const int *lane_off;         /* [2 * gridDim.x + 1] */
    const ShGruSegD *seg;
    const int *wg_iter;          /* [gridDim.x] */
    float *hstate;               /* [ntile][NU * 256] */
    unsigned *flag;              /* [ntile + 1]; last = error flag */
    int ntile;
};

template <int NU, bool STAMP = false>
__global__ __launch_bounds__(128 * NU) void k_gru_lanes(const float *__restrict__ xaff, float *__restrict__ out,
                                                       const float *__restrict__ resid,
                                                       const float *__restrict__ sWfrag,
                                                       const float *__restrict__ sW2frag, ShMeta md,
                                                       int backward, ShGruLanes L, unsigned long long *dbgbuf = nullptr) {
    constexpr int KR = NU * 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];   /* [2 lanes][h | r*h][NU][256] */
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int u = wave % NU, grp = wave / NU;
    const int ln = blockIdx.x * 2 + grp;

    float wz[KR], wr[KR], wh[KR];
#pragma unroll
    for (int r = 0; r < KR; r++) {
        wz[r] = sWfrag[((long long)u * KR + r) * 64 + lane];
        wr[r] = sWfrag[((long long)(NU + u) * KR + r) * 64 + lane];
        wh[r] = sW2frag[((long long)u * KR + r) * 64 + lane];
    }
    float *lds_h = lds + grp * 2 * NU * 256, *lds_rh = lds_h + NU * 256;
    const long long xstride = 3LL * NU * 256;
    const int nit = L.wg_iter[blockIdx.x];
    int sgi = __builtin_amdgcn_readfirstlane(L.lane_off[ln]);
    const int sge = __builtin_amdgcn_readfirstlane(L.lane_off[ln + 1]);
    int my_it = 0;                                  /* steps of this lane; it idles (barriers only) afterwards */
    for (int i = sgi; i < sge; i++) my_it += L.seg[i].s1 - L.seg[i].s0;
    my_it = __builtin_amdgcn_readfirstlane(my_it);

    /* everything that steers the lane is wave-uniform and lives in scalar registers:
     * the current segment, and the next one (so the gate inputs of its first block
     * can be prefetched like any other block's) */
    int tile = 0, s = 0, s1 = 0, Tt = 0, boff = 0;
    int n_tile = 0, n_s0 = 0, n_s1 = 0, n_Tt = 0, n_boff = 0;
    bool n_ok = false;
    int myT = 0, n_myT = 0;
    auto fetch_next = [&](int i) {
        n_ok = i < sge;
        if (n_ok) {
            const ShGruSegD sg = L.seg[i];
            n_tile = __builtin_amdgcn_readfirstlane(sg.tile);
            n_s0 = __builtin_amdgcn_readfirstlane(sg.s0);
            n_s1 = __builtin_amdgcn_readfirstlane(sg.s1);
            n_Tt = __builtin_amdgcn_readfirstlane(md.tile_T[n_tile]);
            n_boff = __builtin_amdgcn_readfirstlane((int)md.tile_boff[n_tile]);
            n_myT = md.rT[n_tile * 16 + (lane & 15)];
        }
    };
    auto advance = [&]() { tile = n_tile; s = n_s0; s1 = n_s1; Tt = n_Tt; boff = n_boff; myT = n_myT; };
    f32x4 h = {0.f, 0.f, 0.f, 0.f};
    auto take_over = [&]() {                        /* initial state of the (new) current segment */
        h = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (s > 0) {                                /* continuation of a tile begun on another lane */
            if (!sh_wait_flag(L.flag + tile, (unsigned)NU) && lane == 0)      /* give up loudly instead of hanging the device */
                __hip_atomic_store(L.flag + L.ntile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const float *hs = L.hstate + ((long long)tile * NU + u) * 256 + lane * 4;
#pragma unroll
            for (int k = 0; k < 4; k++) h[k] = __hip_atomic_load(hs + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    /* gate inputs of one block: [update | reset | candidate] rows of this wave's unit tile */
    f32x4 xz = h, xr = h, xh = h;
    auto xload = [&](long long col) {
        const float *p = xaff + col * xstride + lane * 4;
        xz = *(const f32x4 *)(p + u * 256);
        xr = *(const f32x4 *)(p + (NU + u) * 256);
        xh = *(const f32x4 *)(p + (2 * NU + u) * 256);
    };
    if (my_it > 0) {
        fetch_next(sgi);
        advance();
        fetch_next(++sgi);
        take_over();
        *(f32x4 *)(lds_h + u * 256 + lane * 4) = h;
        xload(boff + (backward ? Tt - 1 - s : s));
    }
    __syncthreads();

    unsigned long long g1 = 0, g2 = 0, g3 = 0, g4 = 0, gt0 = 0, gt1;
#define LSTAMP(acc) do { if (STAMP) { gt1 = __builtin_readcyclecounter(); acc += gt1 - gt0; gt0 = gt1; } } while (0)
    if (STAMP) gt0 = __builtin_readcyclecounter();
    int it = 0;
    for (; it < my_it; it++) {
        /* phase 1: reset gate on h, r*h -> LDS */
        f32x4 hb[NU];
#pragma unroll
        for (int mm = 0; mm < NU; mm++) hb[mm] = *(const f32x4 *)(lds_h + mm * 256 + lane * 4);
        f32x4 ar = xr, ar2 = {0.f, 0.f, 0.f, 0.f}, az = xz, ah = xh;
        const int t = backward ? Tt - 1 - s : s;
        {   /* the block this lane works on next: a whole step ahead of its use, never conditional */
            long long ncol = boff + t;
            if (s + 1 < s1) ncol = boff + (backward ? t - 1 : t + 1);
            else if (n_ok) ncol = n_boff + (backward ? n_Tt - 1 - n_s0 : n_s0);
            xload(ncol);
        }
#pragma unroll
        for (int mm = 0; mm < NU; mm++) {
            ar = mfma4(wr[mm * 4 + 0], hb[mm][0], ar);
            ar2 = mfma4(wr[mm * 4 + 1], hb[mm][1], ar2);
            ar = mfma4(wr[mm * 4 + 2], hb[mm][2], ar);
            ar2 = mfma4(wr[mm * 4 + 3], hb[mm][3], ar2);
        }
        ar += ar2;
        const f32x4 rh = d_logistic4(ar) * h;                                      /* layers.c:515 */
        *(f32x4 *)(lds_rh + u * 256 + lane * 4) = rh;
        LSTAMP(g1);
        lds_barrier();
        LSTAMP(g2);
        /* phase 2: update gate on h (still in registers), candidate on r*h, blend, publish */
        f32x4 rb[NU];
#pragma unroll
        for (int mm = 0; mm < NU; mm++) rb[mm] = *(const f32x4 *)(lds_rh + mm * 256 + lane * 4);
        f32x4 az2 = {0.f, 0.f, 0.f, 0.f}, ah2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mm = 0; mm < NU; mm++) {
            az = mfma4(wz[mm * 4 + 0], hb[mm][0], az);
            az2 = mfma4(wz[mm * 4 + 1], hb[mm][1], az2);
            az = mfma4(wz[mm * 4 + 2], hb[mm][2], az);
            az2 = mfma4(wz[mm * 4 + 3], hb[mm][3], az2);
        }
#pragma unroll
        for (int mm = 0; mm < NU; mm++) {
            ah = mfma4(wh[mm * 4 + 0], rb[mm][0], ah);
            ah2 = mfma4(wh[mm * 4 + 1], rb[mm][1], ah2);
            ah = mfma4(wh[mm * 4 + 2], rb[mm][2], ah);
            ah2 = mfma4(wh[mm * 4 + 3], rb[mm][3], ah2);
        }
        az += az2;
        ah += ah2;
        const bool active = t < myT;
        {
            const f32x4 z = d_logistic4(az), hbar = d_tanh4(ah);
            const f32x4 hn = z * h + (1.0f - z) * hbar;                            /* layers.c:525 */
#pragma unroll
            for (int k = 0; k < 4; k++) h[k] = active ? hn[k] : 0.0f;
        }
        f32x4 o = h;
        const long long oidx = ((long long)(boff + t) * NU + u) * 256 + lane * 4;
        if (resid) o += *(const f32x4 *)(resid + oidx);                           /* networks.c:583 */
        *(f32x4 *)(out + oidx) = o;
        s++;
        if (s == s1) {                                       /* segment done */
            if (s1 < Tt) {                                   /* the tile continues on another lane */
                float *hs = L.hstate + ((long long)tile * NU + u) * 256 + lane * 4;
#pragma unroll
                for (int k = 0; k < 4; k++) __hip_atomic_store(hs + k, h[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                if (lane == 0) __hip_atomic_fetch_add(L.flag + tile, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (n_ok) {
                advance();
                fetch_next(++sgi);
                take_over();
            }
        }
        *(f32x4 *)(lds_h + u * 256 + lane * 4) = h;
        LSTAMP(g3);
        lds_barrier();
        LSTAMP(g4);
    }
    for (; it < nit; it++) { lds_barrier(); lds_barrier(); }   /* the other lane of the workgroup is still stepping */
    if (STAMP && dbgbuf && lane == 0) { unsigned long long *d = dbgbuf + ((long long)blockIdx.x * 2 * NU + wave) * 8; d[0] = g1; d[1] = g2; d[2] = g3; d[3] = g4; d[4] = nit; }
}


/* ------------------------------------------------------------------ */
/* G1/G2 as split products (split8 / split_step): the lane-schedule      */
/* recurrence of k_gru_lanes with its three contractions on the bf16     */
/* matrix pipe.  A wave keeps its rows of sW / sW2 as bf16 pieces in      */
/* registers (108 VGPRs for S = 96); h and r*h travel through LDS as      */
/* pieces: the wave that owns unit tile u cuts its four values per lane    */
/* into pieces once and writes them into its half of the k step's          */
/* 8-value slots, every wave reads whole slots (ds_read_b128) as B         */
/* operands.  Per step and wave: 54 MFMAs of 16 cycles instead of 72 of    */
/* 32.  The reset and update gates share the h pieces (phase 1), the       */
/* candidate runs on the r*h pieces after the barrier (phase 2).           */
/* ------------------------------------------------------------------ */
template <int NU>
__global__ __launch_bounds__(128 * NU) void k_gru_split(const float *__restrict__ xaff, float *__restrict__ out,
                                                        const float *__restrict__ resid,
                                                        const unsigned *__restrict__ sWp, const unsigned *__restrict__ sW2p,
                                                        ShMeta md, int backward, ShGruLanes L) {
    static_assert(NU % 2 == 0, "k steps of 32 units");
    constexpr int KS = NU / 2;
    constexpr int PBUF = KS * 2 * 64 * 4;          /* one operand as fp16 pieces, in 32-bit words: [ks][piece][lane][4] */
    extern __shared__ __attribute__((aligned(16))) unsigned ldsw[];   /* [2 lanes][h | rh][PBUF] */
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int u = wave % NU, grp = wave / NU;
    const int ln = blockIdx.x * 2 + grp;

    ShSplit wz[KS], wr[KS], wh[KS];                 /* this wave's rows of sW / sW2, cut into pieces on the host */
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
        wz[ks] = load_pieces(sWp + ((long long)u * KS + ks) * 512, lane);
        wr[ks] = load_pieces(sWp + ((long long)(NU + u) * KS + ks) * 512, lane);
        wh[ks] = load_pieces(sW2p + ((long long)u * KS + ks) * 512, lane);
    }
    unsigned *lds_h = ldsw + grp * 2 * PBUF, *lds_rh = lds_h + PBUF;
    /* this wave's half (u & 1) of k step u / 2: two words per piece */
    const int wofs = (((u >> 1) * 2) * 64 + lane) * 4 + (u & 1) * 2;
    auto publish = [&](unsigned *buf, f32x4 v) {
        unsigned a1, a2, b1, b2;
        split_pair(v[0], v[1], a1, a2);
        split_pair(v[2], v[3], b1, b2);
        *(uint2 *)(buf + wofs) = make_uint2(a1, b1);
        *(uint2 *)(buf + wofs + 256) = make_uint2(a2, b2);
    };
    auto pieces = [&](const unsigned *buf, int ks) {
        return load_pieces(buf + ks * 512, lane);
    };
    const long long xstride = 3LL * NU * 256;
    const int nit = L.wg_iter[blockIdx.x];
    int sgi = __builtin_amdgcn_readfirstlane(L.lane_off[ln]);
    const int sge = __builtin_amdgcn_readfirstlane(L.lane_off[ln + 1]);
    int my_it = 0;                                  /* steps of this lane; it idles (barriers only) afterwards */
    for (int i = sgi; i < sge; i++) my_it += L.seg[i].s1 - L.seg[i].s0;
    my_it = __builtin_amdgcn_readfirstlane(my_it);

    /* lane state: wave-uniform, in scalar registers (see k_gru_lanes) */
    int tile = 0, s = 0, s1 = 0, Tt = 0, boff = 0;
    int n_tile = 0, n_s0 = 0, n_s1 = 0, n_Tt = 0, n_boff = 0;
    bool n_ok = false;
    int myT = 0, n_myT = 0;
    auto fetch_next = [&](int i) {
        n_ok = i < sge;
        if (n_ok) {
            const ShGruSegD sg = L.seg[i];
            n_tile = __builtin_amdgcn_readfirstlane(sg.tile);
            n_s0 = __builtin_amdgcn_readfirstlane(sg.s0);
            n_s1 = __builtin_amdgcn_readfirstlane(sg.s1);
            n_Tt = __builtin_amdgcn_readfirstlane(md.tile_T[n_tile]);
            n_boff = __builtin_amdgcn_readfirstlane((int)md.tile_boff[n_tile]);
            n_myT = md.rT[n_tile * 16 + (lane & 15)];
        }
    };
    auto advance = [&]() { tile = n_tile; s = n_s0; s1 = n_s1; Tt = n_Tt; boff = n_boff; myT = n_myT; };
    f32x4 h = {0.f, 0.f, 0.f, 0.f};
    auto take_over = [&]() {                        /* initial state of the (new) current segment */
        h = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (s > 0) {                                /* continuation of a tile begun on another lane */
            if (!sh_wait_flag(L.flag + tile, (unsigned)NU) && lane == 0)      /* give up loudly instead of hanging the device */
                __hip_atomic_store(L.flag + L.ntile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const float *hs = L.hstate + ((long long)tile * NU + u) * 256 + lane * 4;
#pragma unroll
            for (int k = 0; k < 4; k++) h[k] = __hip_atomic_load(hs + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    /* gate inputs of one block: [update | reset | candidate] rows of this wave's unit tile */
    f32x4 xz = h, xr = h, xh = h;
    auto xload = [&](long long col) {
        const float *p = xaff + col * xstride + lane * 4;
        xz = *(const f32x4 *)(p + u * 256);
        xr = *(const f32x4 *)(p + (NU + u) * 256);
        xh = *(const f32x4 *)(p + (2 * NU + u) * 256);
    };
    if (my_it > 0) {
        fetch_next(sgi);
        advance();
        fetch_next(++sgi);
        take_over();
        publish(lds_h, h);
        xload(boff + (backward ? Tt - 1 - s : s));
    }
    __syncthreads();

    int it = 0;
    for (; it < my_it; it++) {
        /* phase 1: reset and update gates on the h pieces; r*h -> LDS */
        f32x4 ar = xr * SH_OSCALE, az = xz * SH_OSCALE, ah = xh * SH_OSCALE;    /* accumulator units (exact: the projection's own bits) */
        const int t = backward ? Tt - 1 - s : s;
        {   /* the block this lane works on next: a whole step ahead of its use, never conditional */
            long long ncol = boff + t;
            if (s + 1 < s1) ncol = boff + (backward ? t - 1 : t + 1);
            else if (n_ok) ncol = n_boff + (backward ? n_Tt - 1 - n_s0 : n_s0);
            xload(ncol);
        }
        {
            ShSplit hp[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ks++) hp[ks] = pieces(lds_h, ks);
            split_dot2<KS>(wr, wz, hp, ar, az);
        }
        publish(lds_rh, d_logistic4_acc(ar) * h);                                  /* layers.c:515 */
        const f32x4 z = d_logistic4_acc(az);
        lds_barrier();
        /* phase 2: candidate on the r*h pieces, blend, publish */
        {
            ShSplit rp[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ks++) rp[ks] = pieces(lds_rh, ks);
            ah = split_dot<KS>(wh, rp, ah);
        }
        const bool active = t < myT;
        {
            const f32x4 hbar = d_tanh4_acc(ah);
            const f32x4 hn = z * h + (1.0f - z) * hbar;                            /* layers.c:525 */
#pragma unroll
            for (int k = 0; k < 4; k++) h[k] = active ? hn[k] : 0.0f;
        }
        f32x4 o = h;
        const long long oidx = ((long long)(boff + t) * NU + u) * 256 + lane * 4;
        if (resid) o += *(const f32x4 *)(resid + oidx);                           /* networks.c:583 */
        *(f32x4 *)(out + oidx) = o;
        s++;
        if (s == s1) {                                       /* segment done */
            if (s1 < Tt) {                                   /* the tile continues on another lane */
                float *hs = L.hstate + ((long long)tile * NU + u) * 256 + lane * 4;
#pragma unroll
                for (int k = 0; k < 4; k++) __hip_atomic_store(hs + k, h[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                if (lane == 0) __hip_atomic_fetch_add(L.flag + tile, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (n_ok) {
                advance();
                fetch_next(++sgi);
                take_over();
            }
        }
        publish(lds_h, h);
        lds_barrier();
    }
    for (; it < nit; it++) { lds_barrier(); lds_barrier(); }   /* the other lane of the workgroup is still stepping */
}

/* ------------------------------------------------------------------ */
/* L1 + G1/G2 in one kernel, split products throughout: a workgroup runs   */
/* NT lanes of the schedule (NT tiles of 16 reads at a time) on two teams  */
/* of S/16 waves.  The projection team turns the layer's input column of   */
/* the NEXT step into that step's gate inputs (wave u: the update / reset  */
/* / candidate rows of unit tile u, its rows of iW as fp16 pieces in        */
/* registers) and leaves them in LDS; the recurrence team (k_gru_split's    */
/* step) takes them from there.  The 3S gate inputs per read per block --   */
/* 9.2 GB per layer and direction at 10 000 reads -- never exist in HBM: a   */
/* layer reads S and writes S floats per read per block.  Both teams keep   */
/* the same two barriers per step:                                          */
/*   interval A   recurrence: reset + update gates, r*h -> LDS              */
/*                projection: candidate rows of the next block              */
/*   interval B   recurrence: candidate, blend, h -> LDS, h -> HBM          */
/*                projection: update + reset rows -> x ring, next input     */
/*                chunk -> pieces                                           */
/* With NT = 2 every wave steps two independent tiles inside each interval:  */
/* tile 1's MFMAs are in flight while tile 0's gate activations issue (and   */
/* the other way round in the next interval), so the matrix pipe and the     */
/* VALU overlap within a wave instead of taking turns, and the LDS / barrier  */
/* latencies of a step are paid once for two tiles.  A tile's arithmetic is   */
/* the same for every NT: results do not depend on it.                       */
/* The input column travels through LDS as pieces exactly like h: each       */
/* projection wave fetches and cuts the chunk of its own unit tile.          */
/* ------------------------------------------------------------------ */
struct ShLaneCursor {          /* walks a lane's segments step by step; everything wave-uniform */
    int sgi, sge;
    int tile, s, s1, Tt, boff;
    bool ok;
};

#ifndef SH_REC_PRIO
#define SH_REC_PRIO 0       /* s_setprio of the recurrence team (projection stays at 0) */
#endif
#ifndef SH_PDELAY_A
#define SH_PDELAY_A 0       /* s_sleep argument in front of the projection team's MFMAs of interval A / B (0: none) */
#endif
#ifndef SH_PDELAY_B
#define SH_PDELAY_B 0
#endif
#ifndef SH_PROJ_PRIO
#define SH_PROJ_PRIO 0      /* s_setprio of the projection team */
#endif
#ifndef SH_RFIRST
#define SH_RFIRST 1         /* 1 (measured -2.6 %): interval A issues the reset-gate products of all tiles first and publishes r*h before the update gate's results are looked at */
#endif
#ifndef SH_PROJ_VALU_FIRST
#define SH_PROJ_VALU_FIRST 1   /* interval B: publish / fetch before the update + reset rows (measured -2 %) instead of after */
#endif
#ifndef SH_ABL
#define SH_ABL 0            /* timing ablations of k_gru_proj (tools/ab.sh); results are invalid unless 0 */
#endif
__device__ __forceinline__ f32x4 abl_logistic4(f32x4 a) { return (SH_ABL & 1) ? a * (0.25f * SH_OINV) + 0.5f : d_logistic4_acc(a); }
__device__ __forceinline__ f32x4 abl_tanh4(f32x4 a) { return (SH_ABL & 1) ? a * (0.5f * SH_OINV) : d_tanh4_acc(a); }

template <int NU, int NT, bool RESID, bool STAMP = false>
__global__ __launch_bounds__(128 * NU) void k_gru_proj(const float *__restrict__ in, float *__restrict__ out,
                                                       const float *__restrict__ resid,
                                                       const unsigned *__restrict__ iWp, const float *__restrict__ ibfrag,
                                                       const unsigned *__restrict__ sWp, const unsigned *__restrict__ sW2p,
                                                       ShMeta md, int backward, ShGruLanes L,
                                                       unsigned long long *dbg = nullptr) {
    static_assert(NU % 2 == 0, "k steps of 32 units");
    constexpr int KS = NU / 2;
    constexpr int PBUF = KS * 2 * 64 * 4;          /* one operand as fp16 pieces, in 32-bit words: [ks][piece][lane][4] */
    constexpr int XBUF = 3 * NU * 256;             /* one block's gate inputs, accumulator layout [gate][u][lane][4] */
    constexpr int TBUF = 4 * PBUF + 2 * XBUF;      /* words per tile slot: h | r*h | in[2] | x[2] */
    unsigned long long pa = 0, pb = 0, pc = 0, pd = 0, pt0 = 0, pt1;
    unsigned long long q1 = 0, q2 = 0, q3 = 0, q4 = 0, q5 = 0, qt0 = 0, qt1;      /* finer marks inside the recurrence team's interval B */
#define QSTAMP(acc) do { if (STAMP) { qt1 = __builtin_readcyclecounter(); acc += qt1 - qt0; qt0 = qt1; } } while (0)
#define PSTAMP(acc) do { if (STAMP) { pt1 = __builtin_readcyclecounter(); acc += pt1 - pt0; pt0 = pt1; } } while (0)
#define PDUMP() do { if (STAMP && dbg && lane == 0) { unsigned long long *d_ = dbg + ((long long)blockIdx.x * 2 * NU + wave) * 16; d_[0] = pa; d_[1] = pb; d_[2] = pc; d_[3] = pd; d_[4] = nit; d_[5] = q1; d_[6] = q2; d_[7] = q3; d_[8] = q4; d_[9] = q5; } } while (0)
    extern __shared__ __attribute__((aligned(16))) unsigned ldsw[];
    const int lane = threadIdx.x & 63;
    /* global accesses as (uniform 64-bit base in scalar registers) + (this 32-bit lane offset).  The base is made
     * opaque (else the compiler re-associates to (pointer + lane offset) + uniform, hoists that 64-bit VGPR pair
     * out of the step loop and -- in the residual variant -- spills it: a scratch reload and a vmcnt(0) per step) */
    const unsigned lofs = (unsigned)lane * 4u;
    typedef __attribute__((address_space(1))) float *gf32;
    typedef __attribute__((address_space(1))) f32x4 *gf32x4;
    auto gload = [&](const float *base) { gf32 b = (gf32)base; asm volatile("" : "+s"(b)); return *(gf32x4)(b + lofs); };
    auto gstore = [&](float *base, f32x4 v) { gf32 b = (gf32)base; asm volatile("" : "+s"(b)); *(gf32x4)(b + lofs) = v; };
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool rec = wave < NU;
    const int u = rec ? wave : wave - NU;

    /* this wave's three m-tiles as pieces (cut on the host): rows of sW / sW2 (recurrence) or of iW (projection) */
    ShSplit w0[KS], w1[KS], w2[KS];
    {
        const unsigned *f0 = rec ? sWp + (long long)u * KS * 512 : iWp + (long long)u * KS * 512;                    /* update */
        const unsigned *f1 = rec ? sWp + (long long)(NU + u) * KS * 512 : iWp + (long long)(NU + u) * KS * 512;      /* reset */
        const unsigned *f2 = rec ? sW2p + (long long)u * KS * 512 : iWp + (long long)(2 * NU + u) * KS * 512;        /* candidate */
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            w0[ks] = load_pieces(f0 + ks * 512, lane);
            w1[ks] = load_pieces(f1 + ks * 512, lane);
            w2[ks] = load_pieces(f2 + ks * 512, lane);
        }
        /* wait for the weights HERE, once: left to itself the compiler waits at their first use inside the step
         * loop, with a count that also covers the previous step's output store -- on every step */
#pragma unroll
        for (int ks = 0; ks < KS; ks++)
            asm volatile("" : "+v"(w0[ks].p1), "+v"(w0[ks].p2), "+v"(w1[ks].p1), "+v"(w1[ks].p2), "+v"(w2[ks].p1), "+v"(w2[ks].p2));
    }
    const int wofs = (((u >> 1) * 2) * 64 + lane) * 4 + (u & 1) * 2;
    auto publish = [&](unsigned *buf, f32x4 v) {
        unsigned a1, a2, b1, b2;
        if (SH_ABL & 4) { a1 = __float_as_uint(v[0]); a2 = __float_as_uint(v[1]); b1 = __float_as_uint(v[2]); b2 = __float_as_uint(v[3]); }
        else { split_pair(v[0], v[1], a1, a2); split_pair(v[2], v[3], b1, b2); }
        *(uint2 *)(buf + wofs) = make_uint2(a1, b1);
        *(uint2 *)(buf + wofs + 256) = make_uint2(a2, b2);
    };
    auto pieces = [&](const unsigned *buf, int ks) { return load_pieces(buf + ks * 512, lane); };
    auto lds_h = [&](int tl) { return ldsw + tl * TBUF; };
    auto lds_rh = [&](int tl) { return ldsw + tl * TBUF + PBUF; };
    auto lds_in = [&](int tl, int par) { return ldsw + tl * TBUF + (2 + par) * PBUF; };
    auto lds_x = [&](int tl, int par) { return (float *)(ldsw + tl * TBUF + 4 * PBUF + par * XBUF); };

    ShLaneCursor c[NT] = {};
    int my_it[NT], nit = 0;
#pragma unroll
    for (int tl = 0; tl < NT; tl++) {
        const int ln = blockIdx.x * NT + tl;
        c[tl].sgi = __builtin_amdgcn_readfirstlane(L.lane_off[ln]);
        c[tl].sge = __builtin_amdgcn_readfirstlane(L.lane_off[ln + 1]);
        int n = 0;
        for (int i = c[tl].sgi; i < c[tl].sge; i++) n += L.seg[i].s1 - L.seg[i].s0;
        my_it[tl] = __builtin_amdgcn_readfirstlane(n);
        nit = max(nit, my_it[tl]);
    }
    if (nit == 0) return;                                     /* (uniform over the workgroup) */
    auto enter = [&](ShLaneCursor &cc) {                      /* make segment cc.sgi current */
        cc.ok = cc.sgi < cc.sge;
        if (cc.ok) {
            const ShGruSegD sg = L.seg[cc.sgi];
            cc.tile = __builtin_amdgcn_readfirstlane(sg.tile);
            cc.s = __builtin_amdgcn_readfirstlane(sg.s0);
            cc.s1 = __builtin_amdgcn_readfirstlane(sg.s1);
            cc.Tt = __builtin_amdgcn_readfirstlane(md.tile_T[cc.tile]);
            cc.boff = __builtin_amdgcn_readfirstlane((int)md.tile_boff[cc.tile]);
        }
    };
    auto column = [&](const ShLaneCursor &cc) { return (long long)cc.boff + (backward ? cc.Tt - 1 - cc.s : cc.s); };

    if (!rec) {
        /* ---------------- projection team: one block ahead of the recurrence ---------------- */
        if (SH_PROJ_PRIO) __builtin_amdgcn_s_setprio(SH_PROJ_PRIO);
        f32x4 bz = *(const f32x4 *)(ibfrag + (u * 64 + lane) * 4);
        f32x4 br = *(const f32x4 *)(ibfrag + ((NU + u) * 64 + lane) * 4);
        asm volatile("" : "+v"(bz), "+v"(br));
        f32x4 bh = *(const f32x4 *)(ibfrag + ((2 * NU + u) * 64 + lane) * 4);
        asm volatile("" : "+v"(bh));
        /* the input chunk of a block is fetched three blocks before it is cut into pieces (a step is about
         * as long as an HBM access): a queue of two in registers behind the one in use */
        /* (the load itself is unconditional -- past the end of the lane it re-reads the layer's first chunk -- so
         * that the number of loads in flight is the same on every path and the compiler can wait for exactly the
         * oldest one instead of for all of them) */
        auto fetch = [&](ShLaneCursor &cc) {
            const long long col = cc.ok ? column(cc) : 0;
            const f32x4 v = gload(in + (col * NU + u) * 256);
            if (cc.ok) {
                cc.s++;
                if (cc.s == cc.s1) { cc.sgi++; enter(cc); }
            }
            return v;
        };
        f32x4 xq1[NT], xq2[NT], ah[NT];
        /* a block's 27 MFMAs: the candidate rows (9) in interval A, where the recurrence team issues 18 per
         * wave and tile, the update and reset rows (18) in interval B, where it issues 9 */
        /* (the affine kernels' order: bit-identical to them; the gate inputs stay in accumulator units) */
        auto project_h = [&](const unsigned *ibuf, f32x4 &dst) {
            ShSplit ip[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ks++) ip[ks] = pieces(ibuf, ks);
            dst = (SH_ABL & 8) ? bh : split_dot<KS>(w2, ip, bh);
        };
        auto project_zr = [&](const unsigned *ibuf, float *xdst, f32x4 hv) {
            f32x4 cz = bz, cr = br;
            if (!(SH_ABL & 8)) {
                ShSplit ip[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ks++) ip[ks] = pieces(ibuf, ks);
                split_dot2<KS>(w0, w1, ip, cz, cr);
            }
            *(f32x4 *)(xdst + (u * 64 + lane) * 4) = cz;
            *(f32x4 *)(xdst + ((NU + u) * 64 + lane) * 4) = cr;
            *(f32x4 *)(xdst + ((2 * NU + u) * 64 + lane) * 4) = hv;
        };
        /* prologue: block 0's gate inputs, block 1 as pieces */
#pragma unroll
        for (int tl = 0; tl < NT; tl++) {
            enter(c[tl]);
            const f32x4 xin = fetch(c[tl]);
            xq1[tl] = fetch(c[tl]); xq2[tl] = fetch(c[tl]);
            publish(lds_in(tl, 0), xin);
        }
        lds_barrier();
#pragma unroll
        for (int tl = 0; tl < NT; tl++) {
            project_h(lds_in(tl, 0), ah[tl]);
            project_zr(lds_in(tl, 0), lds_x(tl, 0), ah[tl]);
            publish(lds_in(tl, 1), xq1[tl]);
            xq1[tl] = xq2[tl];
            xq2[tl] = fetch(c[tl]);
        }
        lds_barrier();
        if (STAMP) pt0 = __builtin_readcyclecounter();
        for (int it = 0; it < nit; it++) {
            const int np = (it + 1) & 1;
            if (SH_PDELAY_A) __builtin_amdgcn_s_sleep(SH_PDELAY_A);
#pragma unroll
            for (int tl = 0; tl < NT; tl++) project_h(lds_in(tl, np), ah[tl]);                 /* interval A: block it + 1 */
            PSTAMP(pa);
            lds_barrier();
            PSTAMP(pb);
            if (SH_PROJ_VALU_FIRST) {
#pragma unroll
                for (int tl = 0; tl < NT; tl++) {
                    publish(lds_in(tl, it & 1), xq1[tl]);                                      /* block it + 2 as pieces */
                    xq1[tl] = xq2[tl];
                    xq2[tl] = fetch(c[tl]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (SH_PDELAY_B) __builtin_amdgcn_s_sleep(SH_PDELAY_B);
#pragma unroll
            for (int tl = 0; tl < NT; tl++) project_zr(lds_in(tl, np), lds_x(tl, np), ah[tl]);  /* interval B */
            if (!SH_PROJ_VALU_FIRST) {
#pragma unroll
                for (int tl = 0; tl < NT; tl++) {
                    publish(lds_in(tl, it & 1), xq1[tl]);                                      /* block it + 2 as pieces */
                    xq1[tl] = xq2[tl];
                    xq2[tl] = fetch(c[tl]);
                }
            }
            PSTAMP(pc);
            lds_barrier();
            PSTAMP(pd);
        }
        PDUMP();
        return;
    }

    /* ---------------- recurrence team ---------------- */
    if (SH_REC_PRIO) __builtin_amdgcn_s_setprio(SH_REC_PRIO);
    /* block counts of this lane's reads; with two tile slots 16 bits each in one register (the residual variant
     * is one VGPR short of keeping its step loop free of scratch otherwise; the host schedules two tiles per
     * workgroup only when no tile has 65536 blocks or more) */
    static_assert(NT <= 2, "two block counts per register");
    unsigned myT2 = 0;
    f32x4 h[NT];
    auto take_over = [&](int tl) {                  /* initial state of lane tl's (new) current segment */
        h[tl] = (f32x4){0.f, 0.f, 0.f, 0.f};
        int mt = 0;
        if (c[tl].ok) mt = md.rT[c[tl].tile * 16 + (lane & 15)];
        if (NT == 1) myT2 = (unsigned)mt;           /* (the host uses two tiles per workgroup only below 65536 blocks per tile) */
        else myT2 = tl ? ((myT2 & 0xffffu) | ((unsigned)mt << 16)) : ((myT2 & 0xffff0000u) | ((unsigned)mt & 0xffffu));
        if (!c[tl].ok) return;
        if (c[tl].s > 0) {                          /* continuation of a tile begun on another lane */
            if (!sh_wait_flag(L.flag + c[tl].tile, (unsigned)NU) && lane == 0)      /* give up loudly instead of hanging the device */
                __hip_atomic_store(L.flag + L.ntile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const float *hs = L.hstate + ((long long)c[tl].tile * NU + u) * 256 + lane * 4;
#pragma unroll
            for (int k = 0; k < 4; k++) h[tl][k] = __hip_atomic_load(hs + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        /* the values loaded on this (rare) path are consumed HERE: otherwise the compiler waits for them where the
         * paths join -- a wait for every vector memory operation in flight, the step's output store included, on
         * every step */
        asm volatile("" : "+v"(myT2), "+v"(h[tl][0]), "+v"(h[tl][1]), "+v"(h[tl][2]), "+v"(h[tl][3]));
    };
#pragma unroll
    for (int tl = 0; tl < NT; tl++) {
        enter(c[tl]);
        take_over(tl);
        publish(lds_h(tl), h[tl]);
    }
    lds_barrier();                                  /* (prologue of the projection team) */
    lds_barrier();
    if (STAMP) pt0 = __builtin_readcyclecounter();
    /* rnnrf (networks.c:583): the layer's input column is added to its output; fetched a step ahead */
    f32x4 rs[NT];
    auto resid_fetch = [&](int tl) {
        const long long col = c[tl].ok ? column(c[tl]) : 0;
        rs[tl] = gload(resid + (col * NU + u) * 256);
    };
    if (RESID) {
#pragma unroll
        for (int tl = 0; tl < NT; tl++) resid_fetch(tl);
    }
    for (int it = 0; it < nit; it++) {
        const int par = it & 1;
        /* interval A: reset and update gates on the h pieces; r*h -> LDS.  All tiles' MFMAs first, then the
         * activations: tile 1's products are in flight while tile 0's logistic issues */
        f32x4 cr[NT], cz[NT];
        f32x4 z[NT];
        if (SH_RFIRST) {
            /* the reset gate is what the other waves wait for: its products go first, r*h is published as soon as
             * they are in, and the update gate's products (issued behind them, needed only for the blend) complete
             * while this wave is at the barrier and beyond */
            ShSplit hp[NT][KS];
#pragma unroll
            for (int tl = 0; tl < NT; tl++) {
                const float *xs = lds_x(tl, par);
                cz[tl] = *(const f32x4 *)(xs + (u * 64 + lane) * 4);
                cr[tl] = *(const f32x4 *)(xs + ((NU + u) * 64 + lane) * 4);
#pragma unroll
                for (int ks = 0; ks < KS; ks++) hp[tl][ks] = pieces(lds_h(tl), ks);
                cr[tl] = split_dot<KS>(w1, hp[tl], cr[tl]);
            }
#pragma unroll
            for (int tl = 0; tl < NT; tl++) cz[tl] = split_dot<KS>(w0, hp[tl], cz[tl]);
#pragma unroll
            for (int tl = 0; tl < NT; tl++) publish(lds_rh(tl), abl_logistic4(cr[tl]) * h[tl]);      /* layers.c:515 */
        } else {
#pragma unroll
        for (int tl = 0; tl < NT; tl++) {
            const float *xs = lds_x(tl, par);
            cz[tl] = *(const f32x4 *)(xs + (u * 64 + lane) * 4);
            cr[tl] = *(const f32x4 *)(xs + ((NU + u) * 64 + lane) * 4);
            ShSplit hp[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ks++) hp[ks] = pieces(lds_h(tl), ks);
            split_dot2<KS>(w1, w0, hp, cr[tl], cz[tl]);
        }
#pragma unroll
        for (int tl = 0; tl < NT; tl++) {
            publish(lds_rh(tl), abl_logistic4(cr[tl]) * h[tl]);                       /* layers.c:515 */
            z[tl] = abl_logistic4(cz[tl]);
        }
        }
        PSTAMP(pa);
        lds_barrier();
        PSTAMP(pb);
        /* interval B: candidate on the r*h pieces, blend, publish */
        if (STAMP) qt0 = __builtin_readcyclecounter();
        f32x4 ch[NT];
#pragma unroll
        for (int tl = 0; tl < NT; tl++) {
            ShSplit rp[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ks++) rp[ks] = pieces(lds_rh(tl), ks);
            ch[tl] = split_dot<KS>(w2, rp, *(const f32x4 *)(lds_x(tl, par) + ((2 * NU + u) * 64 + lane) * 4));
            if (STAMP) __builtin_amdgcn_sched_barrier(0);
            QSTAMP(q1);                                        /* both tiles: LDS reads back + candidate MFMAs issued */
        }
        if (SH_RFIRST) {
#pragma unroll
            for (int tl = 0; tl < NT; tl++) z[tl] = abl_logistic4(cz[tl]);
        }
        if (STAMP) { asm volatile("" :: "v"(z[0]), "v"(z[NT - 1])); __builtin_amdgcn_sched_barrier(0); }
        QSTAMP(q2);                                            /* update-gate logistic */
#pragma unroll
        for (int tl = 0; tl < NT; tl++) {
            const bool live = it < my_it[tl];                                          /* (wave-uniform) */
            const int t = backward ? c[tl].Tt - 1 - c[tl].s : c[tl].s;
            const bool active = t < (int)(NT == 1 ? myT2 : (tl ? (myT2 >> 16) : (myT2 & 0xffffu)));
            {
                const f32x4 hbar = abl_tanh4(ch[tl]);
                const f32x4 hn = z[tl] * h[tl] + (1.0f - z[tl]) * hbar;                /* layers.c:525 */
#pragma unroll
                for (int k = 0; k < 4; k++) h[tl][k] = active ? hn[k] : 0.0f;
            }
            if (STAMP) { asm volatile("" :: "v"(h[tl])); __builtin_amdgcn_sched_barrier(0); }
            QSTAMP(q3);                                        /* tanh + blend (waits for the candidate's MFMAs) */
            if (live) {
                f32x4 o = h[tl];
                const long long oidx = ((long long)(c[tl].boff + t) * NU + u) * 256;       /* uniform */
                if (RESID) o += rs[tl];                                               /* networks.c:583 */
                if (!(SH_ABL & 2)) gstore(out + oidx, o);
                c[tl].s++;
                if (c[tl].s == c[tl].s1) {                           /* segment done */
                    if (c[tl].s1 < c[tl].Tt) {                       /* the tile continues on another lane */
                        float *hs = L.hstate + ((long long)c[tl].tile * NU + u) * 256 + lane * 4;
#pragma unroll
                        for (int k = 0; k < 4; k++) __hip_atomic_store(hs + k, h[tl][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                        if (lane == 0) __hip_atomic_fetch_add(L.flag + c[tl].tile, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    c[tl].sgi++;
                    enter(c[tl]);
                    take_over(tl);
                }
            }
            if (STAMP) __builtin_amdgcn_sched_barrier(0);
            QSTAMP(q4);                                        /* output store, lane bookkeeping */
            if (RESID) resid_fetch(tl);                                                /* the next step's column */
            publish(lds_h(tl), h[tl]);
            if (STAMP) __builtin_amdgcn_sched_barrier(0);
            QSTAMP(q5);                                        /* cut into pieces + LDS write */
        }
        PSTAMP(pc);
        lds_barrier();
        PSTAMP(pd);
    }
    PDUMP();
#undef PSTAMP
#undef PDUMP
#undef QSTAMP
}

/* ------------------------------------------------------------------ */
/* (f).4  events: feature columns -> chunk layout, and the peephole LSTM  */
/* ------------------------------------------------------------------ */
/* feature3 columns (12 floats per event: networks.c:155-157) of the reads of a tile into one
 * 16-unit chunk per column block (units 12..15 zero; the weights are padded to match) */
__global__ __launch_bounds__(256) void k_feat_in(const float *__restrict__ feat, ShMeta md, int nfeat,
                                                 float *__restrict__ act, long long ncb_total) {
    const int tile = blockIdx.x;
    const int Tt = md.tile_T[tile];
    const long long boff = md.tile_boff[tile];
    const int lane = threadIdx.x & 63, b = lane & 15, q = lane >> 4;
    const int rd = tile * 16 + b;
    const int myT = md.rT[rd];
    const unsigned long long off = md.sig_off[rd];
    for (int t = blockIdx.y * 4 + (threadIdx.x >> 6); t < Tt; t += gridDim.y * 4) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (t < myT) {
#pragma unroll
            for (int k = 0; k < 4; k++) if (4 * q + k < nfeat) v[k] = feat[off + (unsigned long long)t * nfeat + 4 * q + k];
        }
        *(f32x4 *)(act + (boff + t) * 256 + lane * 4) = v;
    }
}

/* lstm_forward / lstm_backward / lstm_step (layers.c:673-832) for a tile of 16 reads, its four gate
 * contractions as split products (round 1: exact-fp32 MFMAs, 96 of 32 cycles per step and wave; now 36 of 16).
 * Two lanes of NU waves per workgroup as in k_gru_split; wave u owns unit tile u of all four gates (its rows of
 * sW as fp16 pieces: 96 VGPRs for S = 96), so the cell state never leaves its registers and only the output h is
 * exchanged, through LDS as pieces (double buffered: one barrier per step).  Gate pre-activations
 * [input | update | forget | output] arrive as accumulator initial values.  Lane schedule and state hand-over
 * as in k_gru_split (the hand-over carries h and the cell state). */
template <int NU>
__global__ __launch_bounds__(128 * NU) void k_lstm_lanes(const float *__restrict__ xaff, float *__restrict__ out,
                                                        const unsigned *__restrict__ sWp,
                                                        const float *__restrict__ pfrag, ShMeta md,
                                                        int backward, ShGruLanes L) {
    static_assert(NU % 2 == 0, "k steps of 32 units");
    constexpr int KS = NU / 2;
    constexpr int PBUF = KS * 2 * 64 * 4;          /* h as fp16 pieces, in 32-bit words: [ks][piece][lane][4] */
    __shared__ __attribute__((aligned(16))) unsigned lds[2 * 2 * PBUF];      /* [lane][parity][PBUF] */
    /* peepholes: read back from LDS each step (three ds_read_b128) rather than held in 12 VGPRs the weights need */
    __shared__ __attribute__((aligned(16))) float peep[3 * NU * 256];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int u = wave % NU, grp = wave / NU;
    const int ln = blockIdx.x * 2 + grp;

    ShSplit wi[KS], wu[KS], wf[KS], wo[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
        wi[ks] = load_pieces(sWp + ((long long)u * KS + ks) * 512, lane);
        wu[ks] = load_pieces(sWp + ((long long)(NU + u) * KS + ks) * 512, lane);
        wf[ks] = load_pieces(sWp + ((long long)(2 * NU + u) * KS + ks) * 512, lane);
        wo[ks] = load_pieces(sWp + ((long long)(3 * NU + u) * KS + ks) * 512, lane);
    }
    if (grp == 0) {
#pragma unroll
        for (int g = 0; g < 3; g++)
            *(f32x4 *)(peep + ((g * NU + u) * 64 + lane) * 4) = *(const f32x4 *)(pfrag + ((long long)(g * NU + u) * 64 + lane) * 4);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ks++)       /* the weights are waited for here, once (see k_gru_proj) */
        asm volatile("" : "+v"(wi[ks].p1), "+v"(wi[ks].p2), "+v"(wu[ks].p1), "+v"(wu[ks].p2), "+v"(wf[ks].p1), "+v"(wf[ks].p2), "+v"(wo[ks].p1), "+v"(wo[ks].p2));
    unsigned *lds_h = lds + grp * 2 * PBUF;
    const int wofs = (((u >> 1) * 2) * 64 + lane) * 4 + (u & 1) * 2;
    auto publish = [&](unsigned *buf, f32x4 v) {
        unsigned a1, a2, b1, b2;
        split_pair(v[0], v[1], a1, a2);
        split_pair(v[2], v[3], b1, b2);
        *(uint2 *)(buf + wofs) = make_uint2(a1, b1);
        *(uint2 *)(buf + wofs + 256) = make_uint2(a2, b2);
    };
    const long long xstride = 4LL * NU * 256;
    const int nit = L.wg_iter[blockIdx.x];
    int sgi = __builtin_amdgcn_readfirstlane(L.lane_off[ln]);
    const int sge = __builtin_amdgcn_readfirstlane(L.lane_off[ln + 1]);
    int my_it = 0;
    for (int i = sgi; i < sge; i++) my_it += L.seg[i].s1 - L.seg[i].s0;
    my_it = __builtin_amdgcn_readfirstlane(my_it);

    /* lane state in scalar registers: current segment and the next one (k_gru_lanes) */
    int tile = 0, s = 0, s1 = 0, Tt = 0, boff = 0;
    int n_tile = 0, n_s0 = 0, n_s1 = 0, n_Tt = 0, n_boff = 0;
    bool n_ok = false;
    int myT = 0, n_myT = 0;
    auto fetch_next = [&](int i) {
        n_ok = i < sge;
        if (n_ok) {
            const ShGruSegD sg = L.seg[i];
            n_tile = __builtin_amdgcn_readfirstlane(sg.tile);
            n_s0 = __builtin_amdgcn_readfirstlane(sg.s0);
            n_s1 = __builtin_amdgcn_readfirstlane(sg.s1);
            n_Tt = __builtin_amdgcn_readfirstlane(md.tile_T[n_tile]);
            n_boff = __builtin_amdgcn_readfirstlane((int)md.tile_boff[n_tile]);
            n_myT = md.rT[n_tile * 16 + (lane & 15)];
            asm volatile("" : "+v"(n_myT));
        }
    };
    auto advance = [&]() { tile = n_tile; s = n_s0; s1 = n_s1; Tt = n_Tt; boff = n_boff; myT = n_myT; };
    f32x4 h = {0.f, 0.f, 0.f, 0.f}, c = h;
    auto take_over = [&]() {                        /* initial h and cell state of the (new) current segment */
        h = (f32x4){0.f, 0.f, 0.f, 0.f}; c = h;
        if (s > 0) {                                /* continuation of a tile begun on another lane */
            if (!sh_wait_flag(L.flag + tile, (unsigned)NU) && lane == 0)      /* give up loudly instead of hanging the device */
                __hip_atomic_store(L.flag + L.ntile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const float *hs = L.hstate + ((long long)tile * 2 * NU + u) * 256 + lane * 4;       /* [h | c] */
#pragma unroll
            for (int k = 0; k < 4; k++) {
                h[k] = __hip_atomic_load(hs + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                c[k] = __hip_atomic_load(hs + NU * 256 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("" : "+v"(h), "+v"(c));    /* consumed here, not where the paths join (see k_gru_proj) */
        }
    };
    f32x4 xi = h, xu = h, xf = h, xo = h;
    auto xload = [&](long long col) {
        const float *p = xaff + col * xstride + lane * 4;
        xi = *(const f32x4 *)(p + u * 256);
        xu = *(const f32x4 *)(p + (NU + u) * 256);
        xf = *(const f32x4 *)(p + (2 * NU + u) * 256);
        xo = *(const f32x4 *)(p + (3 * NU + u) * 256);
    };
    int par = 0;
    if (my_it > 0) {
        fetch_next(sgi);
        advance();
        fetch_next(++sgi);
        take_over();
        publish(lds_h + par * PBUF, h);
        xload(boff + (backward ? Tt - 1 - s : s));
    }
    __syncthreads();

    int it = 0;
    for (; it < my_it; it++) {
        const int t = backward ? Tt - 1 - s : s;
        ShSplit hp[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ks++) hp[ks] = load_pieces(lds_h + par * PBUF + ks * 512, lane);
        /* the gate inputs in accumulator units (exact: a power of two) */
        f32x4 ai = xi * SH_OSCALE, au = xu * SH_OSCALE, af = xf * SH_OSCALE, ao = xo * SH_OSCALE;
        {   /* the block this lane works on next: a whole step ahead, never conditional */
            long long ncol = boff + t;
            if (s + 1 < s1) ncol = boff + (backward ? t - 1 : t + 1);
            else if (n_ok) ncol = n_boff + (backward ? n_Tt - 1 - n_s0 : n_s0);
            xload(ncol);
        }
        /* the three passes of the split products (cross terms first), the four gates interleaved */
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            ai = mfma16(wi[ks].p1, hp[ks].p2, ai); au = mfma16(wu[ks].p1, hp[ks].p2, au);
            af = mfma16(wf[ks].p1, hp[ks].p2, af); ao = mfma16(wo[ks].p1, hp[ks].p2, ao);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            ai = mfma16(wi[ks].p2, hp[ks].p1, ai); au = mfma16(wu[ks].p2, hp[ks].p1, au);
            af = mfma16(wf[ks].p2, hp[ks].p1, af); ao = mfma16(wo[ks].p2, hp[ks].p1, ao);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            ai = mfma16(wi[ks].p1, hp[ks].p1, ai); au = mfma16(wu[ks].p1, hp[ks].p1, au);
            af = mfma16(wf[ks].p1, hp[ks].p1, af); ao = mfma16(wo[ks].p1, hp[ks].p1, ao);
        }
        const bool active = t < myT;
        f32x4 o;
        const f32x4 ti = d_tanh4_acc(ai);
        const f32x4 pu = *(const f32x4 *)(peep + (u * 64 + lane) * 4);
        const f32x4 pf = *(const f32x4 *)(peep + ((NU + u) * 64 + lane) * 4);
        const f32x4 po = *(const f32x4 *)(peep + ((2 * NU + u) * 64 + lane) * 4);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float forget = d_logistic(af[k] * SH_OINV + c[k] * pf[k]) * c[k];         /* layers.c:811-813 */
            const float update = d_logistic(au[k] * SH_OINV + c[k] * pu[k]) * ti[k];        /* :815-817 */
            const float ns = forget + update;
            const float ho = d_logistic(ao[k] * SH_OINV + ns * po[k]) * d_tanh(ns);         /* :820-825 */
            c[k] = active ? ns : 0.0f;
            h[k] = active ? ho : 0.0f;
            o[k] = h[k];
        }
        *(f32x4 *)(out + ((long long)(boff + t) * NU + u) * 256 + lane * 4) = o;
        s++;
        if (s == s1) {                                       /* segment done */
            if (s1 < Tt) {                                   /* the tile continues on another lane */
                float *hs = L.hstate + ((long long)tile * 2 * NU + u) * 256 + lane * 4;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    __hip_atomic_store(hs + k, h[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(hs + NU * 256 + k, c[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                if (lane == 0) __hip_atomic_fetch_add(L.flag + tile, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (n_ok) {
                advance();
                fetch_next(++sgi);
                take_over();
            }
        }
        par ^= 1;
        publish(lds_h + par * PBUF, h);
        lds_barrier();
    }
    for (; it < nit; it++) lds_barrier();          /* the other lane of the workgroup is still stepping */
}

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
__device__ __forceinline__ float fin_post(float e, float recip, float mp, float mpm1, int want_log) {
    return want_log ? fin_log(e, recip * mpm1, mp) : e * recip;
}

/* ------------------------------------------------------------------ */
/* D1: transducer Viterbi, one tile of 16 reads per workgroup, the read  */
/* index innermost in every LDS/HBM access (decode.c:123-351).           */
/* Thread (qq = tid>>4, b = tid&15) owns quads Q = qq + 16 i of read b   */
/* (a quad = 4 consecutive k-mer states = the four one-base extensions   */
/* of one (k-1)-mer).  Moves are applied in the reference's order with   */
/* strict comparisons; suffix maxima keep the lowest prefix on ties.     */
/* Traceback is one byte per state per block (move type + prefix).       */
/* ------------------------------------------------------------------ */
struct ShVitArgs {
    const float *E;
    const float *sums;            /* NULL: E already final log-posterior */
    long long strideT;            /* floats between blocks */
    int strideQ, strideB;         /* floats between state quads / reads */
    int want_log;
    float min_prob, stay_pen, skip_pen, local_pen;
    int use_slip;
    unsigned *tb;                 /* [ncb][NQ][16] */
    int *tb_end;                  /* [ncb][16] */
    int *final_state;             /* [npad] */
    float *final_score;           /* [npad] */
    float *hp_side;               /* [sum T][5] or NULL */
    const long long *hp_off;      /* [npad] */
    unsigned long long *dbg;      /* experiment: per-wave phase cycle totals, or NULL */
    /* seg == NULL: workgroup g decodes tile g whole; else workgroup g decodes piece seg[g] (sh_sched.h) */
    const ShGruSegD *seg;
    float *vstate;                /* [ntile][NH * 16 + 32]: scores, start and end state of a tile cut between lanes */
    unsigned *flag;               /* [ntile] hand-over done */
    unsigned *err;                /* set when a hand-over never arrives */
};

__device__ __forceinline__ void argmax_merge(float &v, int &i, float ov, int oi) {
    /* keep the larger value; on equal values the lower index (first wins).
     * Written as selects: as an if() this compiles to exec-mask branches. */
    const bool take = (ov > v) | ((ov == v) & (oi < i));
    v = take ? ov : v;
    i = take ? oi : i;
}

/* FIN: the emissions are exp values to be normalised and logged here (a.sums given, log output);
 * SLIP: decode with the slip move.  Both are compile-time so that the block loop is straight-line code. */
/* one conditional move of the traceback code of state E of a quad: byte E of `codes` becomes byte 0 of `x` where a < b
 * (strict, as the reference compares).  v_cndmask_b32_sdwa writes the byte in place, so the four states of a quad
 * share one register without a shift and an or per state. */
#define SH_CODE_LT(E, codes, a, b, x)                                                                                       \
    asm("v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32_sdwa %0, %0, %3, vcc dst_sel:BYTE_" #E                                   \
        " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #E " src1_sel:BYTE_0"                                                  \
        : "+v"(codes) : "v"(a), "v"(b), "v"(x) : "vcc")

/* SKIP0: skip_pen == 0 (the default): the subtraction of the penalty is the identity and is left out */
#ifndef SH_VIT_RING
#define SH_VIT_RING 4
#endif
template <int NTH, int PPT, bool FIN, bool SLIP, bool SKIP0>
__global__ __launch_bounds__(NTH, NTH / 256) void k_viterbi(ShVitArgs a, ShMeta md) {
    constexpr int RING = (PPT >= SH_VIT_RING) ? SH_VIT_RING : PPT;           /* emission quads in flight */
    constexpr int QSTR = NTH / 16, NW = NTH / 64;      /* quads covered per pass, waves */
    constexpr int NQ = QSTR * PPT, NH = 4 * NQ;
    constexpr int NSKIP = NH / 16, NSLIP = (NH / 64 > 0) ? NH / 64 : 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    /* scores live in LDS, double buffered, read index innermost:
     * state s of read b at ((s>>2)*16 + b)*4 + (s&3) */
    float *scA = smem;                             /* NH*16 */
    float *scB = scA + NH * 16;                    /* NH*16 */
    float *skv = scB + NH * 16;                    /* NSKIP*16 */
    int *ski = (int *)(skv + NSKIP * 16);
    float *slv = (float *)(ski + NSKIP * 16);      /* NSLIP*16 */
    int *sli = (int *)(slv + NSLIP * 16);
    float *redv = (float *)(sli + NSLIP * 16);     /* 2*NW*16 */
    int *redi = (int *)(redv + 2 * NW * 16);

    const int tid = threadIdx.x, b = tid & 15, qq = tid >> 4, wave = tid >> 6, lane = tid & 63;
    const float mp = a.min_prob, mpm1 = 1.0f - a.min_prob;
    const float slip_pen = (float)(2.0 * a.skip_pen);     /* decode.c:275 */
    constexpr bool slip = SLIP && (NH / 64 > 0);
    unsigned long long vA = 0, vB = 0, vC = 0, vD = 0, vt0 = 0, vt1;
    long long vblocks = 0;
#define VSTAMP(acc) do { if (a.dbg) { vt1 = __builtin_readcyclecounter(); acc += vt1 - vt0; vt0 = vt1; } } while (0)

    /* this workgroup's piece of work: blocks [s0, s1) of one tile.  Pieces are numbered
     * so that a tile's earlier piece has the lower workgroup index (dispatched first). */
    int tile = blockIdx.x, s0 = 0, s1 = -1, ord = 0;
    if (a.seg) { const ShGruSegD sg = a.seg[blockIdx.x]; tile = sg.tile; s0 = sg.s0; s1 = sg.s1; ord = sg.pad; }
    tile = __builtin_amdgcn_readfirstlane(tile); s0 = __builtin_amdgcn_readfirstlane(s0); s1 = __builtin_amdgcn_readfirstlane(s1);
    const int Tt = __builtin_amdgcn_readfirstlane(md.tile_T[tile]);
    if (s1 < 0) s1 = Tt;
    const long long boff = __builtin_amdgcn_readfirstlane((int)md.tile_boff[tile]);
    const int rd = tile * 16 + b;
    const int myT = md.rT[rd];
    float pstart = 0.0f, pend = -SH_BIG;
    if (s0 == 0) {
        /* decode.c:155-159 */
#pragma unroll
        for (int i = 0; i < PPT; i++)
            *(f32x4 *)(scA + ((qq + QSTR * i) * 16 + b) * 4) = (f32x4){-SH_BIG, -SH_BIG, -SH_BIG, -SH_BIG};
        if (lane < 16) { redv[wave * 16 + b] = -SH_BIG - a.local_pen; redi[wave * 16 + b] = 4 * wave; }
    } else {
        /* the tile's earlier blocks ran on another workgroup: take over its state */
        if (tid == 0) {
            /* flag[tile] = number of pieces of the tile that are finished */
            if (!sh_wait_flag(a.flag + tile, (unsigned)ord)) __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const float *vst = a.vstate + (long long)tile * (NH * 16 + 32);
        float bv = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < PPT; i++) {
            const int Q = qq + QSTR * i;
            const f32x4 pv = *(const f32x4 *)(vst + (Q * 16 + b) * 4);
            *(f32x4 *)(scA + (Q * 16 + b) * 4) = pv;
            {   /* the end-state scan the last block would have left behind (per quad, as in the block loop) */
                const float ve = __builtin_fmaxf(__builtin_fmaxf(pv[0], pv[1]), __builtin_fmaxf(pv[2], pv[3])) - a.local_pen;
                bi = (ve > bv) ? Q : bi;
                bv = __builtin_fmaxf(bv, ve);
            }
        }
        pstart = vst[NH * 16 + b];
        pend = vst[NH * 16 + 16 + b];
        float ov = __shfl_xor(bv, 16); int oi = __shfl_xor(bi, 16);
        argmax_merge(bv, bi, ov, oi);
        ov = __shfl_xor(bv, 32); oi = __shfl_xor(bi, 32);
        argmax_merge(bv, bi, ov, oi);
        if (lane < 16) { redv[((s0 & 1) * NW + wave) * 16 + b] = bv; redi[((s0 & 1) * NW + wave) * 16 + b] = bi; }
    }
    __syncthreads();
    float *cur = scA, *nxt = scB;

    /* emissions are independent of the recurrence: block t+1's are fetched into
     * registers while block t is being processed */
    f32x4 ring[RING];
    float stay_nx = 0.f, sum_nx = 1.f;
    float hp_nx[4] = {0.f, 0.f, 0.f, 0.f};
    const bool hp_lane = FIN && a.hp_side && qq == 0;
    auto fetch = [&](int t) {
        const float *Ecb = a.E + (boff + t) * a.strideT + b * a.strideB;
        stay_nx = Ecb[NQ * a.strideQ];
        if (FIN) sum_nx = a.sums[(boff + t) * 16 + b];
        if (hp_lane) {
            /* the only five posterior rows homopolymer_path reads (homopolymer.c:200,209) */
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int s = k * ((NH - 1) / 3);            /* repeatblock(k, klen) */
                hp_nx[k] = Ecb[(s >> 2) * a.strideQ + (s & 3)];
            }
        }
    };
    /* global addresses as (wave-uniform 64-bit base) + (32-bit lane offset): the bases live in scalar
     * registers, one VGPR serves all quads */
    const unsigned eofs = (unsigned)(b * a.strideB + qq * a.strideQ);
    const unsigned tofs = (unsigned)(qq * 16 + b);
    auto qload = [&](int t, int i) {
        const float *base = a.E + (boff + t) * a.strideT + (long long)(QSTR * i) * a.strideQ;     /* uniform */
        return *(const f32x4 *)(base + eofs);
    };
    if (s1 > s0) {
        fetch(s0);
#pragma unroll
        for (int i = 0; i < RING; i++) ring[i] = qload(s0, i);
    }
    if (a.dbg) vt0 = __builtin_readcyclecounter();
    vblocks += s1 - s0;

    for (int t = s0; t < s1; t++) {
        const long long cb = boff + t;
        const int par = t & 1;
        /* raw_nx / stay_nx / sum_nx / hp_nx hold THIS block's emissions (fetched at
         * the end of the previous iteration, in flight across the barriers) */
        float stay_lp = stay_nx;
        const float rmf = (1.0f / sum_nx) * mpm1;           /* fin_log's factor */

        /* phase B: skip / slip suffix maxima, each (suffix, read) once; lowest
         * prefix wins ties (decode.c:228-251, :276-302) */
        for (int p = tid; p < NSKIP * 16; p += NTH) {
            const int j = p >> 4, bb = p & 15;
            float v = cur[((j >> 2) * 16 + bb) * 4 + (j & 3)];
            int ri = 0;
#pragma unroll
            for (int r = 1; r < 16; r++) {
                const int s = r * NSKIP + j;
                const float c = cur[((s >> 2) * 16 + bb) * 4 + (s & 3)];
                const bool up = v < c;
                v = up ? c : v;
                ri = up ? r : ri;
            }
            skv[p] = v; ski[p] = ri;
        }
        if (slip) {
            for (int p = tid; p < NSLIP * 16; p += NTH) {
                const int j = p >> 4, bb = p & 15;
                float v = cur[((j >> 2) * 16 + bb) * 4 + (j & 3)];
                int ri = 0;
                for (int r = 1; r < 64; r++) {
                    const int s = r * NSLIP + j;
                    const float c = cur[((s >> 2) * 16 + bb) * 4 + (s & 3)];
                    const bool up = v < c;
                    v = up ? c : v;
                    ri = up ? r : ri;
                }
                slv[p] = v; sli[p] = ri;
            }
        }
        if (FIN) stay_lp = fin_log(stay_lp, rmf, mp);
        if (hp_lane && t < myT) {
            float *hs = a.hp_side + (a.hp_off[rd] + t) * 5;
#pragma unroll
            for (int k = 0; k < 4; k++) hs[k] = fin_log(hp_nx[k], rmf, mp);
            hs[4] = stay_lp;
        }
        VSTAMP(vA);
        __syncthreads();
        VSTAMP(vB);

        /* phase C: update my states, cur -> nxt */
        const bool active = t < myT;
        /* A read past its end keeps its scores.  With the emissions finalised here that needs no select per
         * state: for such a read the emission factor and floor are zeroed -- every emission becomes log 0 =
         * -inf, which loses every strict comparison -- and the stay move adds 0, so each state comes out of the
         * update with the bits it went in with. */
        const float rm = (FIN && !active) ? 0.0f : rmf;
        const float mpx = (FIN && !active) ? 0.0f : mp;
        const float stay_v = (FIN && !active) ? 0.0f : stay_lp - a.stay_pen;          /* decode.c:175-176 */
        float ev = redv[par * NW * 16 + b];
        int ei = redi[par * NW * 16 + b];
        for (int w = 1; w < NW; w++) argmax_merge(ev, ei, redv[(par * NW + w) * 16 + b], redi[(par * NW + w) * 16 + b]);
        const float stay_act = stay_lp - a.stay_pen;
        const float hold = fmaxf(-a.local_pen, stay_act);
        const float nstart = pstart + hold;                 /* decode.c:326 */
        float nend = pend + hold;                           /* decode.c:339 */
        const bool enter_end = ev > nend;                   /* decode.c:343-348 */
        nend = enter_end ? ev : nend;
        if (active && qq == 0) {
            /* ei is the first QUAD that holds the maximum of (score - local_pen); the state is the first of its
             * four that attains it (the subtraction is monotone, so the quad's maximum does) */
            int tbe = NH + 1;
            if (enter_end) {
                const f32x4 q4 = *(const f32x4 *)(cur + (ei * 16 + b) * 4);
                int e0 = 3;
                e0 = (q4[2] - a.local_pen == ev) ? 2 : e0;
                e0 = (q4[1] - a.local_pen == ev) ? 1 : e0;
                e0 = (q4[0] - a.local_pen == ev) ? 0 : e0;
                tbe = 4 * ei + e0;
            }
            a.tb_end[cb * 16 + b] = tbe;
        }
        float bv = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < PPT; i++) {
            const int Q = qq + QSTR * i;
            const f32x4 pv = *(const f32x4 *)(cur + (Q * 16 + b) * 4);
            f32x4 l4 = ring[i % RING];
            /* keep RING quads of emissions in flight: the rest of this block, then the next block's first ones */
            if (i + RING < PPT) ring[i % RING] = qload(t, i + RING);
            else if (t + 1 < s1) ring[i % RING] = qload(t + 1, i + RING - PPT);
            if (FIN) {
#pragma unroll
                for (int e = 0; e < 4; e++) l4[e] = fin_log(l4[e], rm, mpx);
            }
            /* step: max over the 4 prefixes of suffix Q (decode.c:186-210) */
            float sv = cur[((Q >> 2) * 16 + b) * 4 + (Q & 3)];
            int sr = 0;
#pragma unroll
            for (int r = 1; r < 4; r++) {
                const float c = cur[(((r * NQ + Q) >> 2) * 16 + b) * 4 + (Q & 3)];
                const bool up = sv < c;
                sv = up ? c : sv;
                sr = up ? r : sr;
            }
            const float kv = skv[(Q >> 2) * 16 + b];
            const int kr = ski[(Q >> 2) * 16 + b];
            float lv = 0.f; int lr = 0;
            if (slip) { lv = slv[(Q >> 4) * 16 + b]; lr = sli[(Q >> 4) * 16 + b]; }
            const unsigned cstep = SH_TB_STEP + (unsigned)sr, cskip = SH_TB_SKIP + (unsigned)kr, cslip = SH_TB_SLIP + (unsigned)lr;
            const unsigned cstart = SH_TB_START;
            unsigned codes = 0;                             /* four SH_TB_STAY */
            f32x4 ns;
#define SH_VIT_STATE(E)                                                                                         \
            {                                                                                                   \
                /* score: max() is the same value as the reference's compare-and-take (no NaNs here); the     */  \
                /* move code needs the strict comparison                                                      */  \
                float sc = pv[E] + stay_v;                  /* stay  :180 */                                    \
                const float st = l4[E] + sv;                /* step  :214-218 */                                \
                SH_CODE_LT(E, codes, sc, st, cstep);                                                            \
                sc = __builtin_fmaxf(sc, st);                                                                   \
                const float sk = SKIP0 ? l4[E] + kv : (l4[E] + kv) - a.skip_pen;   /* skip  :256-262 */         \
                SH_CODE_LT(E, codes, sc, sk, cskip);                                                            \
                sc = __builtin_fmaxf(sc, sk);                                                                   \
                if (slip) {                                 /* wave-uniform */                                  \
                    const float sl = (l4[E] + lv) - slip_pen;    /* slip :307-314 */                            \
                    SH_CODE_LT(E, codes, sc, sl, cslip);                                                        \
                    sc = __builtin_fmaxf(sc, sl);                                                               \
                }                                                                                               \
                const float fs = pstart + l4[E];            /* leave start :331-335 */                          \
                SH_CODE_LT(E, codes, sc, fs, cstart);                                                           \
                sc = __builtin_fmaxf(sc, fs);                                                                   \
                ns[E] = (FIN || active) ? sc : pv[E];                                                           \
            }
            SH_VIT_STATE(0) SH_VIT_STATE(1) SH_VIT_STATE(2) SH_VIT_STATE(3)
#undef SH_VIT_STATE
            *(f32x4 *)(nxt + (Q * 16 + b) * 4) = ns;
            (a.tb + (cb * NQ + QSTR * i) * 16)[tofs] = codes;   /* also for reads past their end (never read back): no branch */
            {   /* next block's end-state scan, per quad: this thread meets its quads in increasing index order,
                 * so a strict compare keeps the first maximum */
                const float ve = __builtin_fmaxf(__builtin_fmaxf(ns[0], ns[1]), __builtin_fmaxf(ns[2], ns[3])) - a.local_pen;
                bi = (ve > bv) ? Q : bi;
                bv = __builtin_fmaxf(bv, ve);
            }
            __builtin_amdgcn_sched_barrier(0);          /* quads one after the other: register budget of 3 waves per SIMD */
        }
        if (active) { pstart = nstart; pend = nend; }
        {
            float ov = __shfl_xor(bv, 16); int oi = __shfl_xor(bi, 16);
            argmax_merge(bv, bi, ov, oi);
            ov = __shfl_xor(bv, 32); oi = __shfl_xor(bi, 32);
            argmax_merge(bv, bi, ov, oi);
            if (lane < 16) { redv[((par ^ 1) * NW + wave) * 16 + b] = bv; redi[((par ^ 1) * NW + wave) * 16 + b] = bi; }
        }
        if (t + 1 < s1) fetch(t + 1);      /* next block's emissions: no register-heavy code until they are used */
        VSTAMP(vC);
        __syncthreads();
        VSTAMP(vD);
        { float *x = cur; cur = nxt; nxt = x; }
    }

    if (s1 < Tt) {
        /* the tile's later blocks run on another workgroup: leave it the state */
        float *vst = a.vstate + (long long)tile * (NH * 16 + 32);
#pragma unroll
        for (int i = 0; i < PPT; i++) {
            const int Q = qq + QSTR * i;
            *(f32x4 *)(vst + (Q * 16 + b) * 4) = *(const f32x4 *)(cur + (Q * 16 + b) * 4);
        }
        if (qq == 0) { vst[NH * 16 + b] = pstart; vst[NH * 16 + 16 + b] = pend; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(a.flag + tile, (unsigned)ord + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
    /* argmaxf over nh+2 final scores, first maximum wins (decode.c:68, util.c:9) */
    float bv = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < PPT; i++) {
        const int Q = qq + QSTR * i;
        const f32x4 pv = *(const f32x4 *)(cur + (Q * 16 + b) * 4);
#pragma unroll
        for (int e = 0; e < 4; e++) argmax_merge(bv, bi, pv[e], 4 * Q + e);
    }
    {
        float ov = __shfl_xor(bv, 16); int oi = __shfl_xor(bi, 16);
        argmax_merge(bv, bi, ov, oi);
        ov = __shfl_xor(bv, 32); oi = __shfl_xor(bi, 32);
        argmax_merge(bv, bi, ov, oi);
        __syncthreads();
        if (lane < 16) { redv[wave * 16 + b] = bv; redi[wave * 16 + b] = bi; }
    }
    __syncthreads();
    if (qq == 0) {
        float ev = redv[b]; int ei = redi[b];
        for (int w = 1; w < NW; w++) argmax_merge(ev, ei, redv[w * 16 + b], redi[w * 16 + b]);
        if (pstart > ev) { ev = pstart; ei = NH; }
        if (pend > ev) { ev = pend; ei = NH + 1; }
        a.final_state[rd] = ei;
        a.final_score[rd] = ev;
    }
    }
    if (a.dbg && lane == 0) { unsigned long long *d = a.dbg + ((long long)blockIdx.x * NW + wave) * 8; d[0] = vA; d[1] = vB; d[2] = vC; d[3] = vD; d[4] = (unsigned long long)vblocks; }
}

/* ------------------------------------------------------------------ */
/* S1 + D1 in one kernel: the exp-posterior of a block is produced by the */
/* decoder's own waves, in the registers of the threads that consume it,  */
/* and never exists in memory (round 1 and the first half of round 2:     */
/* k_ff_lds wrote 33.6 GB per 10 000 x 4000-sample step, k_viterbi read    */
/* them back).  For 4^5 + 1 states over a 96-wide trunk:                   */
/*  * 8 waves, one tile of 16 reads per workgroup, scores in LDS as in     */
/*    k_viterbi.  Wave w owns m-tiles 8w .. 8w+7 of the S1 weight matrix   */
/*    (their fp16 pieces stream from L2 once per block, see below)         */
/*    -- and the MFMA result layout (lane = (q, read b), 4 consecutive     */
/*    rows) is exactly a state quad of read b, so thread (w, q, b) decodes */
/*    quads 32w + 4i + q, i < 8: the ones it has the emissions of.         */
/*  * Block t+1's emissions are multiplied and exponentiated while block t */
/*    is decoded (the matrix pipe is otherwise idle); their row sum goes   */
/*    through LDS in SH_SUM_GROUP order (group w = wave w, group 8 = the   */
/*    stay state's tile, computed by wave 7 from weights it re-reads from  */
/*    L2), so the bits are those of k_ff_lds + k_viterbi<.., FIN>.         */
/*  * The trunk output of block t+2 is cut into pieces once per workgroup  */
/*    (waves 0-2) and shared through LDS.                                  */
/* ------------------------------------------------------------------ */
#ifndef SH_FV_MIX
#define SH_FV_MIX 0         /* VALU instructions between two MFMAs of a quad's chain in k_ff_viterbi (0: compiler's order) */
#endif
#ifndef SH_FV_SB
#define SH_FV_SB 1          /* scheduling barrier after every SH_FV_SB quads of k_ff_viterbi's update loop (0: none) */
#endif
struct ShFfArgs {
    const float *in;              /* trunk output [ncb][6][64][4] */
    const unsigned *wpiece;       /* S1 weights as pieces [65][3][2][64][4] */
    const float *bfrag;           /* bias fragments x 2^14 [65][64][4] */
    float in_div, out_div;        /* softmax_with_temperature's two divisions (1: none) */
};
#define SH_FV_LDS_FLOATS (2 * 1024 * 16 + 2 * 64 * 16 + 2 * 16 * 16 + 4 * 8 * 16 + 2 * 3 * 512 + 2 * 9 * 16 + 65 * 16 + 3 * 2 * 4 * 4)

template <bool SLIP, bool SKIP0, bool DIV>
__global__ __launch_bounds__(512, 2) void k_ff_viterbi(ShFfArgs f, ShVitArgs a, ShMeta md) {
    constexpr int NTH = 512, NW = 8, PPT = 8, NQ = 256, NH = 1024, NSKIP = NH / 16, NSLIP = NH / 64, KS = 3, KQ = 6;
    static_assert(PPT == SH_SUM_GROUP, "a wave's tiles are one row-sum group");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *scA = smem;                             /* scores, double buffered: state s of read b at ((s>>2)*16 + b)*4 + (s&3) */
    float *scB = scA + NH * 16;
    float *skv = scB + NH * 16;
    int *ski = (int *)(skv + NSKIP * 16);
    float *slv = (float *)(ski + NSKIP * 16);
    int *sli = (int *)(slv + NSLIP * 16);
    float *redv = (float *)(sli + NSLIP * 16);     /* 2*NW*16 */
    int *redi = (int *)(redv + 2 * NW * 16);
    unsigned *xp = (unsigned *)(redi + 2 * NW * 16);     /* trunk columns as pieces [2][KS][2][64][4] */
    float *gsum = (float *)(xp + 2 * KS * 512);          /* row-sum groups [2][NW + 1][16] */
    float *sBias = gsum + 2 * (NW + 1) * 16;             /* bias x 2^14 by state row [65 * 16] */
    unsigned *sStay = (unsigned *)(sBias + 65 * 16);     /* row 1024 of the weights as pieces [KS][2][4 k groups][4] */

    const int tid = threadIdx.x, lane = tid & 63, b = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float mp = a.min_prob, mpm1 = 1.0f - a.min_prob;
    const float slip_pen = (float)(2.0 * a.skip_pen);     /* decode.c:275 */
    unsigned long long vA = 0, vB = 0, vC = 0, vD = 0, vt0 = 0, vt1;

    /* this wave's rows of the S1 weights: 48 KB of fp16 pieces, streamed from L2 once per block, one m-tile (24
     * VGPRs) ahead of the MFMAs that use it.  (The whole matrix is 394 KB -- more than the CU's LDS, and with the
     * decoder's state more than its register file; tools/l2_stream_probe.hip: all 256 CUs re-reading it
     * concurrently take 2.9 us per pass, 35 TB/s aggregate, against ~5 us of decoding per block.) */
    const unsigned *wmine = f.wpiece + (long long)(PPT * wave) * KS * 512;
    ShSplit W[2][KS];
    /* global addresses as (wave-uniform 64-bit base in scalar registers) + (32-bit lane offset): one VGPR serves all */
    const unsigned lofs = (unsigned)lane * 4u, tofs = (unsigned)lane;
    auto w_load = [&](int i) {
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            typedef const __attribute__((address_space(1))) unsigned *gu32;
            typedef const __attribute__((address_space(1))) u32x4 *gu32x4;
            gu32 base = (gu32)(wmine + (i * KS + ks) * 512);             /* uniform */
            asm volatile("" : "+s"(base));       /* ... and kept so: else 48 loop-invariant 64-bit VGPR addresses are formed (and spilled) */
            W[i & 1][ks].p1 = __builtin_bit_cast(f16x8, *(gu32x4)(base + lofs));
            W[i & 1][ks].p2 = __builtin_bit_cast(f16x8, *(gu32x4)(base + 256 + lofs));
        }
    };
    if (tid < KS * 2 * 4 * 4) sStay[tid] = f.wpiece[(long long)(PPT * NW) * KS * 512 + (tid >> 4) * 256 + ((tid >> 2) & 3) * 64 + (tid & 3)];
    for (int j = tid; j < 65 * 16; j += NTH) sBias[j] = f.bfrag[((j >> 4) * 64 + ((j >> 2) & 3) * 16) * 4 + (j & 3)];

    int tile = blockIdx.x, s0 = 0, s1 = -1, ord = 0;
    if (a.seg) { const ShGruSegD sg = a.seg[blockIdx.x]; tile = sg.tile; s0 = sg.s0; s1 = sg.s1; ord = sg.pad; }
    tile = __builtin_amdgcn_readfirstlane(tile); s0 = __builtin_amdgcn_readfirstlane(s0); s1 = __builtin_amdgcn_readfirstlane(s1);
    const int Tt = __builtin_amdgcn_readfirstlane(md.tile_T[tile]);
    if (s1 < 0) s1 = Tt;
    const long long boff = __builtin_amdgcn_readfirstlane((int)md.tile_boff[tile]);
    const int rd = tile * 16 + b;
    const int myT = md.rT[rd];
    const long long hpo = a.hp_side ? a.hp_off[rd] : 0;
    float pstart = 0.0f, pend = -SH_BIG;
    if (s0 == 0) {
        /* decode.c:155-159 */
#pragma unroll
        for (int i = 0; i < PPT; i++)
            *(f32x4 *)(scA + ((32 * wave + 4 * i + q) * 16 + b) * 4) = (f32x4){-SH_BIG, -SH_BIG, -SH_BIG, -SH_BIG};
        if (lane < 16) { redv[wave * 16 + b] = -SH_BIG - a.local_pen; redi[wave * 16 + b] = 32 * wave; }
    } else {
        /* the tile's earlier blocks ran on another workgroup: take over its state */
        if (tid == 0) {
            if (!sh_wait_flag(a.flag + tile, (unsigned)ord)) __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const float *vst = a.vstate + (long long)tile * (NH * 16 + 32);
        float bv = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < PPT; i++) {
            const int Q = 32 * wave + 4 * i + q;
            const f32x4 pv = *(const f32x4 *)(vst + (Q * 16 + b) * 4);
            *(f32x4 *)(scA + (Q * 16 + b) * 4) = pv;
            {   /* the end-state scan the last block would have left behind (per quad, as in the block loop) */
                const float ve = __builtin_fmaxf(__builtin_fmaxf(pv[0], pv[1]), __builtin_fmaxf(pv[2], pv[3])) - a.local_pen;
                bi = (ve > bv) ? Q : bi;
                bv = __builtin_fmaxf(bv, ve);
            }
        }
        pstart = vst[NH * 16 + b];
        pend = vst[NH * 16 + 16 + b];
        float ov = __shfl_xor(bv, 16); int oi = __shfl_xor(bi, 16);
        argmax_merge(bv, bi, ov, oi);
        ov = __shfl_xor(bv, 32); oi = __shfl_xor(bi, 32);
        argmax_merge(bv, bi, ov, oi);
        if (lane < 16) { redv[((s0 & 1) * NW + wave) * 16 + b] = bv; redi[((s0 & 1) * NW + wave) * 16 + b] = bi; }
    }
    float *cur = scA, *nxt = scB;

    /* --- S1 --- */
    /* waves 0-2: trunk column block t, k step `wave`, as raw fp32 (in flight for a whole step) ... */
    f32x4 xr0 = {0.f, 0.f, 0.f, 0.f}, xr1 = xr0;
    auto xraw_load = [&](int t) {
        if (wave < KS) {
            const float *p = f.in + ((boff + min(t, s1 - 1)) * KQ + 2 * wave) * 256;       /* uniform */
            xr0 = *(const f32x4 *)(p + lofs);
            xr1 = *(const f32x4 *)(p + 256 + lofs);
        }
    };
    /* ... and cut into pieces for everybody */
    auto xp_publish = [&](int buf) {
        if (wave < KS) {
            f32x4 v0 = xr0, v1 = xr1;
            if (DIV) { v0 = v0 / f.in_div; v1 = v1 / f.in_div; }      /* shift_scale_matrix_inplace: division (Q5); x / 1 = x */
            const ShSplit sp = split8(v0, v1);
            unsigned *d = xp + (buf * KS + wave) * 512 + lane * 4;
            *(u32x4 *)d = __builtin_bit_cast(u32x4, sp.p1);
            *(u32x4 *)(d + 256) = __builtin_bit_cast(u32x4, sp.p2);
        }
    };
    auto e_of = [&](float acc) { return DIV ? d_exp((acc * SH_OINV) / f.out_div) : d_exp_acc(acc); };   /* no max subtraction (Q2) */
    /* the stay state's m-tile (row 1024 and 15 rows of padding, whose results are masked: only the lanes that
     * hold row 0 of the A operand need real weights -- 384 bytes, kept in LDS): wave 7 */
    auto stay_group = [&](const ShSplit (&bp)[KS], int buf) {
        if (wave == NW - 1) {
            ShSplit Ws[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                const u32x4 a1 = *(const u32x4 *)(sStay + ((ks * 2 + 0) * 4 + q) * 4), a2 = *(const u32x4 *)(sStay + ((ks * 2 + 1) * 4 + q) * 4);
                const u32x4 z = {0u, 0u, 0u, 0u};
                Ws[ks].p1 = __builtin_bit_cast(f16x8, b == 0 ? a1 : z);
                Ws[ks].p2 = __builtin_bit_cast(f16x8, b == 0 ? a2 : z);
            }
            f32x4 acc = *(const f32x4 *)(sBias + (PPT * NW) * 16 + 4 * q);
            acc = split_dot<KS>(Ws, bp, acc);
            f32x4 ex;
#pragma unroll
            for (int r = 0; r < 4; r++) ex[r] = (4 * q + r < 1) ? e_of(acc[r]) : 0.0f;        /* rows >= NS are padding */
            float v = (ex[0] + ex[1]) + (ex[2] + ex[3]);
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (lane < 16) gsum[(buf * (NW + 1) + NW) * 16 + b] = v;
        }
    };
    auto group_out = [&](float part, int buf) {
        float v = part;
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lane < 16) gsum[(buf * (NW + 1) + wave) * 16 + b] = v;
    };

    f32x4 e[PPT];
    if (s1 > s0) {
        xraw_load(s0);
        xp_publish(s0 & 1);
        xraw_load(s0 + 1);
        xp_publish((s0 + 1) & 1);
        xraw_load(s0 + 2);
        __syncthreads();
        ShSplit bp[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ks++) bp[ks] = load_pieces(xp + ((s0 & 1) * KS + ks) * 512, lane);
        float part = 0.0f;
        w_load(0);
#pragma unroll
        for (int i = 0; i < PPT; i++) {
            f32x4 acc = *(const f32x4 *)(sBias + (PPT * wave + i) * 16 + 4 * q);
            w_load((i + 1) & (PPT - 1));
            acc = split_dot<KS>(W[i & 1], bp, acc);
#pragma unroll
            for (int r = 0; r < 4; r++) e[i][r] = e_of(acc[r]);
            part += (e[i][0] + e[i][1]) + (e[i][2] + e[i][3]);
        }
        group_out(part, s0 & 1);
        stay_group(bp, s0 & 1);
    }
    __syncthreads();

    if (a.dbg) vt0 = __builtin_readcyclecounter();
    /* one block; MORE: there is a block t+1 to prepare the emissions of (all but the piece's last) */
    auto block = [&](const int t, auto more_c) {
        constexpr bool more = decltype(more_c)::value;
        const long long cb = boff + t;
        const int par = t & 1;

        /* phase B: skip / slip suffix maxima, each (suffix, read) once; lowest prefix wins ties (decode.c:228-251, :276-302) */
        for (int p = tid; p < NSKIP * 16; p += NTH) {
            const int j = p >> 4, bb = p & 15;
            float v = cur[((j >> 2) * 16 + bb) * 4 + (j & 3)];
            int ri = 0;
#pragma unroll
            for (int r = 1; r < 16; r++) {
                const int s = r * NSKIP + j;
                const float c = cur[((s >> 2) * 16 + bb) * 4 + (s & 3)];
                const bool up = v < c;
                v = up ? c : v;
                ri = up ? r : ri;
            }
            skv[p] = v; ski[p] = ri;
        }
        if (SLIP) {
            for (int p = tid; p < NSLIP * 16; p += NTH) {
                const int j = p >> 4, bb = p & 15;
                float v = cur[((j >> 2) * 16 + bb) * 4 + (j & 3)];
                int ri = 0;
                for (int r = 1; r < 64; r++) {
                    const int s = r * NSLIP + j;
                    const float c = cur[((s >> 2) * 16 + bb) * 4 + (s & 3)];
                    const bool up = v < c;
                    v = up ? c : v;
                    ri = up ? r : ri;
                }
                slv[p] = v; sli[p] = ri;
            }
        }
        VSTAMP(vA);
        __syncthreads();
        VSTAMP(vB);

        /* phase C: update my states, cur -> nxt; block t+1's emissions alongside */
        float tot = 0.0f;
#pragma unroll
        for (int w = 0; w < NW + 1; w++) tot += gsum[(par * (NW + 1) + w) * 16 + b];
        const float rmf = (1.0f / tot) * mpm1;                      /* fin_log's factor */
        const float stay_lp = fin_log(gsum[(par * (NW + 1) + NW) * 16 + b], rmf, mp);
        const bool active = t < myT;
        if (a.hp_side && active && tid < 16) (a.hp_side + (hpo + t) * 5)[4] = stay_lp;
        /* a read past its end keeps its scores: see k_viterbi */
        const float rm = active ? rmf : 0.0f;
        const float mpx = active ? mp : 0.0f;
        const float stay_v = active ? stay_lp - a.stay_pen : 0.0f;  /* decode.c:175-176 */
        float ev = redv[par * NW * 16 + b];
        int ei = 0;
        if (wave == 0) {                                   /* the index is needed by the threads that write tb_end only */
            ei = redi[par * NW * 16 + b];
            for (int w = 1; w < NW; w++) argmax_merge(ev, ei, redv[(par * NW + w) * 16 + b], redi[(par * NW + w) * 16 + b]);
        } else {
#pragma unroll
            for (int w = 1; w < NW; w++) ev = __builtin_fmaxf(ev, redv[(par * NW + w) * 16 + b]);
        }
        const float stay_act = stay_lp - a.stay_pen;
        const float hold = fmaxf(-a.local_pen, stay_act);
        const float nstart = pstart + hold;                 /* decode.c:326 */
        float nend = pend + hold;                           /* decode.c:339 */
        const bool enter_end = ev > nend;                   /* decode.c:343-348 */
        nend = enter_end ? ev : nend;
        if (active && tid < 16) {
            int tbe = NH + 1;
            if (enter_end) {
                const f32x4 q4 = *(const f32x4 *)(cur + (ei * 16 + b) * 4);
                int e0 = 3;
                e0 = (q4[2] - a.local_pen == ev) ? 2 : e0;
                e0 = (q4[1] - a.local_pen == ev) ? 1 : e0;
                e0 = (q4[0] - a.local_pen == ev) ? 0 : e0;
                tbe = 4 * ei + e0;
            }
            a.tb_end[cb * 16 + b] = tbe;
        }
        ShSplit bp[KS];
        if (more) {
#pragma unroll
            for (int ks = 0; ks < KS; ks++) bp[ks] = load_pieces(xp + ((par ^ 1) * KS + ks) * 512, lane);
        }
        float part = 0.0f;
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        /* a quad's inputs from LDS are read one quad ahead: the compiler may not move them across the score
         * stores itself (cur / nxt swap), and their round trips are the critical path of a quad otherwise */
        f32x4 pv_n, sc4_n, bias_n; float kv_n, lv_n = 0.f; int kr_n, lr_n = 0;
        auto q_fetch = [&](int i) {
            const int Q = 32 * wave + 4 * i + q;
            pv_n = *(const f32x4 *)(cur + (Q * 16 + b) * 4);
#pragma unroll
            for (int r = 0; r < 4; r++) sc4_n[r] = cur[(((r * NQ + Q) >> 2) * 16 + b) * 4 + (Q & 3)];
            kv_n = skv[(Q >> 2) * 16 + b];
            kr_n = ski[(Q >> 2) * 16 + b];
            if (SLIP) { lv_n = slv[(Q >> 4) * 16 + b]; lr_n = sli[(Q >> 4) * 16 + b]; }
            if (more) bias_n = *(const f32x4 *)(sBias + (PPT * wave + i) * 16 + 4 * q);
        };
        /* (with the slip move the kernel is at its register limit: there the inputs are read where they are used) */
        constexpr bool AHEAD = !SLIP;
        if (AHEAD) q_fetch(0);
        float hpv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < PPT; i++) {
            const int Q = 32 * wave + 4 * i + q;
            if (!AHEAD) q_fetch(i);
            const f32x4 pv = pv_n, sc4 = sc4_n;
            const float kv = kv_n, lv = lv_n; const int kr = kr_n, lr = lr_n;
            f32x4 accn = bias_n;
            w_load((i + 1) & (PPT - 1));                    /* the next m-tile's weights (after the last: the first, for the next block) */
            if (more) accn = split_dot<KS>(W[i & 1], bp, accn);      /* tile i of block t+1: 9 MFMAs, under the VALU work below */
            if (AHEAD && i + 1 < PPT) q_fetch(i + 1);
            f32x4 l4;
#pragma unroll
            for (int k = 0; k < 4; k++) l4[k] = fin_log(e[i][k], rm, mpx);
            /* the only five posterior rows homopolymer_path reads (homopolymer.c:200,209): repeatblock(k, klen) and
             * stay; kept here, stored after the loop (no branches inside it) */
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int s = k * ((NH - 1) / 3), sq = s >> 2;
                if (i == ((sq >> 2) & 7)) hpv[k] = l4[s & 3];
            }
            /* step: max over the 4 prefixes of suffix Q (decode.c:186-210) */
            float sv = sc4[0];
            int sr = 0;
#pragma unroll
            for (int r = 1; r < 4; r++) {
                const bool up = sv < sc4[r];
                sv = up ? sc4[r] : sv;
                sr = up ? r : sr;
            }
            const unsigned cstep = SH_TB_STEP + (unsigned)sr, cskip = SH_TB_SKIP + (unsigned)kr, cslip = SH_TB_SLIP + (unsigned)lr;
            const unsigned cstart = SH_TB_START;
            unsigned codes = 0;                             /* four SH_TB_STAY */
            f32x4 ns = {0.f, 0.f, 0.f, 0.f};
#define SH_FV_STATE(E)                                                                                          \
            {                                                                                                   \
                float sc = pv[E] + stay_v;                  /* stay  :180 */                                    \
                const float st = l4[E] + sv;                /* step  :214-218 */                                \
                SH_CODE_LT(E, codes, sc, st, cstep);                                                            \
                sc = __builtin_fmaxf(sc, st);                                                                   \
                const float sk = SKIP0 ? l4[E] + kv : (l4[E] + kv) - a.skip_pen;   /* skip  :256-262 */         \
                SH_CODE_LT(E, codes, sc, sk, cskip);                                                            \
                sc = __builtin_fmaxf(sc, sk);                                                                   \
                if (SLIP) {                                                                                     \
                    const float sl = (l4[E] + lv) - slip_pen;    /* slip :307-314 */                            \
                    SH_CODE_LT(E, codes, sc, sl, cslip);                                                        \
                    sc = __builtin_fmaxf(sc, sl);                                                               \
                }                                                                                               \
                const float fs = pstart + l4[E];            /* leave start :331-335 */                          \
                SH_CODE_LT(E, codes, sc, fs, cstart);                                                           \
                sc = __builtin_fmaxf(sc, fs);                                                                   \
                ns[E] = sc;                                                                                     \
            }
            /* The three moves INTO a state add the same emission to three per-quad values, and rounding is monotone:
             * max(l + sv, l + kv, l + pstart) = l + max(sv, kv, pstart) exactly.  So the score needs one addition
             * instead of three -- and the move code is that of the first of (step, skip, start) holding the
             * maximum m, PROVIDED no other candidate x < m rounds to the same sum, i.e. unless m - x <= ulp of the
             * sum.  Quads where the runner-up is within 2^-21 (|m| + max |l|) of m (twice the largest possible
             * ulp), or an emission is -inf, in any lane, take the reference's compare-by-compare form below; the
             * others (all but ~1e-3) get by with 5 instead of 13 operations per state.  Reads past their end have
             * l = -inf: every move loses against stay in either form, so they do not count. */
            bool fast = false;
            if (!SLIP && SKIP0) {
                const float m = __builtin_fmaxf(__builtin_fmaxf(sv, kv), pstart);
                const float md = __builtin_amdgcn_fmed3f(sv, kv, pstart);
                const float amax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(l4[0]), __builtin_fabsf(l4[1])), __builtin_fmaxf(__builtin_fabsf(l4[2]), __builtin_fabsf(l4[3])));
                const bool clear = (m - md) > (amax + __builtin_fabsf(m)) * 4.76837158203125e-07f;       /* false for NaN / inf */
                fast = __builtin_amdgcn_ballot_w64(active && !clear) == 0;
                if (fast) {
                    unsigned cm = cstart;
                    cm = (kv == m) ? cskip : cm;
                    cm = (sv == m) ? cstep : cm;
#define SH_FV_FAST(E)                                                                                           \
                    {                                                                                           \
                        const float sc = pv[E] + stay_v;        /* stay  :180 */                                \
                        const float mv = l4[E] + m;             /* the best move into the state */              \
                        SH_CODE_LT(E, codes, sc, mv, cm);                                                       \
                        ns[E] = __builtin_fmaxf(sc, mv);                                                        \
                    }
                    SH_FV_FAST(0) SH_FV_FAST(1) SH_FV_FAST(2) SH_FV_FAST(3)
#undef SH_FV_FAST
                }
            }
            if (!fast) { SH_FV_STATE(0) SH_FV_STATE(1) SH_FV_STATE(2) SH_FV_STATE(3) }
#undef SH_FV_STATE
            *(f32x4 *)(nxt + (Q * 16 + b) * 4) = ns;
            (a.tb + (cb * NQ + 32 * wave + 4 * i) * 16)[tofs] = codes;   /* also for reads past their end (never read back): no branch */
            {   /* next block's end-state scan, per quad: this thread meets its quads in increasing index order,
                 * so a strict compare keeps the first maximum */
                const float ve = __builtin_fmaxf(__builtin_fmaxf(ns[0], ns[1]), __builtin_fmaxf(ns[2], ns[3])) - a.local_pen;
                bi = (ve > bv) ? Q : bi;
                bv = __builtin_fmaxf(bv, ve);
            }
            if (more) {                                     /* block t's emissions of this quad are used up: in place */
#pragma unroll
                for (int r = 0; r < 4; r++) e[i][r] = e_of(accn[r]);
                part += (e[i][0] + e[i][1]) + (e[i][2] + e[i][3]);
            }
#if SH_FV_MIX
            if (more) {     /* the quad's 9 MFMAs (a dependent chain: 16 cycles each) spread through its VALU work */
#pragma unroll
                for (int k = 0; k < 9; k++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, SH_FV_MIX, 0);
                }
            }
#endif
            if (SH_FV_SB && (i % SH_FV_SB) == SH_FV_SB - 1) __builtin_amdgcn_sched_barrier(0);          /* quads one after the other: register budget */
        }
        if (active) { pstart = nstart; pend = nend; }
        if (a.hp_side && active) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int sq = (k * ((NH - 1) / 3)) >> 2;
                if (wave == (sq >> 5) && q == (sq & 3)) (a.hp_side + (hpo + t) * 5)[k] = hpv[k];
            }
        }
        {
            float ov = __shfl_xor(bv, 16); int oi = __shfl_xor(bi, 16);
            argmax_merge(bv, bi, ov, oi);
            ov = __shfl_xor(bv, 32); oi = __shfl_xor(bi, 32);
            argmax_merge(bv, bi, ov, oi);
            if (lane < 16) { redv[((par ^ 1) * NW + wave) * 16 + b] = bv; redi[((par ^ 1) * NW + wave) * 16 + b] = bi; }
        }
        if (more) {
            group_out(part, par ^ 1);
            stay_group(bp, par ^ 1);
            xp_publish(par);                            /* block t+2 (block t's pieces were last read a step ago) */
            xraw_load(t + 3);
        }
        VSTAMP(vC);
        __syncthreads();
        VSTAMP(vD);
        { float *x = cur; cur = nxt; nxt = x; }
    };
    for (int t = s0; t + 1 < s1; t++) block(t, std::true_type{});
    if (s1 > s0) block(s1 - 1, std::false_type{});

    if (s1 < Tt) {
        /* the tile's later blocks run on another workgroup: leave it the state */
        float *vst = a.vstate + (long long)tile * (NH * 16 + 32);
#pragma unroll
        for (int i = 0; i < PPT; i++) {
            const int Q = 32 * wave + 4 * i + q;
            *(f32x4 *)(vst + (Q * 16 + b) * 4) = *(const f32x4 *)(cur + (Q * 16 + b) * 4);
        }
        if (tid < 16) { vst[NH * 16 + b] = pstart; vst[NH * 16 + 16 + b] = pend; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(a.flag + tile, (unsigned)ord + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        /* argmaxf over nh+2 final scores, first maximum wins (decode.c:68, util.c:9) */
        float bv = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < PPT; i++) {
            const int Q = 32 * wave + 4 * i + q;
            const f32x4 pv = *(const f32x4 *)(cur + (Q * 16 + b) * 4);
#pragma unroll
            for (int k = 0; k < 4; k++) argmax_merge(bv, bi, pv[k], 4 * Q + k);
        }
        {
            float ov = __shfl_xor(bv, 16); int oi = __shfl_xor(bi, 16);
            argmax_merge(bv, bi, ov, oi);
            ov = __shfl_xor(bv, 32); oi = __shfl_xor(bi, 32);
            argmax_merge(bv, bi, ov, oi);
            __syncthreads();
            if (lane < 16) { redv[wave * 16 + b] = bv; redi[wave * 16 + b] = bi; }
        }
        __syncthreads();
        if (tid < 16) {
            float ev = redv[b]; int ei = redi[b];
            for (int w = 1; w < NW; w++) argmax_merge(ev, ei, redv[w * 16 + b], redi[w * 16 + b]);
            if (pstart > ev) { ev = pstart; ei = NH; }
            if (pend > ev) { ev = pend; ei = NH + 1; }
            a.final_state[rd] = ei;
            a.final_score[rd] = ev;
        }
    }
    if (a.dbg && lane == 0) { unsigned long long *d = a.dbg + ((long long)blockIdx.x * NW + wave) * 8; d[0] = vA; d[1] = vB; d[2] = vC; d[3] = vD; d[4] = (unsigned long long)(s1 - s0); }
}

/* viterbi_local_backtrace (decode.c:58-98), one thread per read */
__global__ void k_backtrace(const unsigned *__restrict__ tb, const int *__restrict__ tb_end,
                            const int *__restrict__ final_state, ShMeta md,
                            const long long *__restrict__ seq_off, int *__restrict__ seq,
                            int npad, int NQ) {
    const int rd = blockIdx.x * blockDim.x + threadIdx.x;
    if (rd >= npad) return;
    const int T = md.rT[rd];
    if (T <= 0) return;
    const int tile = rd >> 4, b = rd & 15;
    const long long boff = md.tile_boff[tile];
    const int NH = 4 * NQ;
    int *out = seq + seq_off[rd];
    const unsigned char *tbb = (const unsigned char *)tb;
    int last = final_state[rd];
    for (int ri = T - 1; ri >= 0; ri--) {
        int state;
        if (last < NH) {
            const unsigned code = tbb[(((boff + ri) * NQ + (last >> 2)) * 16 + b) * 4 + (last & 3)];
            if (code == SH_TB_STAY) state = -1;
            else if (code < SH_TB_SKIP) state = (int)(code - SH_TB_STEP) * (NH / 4) + (last >> 2);
            else if (code < SH_TB_SLIP) state = (int)(code - SH_TB_SKIP) * (NH / 16) + (last >> 4);
            else if (code < SH_TB_START) state = (int)(code - SH_TB_SLIP) * (NH / 64) + (last >> 6);
            else state = NH;
        } else if (last == NH) {
            state = NH;                                    /* decode.c:328 */
        } else {
            state = tb_end[(boff + ri) * 16 + b];
        }
        if (state >= 0) { out[ri + 1] = last; last = state; }
        else out[ri + 1] = -1;
    }
    out[0] = last;
    for (int i = 0; i < T; i++) { if (out[i] == NH) out[i] = -1; else break; }
    for (int i = T; i >= 0; i--) { if (out[i] == NH + 1) out[i] = -1; else break; }
}

/* ------------------------------------------------------------------ */
/* K1 + D4: globalnorm partition function, normalisation and the 5-state */
/* CRF Viterbi with traceback (layers.c:835-889, decode.c:836-893).      */
/* C holds the 25 transition scores in 2 chunks.  One tile of 16 reads   */
/* per 128-thread workgroup, 8 lanes per read: lane s < 5 owns the        */
/* transitions INTO state s (its row of 5 scores) and runs that state's   */
/* chain over the 5 source states in the reference's order; the 5 state   */
/* values cross lanes once per block.  (Round 1 ran one lane per read:    */
/* 25 dependent log-sum-exps per block on 157 waves, 4.95 ms per 10 000   */
/* reads x 800 blocks.)  Traceback: one byte per state and block.         */
/* ------------------------------------------------------------------ */
__global__ __launch_bounds__(128) void k_crf(float *__restrict__ C, ShMeta md,
                                             unsigned char *__restrict__ tbbuf /*[ncb][16][8]*/,
                                             const long long *__restrict__ seq_off,
                                             int *__restrict__ seq, float *__restrict__ score, int npad) {
    const int tile = blockIdx.x;
    const int b = threadIdx.x >> 3, st = threadIdx.x & 7;
    const int lane = threadIdx.x & 63, grp = lane & ~7;
    const int rd = tile * 16 + b;                  /* (npad is a whole number of tiles) */
    const int T = md.rT[rd];                       /* the 8 lanes of a read agree; the shuffles below stay inside them */
    const long long boff = md.tile_boff[tile];
    /* lane st < 5: elements 5 st .. 5 st + 4 of the column; lane 5: the three padding floats (kept normalised
     * like the rest, as the one-lane form did); lanes 6, 7 idle.  Element e of read b: chunk e >> 4, float
     * (((e >> 2) & 3) * 16 + b) * 4 + (e & 3). */
    const int ne = st < 5 ? 5 : (st == 5 ? 3 : 0);
    int eo[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int e = min(5 * st + k, 27);
        eo[k] = (e >> 4) * 256 + (((e >> 2) & 3) * 16 + b) * 4 + (e & 3);
    }
    auto fetch = [&](int t, float (&v)[5]) {
        const float *col = C + (boff + min(t, T - 1)) * 512;
#pragma unroll
        for (int k = 0; k < 5; k++) v[k] = (k < ne) ? col[eo[k]] : 0.0f;
    };
    auto gather = [&](float mine, float (&p)[5]) {
#pragma unroll
        for (int k = 0; k < 5; k++) p[k] = __shfl(mine, grp + k);
    };
    if (T <= 0) return;
    /* the column of block t + D is fetched while block t is worked on (a block's work is a few hundred cycles,
     * a global load several times that): a ring of D columns in registers */
    constexpr int D = 4;
    float q[D][5];
    float mine = 0.0f;
#pragma unroll
    for (int d = 0; d < D; d++) fetch(d, q[d]);
    for (int t0 = 0; t0 < T; t0 += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            if (t0 + d < T) {
                float tr[5];
#pragma unroll
                for (int k = 0; k < 5; k++) tr[k] = q[d][k];
                fetch(t0 + d + D, q[d]);
                float p[5];
                gather(mine, p);
                float acc = tr[0] + p[0];
#pragma unroll
                for (int s2 = 1; s2 < 5; s2++) acc = d_lse(acc, tr[s2] + p[s2]);
                mine = acc;
            }
        }
    }
    float p[5];
    gather(mine, p);
    float logZ = p[0];
#pragma unroll
    for (int s = 1; s < 5; s++) logZ = d_lse(logZ, p[s]);
    logZ = logZ / (float)T;                                 /* layers.c:879 */

    mine = 0.0f;
#pragma unroll
    for (int d = 0; d < D; d++) fetch(d, q[d]);
    for (int t0 = 0; t0 < T; t0 += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            const int t = t0 + d;
            if (t < T) {
                float tr[5];
                float *col = C + (boff + t) * 512;
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    tr[k] = q[d][k] - logZ;                     /* layers.c:881-886 */
                    if (k < ne) col[eo[k]] = tr[k];
                }
                fetch(t + D, q[d]);         /* (blocks t + D > t: never one already normalised) */
                gather(mine, p);
                float best = tr[0] + p[0];
                unsigned from = 0;
#pragma unroll
                for (int fr = 1; fr < 5; fr++) {
                    const float sc = tr[fr] + p[fr];
                    if (sc > best) { best = sc; from = fr; }   /* decode.c:873 */
                }
                mine = best;
                if (st < 5) tbbuf[((boff + t) * 16 + b) * 8 + st] = (unsigned char)from;
            }
        }
    }
    gather(mine, p);
    if (st != 0) return;
    /* final state, then the walk back by one lane per read.  (Its read's traceback bytes were written by lanes of
     * this same wave, earlier in program order.)  The eight bytes of a block are one 64-bit word whose address
     * does not depend on the path: words are fetched W blocks ahead, the dependent chain is a shift and a mask. */
    float best = p[0];
    int arg = 0;
#pragma unroll
    for (int s = 1; s < 5; s++) if (p[s] > best) { best = p[s]; arg = s; }
    score[rd] = best;
    int *out = seq + seq_off[rd];
    out[T] = arg;
    const unsigned long long *tb8 = (const unsigned long long *)tbbuf + boff * 16 + b;
    constexpr int W = 8;
    for (int blk0 = T; blk0 > 0; blk0 -= W) {
        unsigned long long w[W];
#pragma unroll
        for (int k = 0; k < W; k++) w[k] = tb8[(long long)max(blk0 - 1 - k, 0) * 16];
#pragma unroll
        for (int k = 0; k < W; k++) {
            if (blk0 - 1 - k >= 0) {
                arg = (int)((w[k] >> (8 * arg)) & 0xffull);
                out[blk0 - 1 - k] = arg;
            }
        }
    }
}

/* ------------------------------------------------------------------ */
__global__ __launch_bounds__(256) void k_inject_prob(const float *__restrict__ prob, const unsigned long long *__restrict__ poff /*[npad], ~0 = none*/,
                                                     ShMeta md, int NS, int mtiles, float *__restrict__ E, float *__restrict__ sums) {
    const int tile = blockIdx.x;
    const int Tt = md.tile_T[tile];
    const long long boff = md.tile_boff[tile];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = lane & 15, q = lane >> 4;
    const int rd = tile * 16 + b;
    const int myT = md.rT[rd];
    const unsigned long long off = poff[rd];
    for (int t = blockIdx.y; t < Tt; t += gridDim.y) {
        const bool live = t < myT && off != ~0ull;
        for (int mt = wave; mt < mtiles; mt += 4) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (live) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int st = 16 * mt + 4 * q + r;
                    if (st < NS) v[r] = prob[off + (unsigned long long)t * NS + st];
                }
            }
            *(f32x4 *)(E + ((boff + t) * mtiles + mt) * 256 + lane * 4) = v;
        }
        if (threadIdx.x < 16) sums[(boff + t) * 16 + threadIdx.x] = 1.0f;
    }
}

/* ------------------------------------------------------------------ */
/* layout converters for the per-read (reference-layout) surface         */
/* ------------------------------------------------------------------ */
/* chunked [cb][nchunk][256] of one read -> reference _Mat [t][stride]  */
__global__ void k_gather_read(const float *__restrict__ src, const float *__restrict__ sums,
                              long long boff, int b, int T, int nr, int nchunk, int out_stride,
                              int finalize, int want_log, float min_prob, float *__restrict__ dst) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)T * nr) return;
    const int t = (int)(idx / nr), m = (int)(idx % nr);
    float v = src[((boff + t) * nchunk + (m >> 4)) * 256 + (((m >> 2) & 3) * 16 + b) * 4 + (m & 3)];
    if (finalize) v = fin_post(v, 1.0f / sums[(boff + t) * 16 + b], min_prob, 1.0f - min_prob, want_log);
    dst[(long long)t * out_stride + m] = v;
}

#endif /* SH_KERNELS_H */
