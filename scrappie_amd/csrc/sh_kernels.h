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

/* v summed over the four 16-lane rows of a wave (lane = 16 q + b: the four q of one read), the result in every lane.
 * (Round 6 tried ds_swizzle_b32 + v_permlane32_swap_b32 in place of the two shuffles -- each __shfl_xor is a ds_bpermute_b32 behind seven VALU instructions
 * that form the source lane: 0.7 % of k_ff_viterbi_teams -- and the form did not reproduce the shuffle form's traceback on the device; not kept.) */
__device__ __forceinline__ float rows_sum(float v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

/* Wait until *flag >= need (relaxed agent-scope polls, one every ~1 us).  The wait is bounded by WALL time
 * (s_memrealtime, 100 MHz), not by a poll count: a producer workgroup that is merely late (shared or
 * pre-empted device, skewed dispatch) is waited for; after SH_HANDOVER_TIMEOUT_S seconds the caller raises
 * the launch group's error word and the host re-runs the group on whole tiles (scrappie_hip_collect). */
#ifndef SH_HANDOVER_TIMEOUT_S
#define SH_HANDOVER_TIMEOUT_S 20ull
#endif
/* `err` is the launch group's error word: once any waiter has given up, every other one returns at its next poll
 * instead of sitting out its own timeout on a flag that will never be raised (the group is re-run anyway). */
__device__ __forceinline__ bool sh_wait_flag(const unsigned *flag, unsigned need, const unsigned *err) {
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) return true;
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        __builtin_amdgcn_s_sleep(32);
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) return true;
        if (err && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
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
 * them, and convolution outputs of med/MAD-normalised signal; both limits are checked (SH_W_LIMIT, SH_ACT_LIMIT
 * below).  A lane holds the
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
/* operand range of the pieces (fp16's largest finite value is 65504): enforced, not assumed -- weights at model load
 * (scrappie_hip_load_model: a recurrent layer with |w| >= SH_W_LIMIT runs on the exact-fp32 kernels, any other
 * such matrix is refused), activations where they enter the network (k_conv_act, k_feat_in: the read is flagged
 * and gets no call).  Gate outputs lie in (-1, 1); residual sums add at most 5 to the first layer's input. */
#define SH_W_LIMIT 255.0f
#define SH_ACT_LIMIT 1000.0f
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
/* ... where the caller has shown |a| x 2^-14 < 88 (the clamp is the identity) */
__device__ __forceinline__ float d_exp_acc_inrange(float a) {
#if SH_FAST_MATH
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

/* Decoded paths (and k_stitch's pos[]) of a launch group lie TILE-INTERLEAVED in HBM: entry t of read b of a tile at
 * tile base + t * SH_SEQ_STRIDE + b, so that the one-thread-per-read kernels behind the decoder (k_backtrace,
 * k_stitch, k_crf's walk back) touch 4 cache lines per wave and entry instead of 64.  (The per-read surface passes
 * a stride of 1.) */
#define SH_SEQ_STRIDE 16

/* the kernels, by stage */
#include "sh_conv_affine.h"
#include "sh_gru.h"
#ifdef SH_EXPERIMENTS      /* kernel forms measured and not adopted: only in libscrappie_hip_exp.so, for the tests that compare them */
#include "sh_gru_mx.h"
#include "sh_gru32.h"
#include "sh_gru32x2.h"
#include "sh_gru_free.h"
#endif
#include "sh_lstm.h"
#include "sh_s1.h"
#include "sh_decode.h"
#include "sh_decode_teams.h"
#include "sh_crf.h"
#include "sh_stitch.h"

#endif /* SH_KERNELS_H */
