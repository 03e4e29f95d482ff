/* sh_p0.h -- k_p0: signal preparation of a batch of reads on the device.
 *
 * What calculate_post does to a read before the network sees it (scrappie_raw.c:270-277):
 *     trim_and_segment_raw   scrappie_common.c:5-21   (trim_raw_by_mad :39-73, then the fixed trims)
 *     medmad_normalise_array util.c:190-205           (medianf :132-138, madf :156-180, quantilef :92-130)
 * The order statistics the reference gets from qsort are found here by selection, like the host form
 * (sh_host.c): a quantile needs the idx-th and (idx+1)-th smallest VALUES, which are the same whichever way
 * they are found; every floating-point expression is written as the reference has it (float / double, no
 * contraction: the library is built with -ffp-contract=off), so windows and samples are bit-identical.
 *
 * One workgroup of 256 threads per read.
 *   1. per chunk of `chunk` samples: MAD about the chunk's median.  A wave takes a chunk.  Up to 128 samples (the
 *      default is 100): a bitonic sort in registers, two values per lane.  Up to SH_P0_WCHUNK: every lane counts, for
 *      its elements v, #(u < v) and #(u <= v) over the chunk: the v whose interval holds position idx / idx + 1 are the
 *      order statistics; O(chunk^2 / 64) per wave.  No barrier either way.  Longer chunks go through the workgroup's
 *      radix selection instead.
 *   2. threshold = quantile of the chunk MADs (radix selection by the workgroup), leading / trailing chunks at
 *      or below it are cut off (index reductions), then trim_start / trim_end.
 *   3. median and MAD of the window by radix selection over the order-preserving 32-bit keys of the samples
 *      (four passes of 8 bits over an LDS histogram, |x - median| formed on the fly for the MAD), then
 *      x <- (x - median) / mad in place.
 * A read of up to SH_P0_STAGE samples is staged in LDS once (4000-sample reads: 16 KB) and every pass reads
 * it from there; longer reads are re-read from global memory (L2).  HBM-bound in principle (4 B in, 4 B out
 * per sample), latency- and LDS-atomic-bound in practice: ~0.3 ms for 10 000 x 4000 samples against 26 ms of
 * basecalling for the same reads.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef SH_P0_THREADS
#define SH_P0_THREADS 256
#endif
#ifndef SH_P0_STAGE
#define SH_P0_STAGE 8192        /* samples of a read kept in LDS */
#endif
#ifndef SH_P0_WCHUNK
#define SH_P0_WCHUNK 1024       /* chunk length up to which a wave ranks a chunk by counting */
#endif

struct ShP0Args {
    float *x;                       /* all reads' samples (raw in, normalised out inside each window) */
    const unsigned long long *off;  /* [n] first sample of read i (rt.raw) */
    const unsigned *len;            /* [n] rt.n */
    const unsigned *st0, *en0;      /* [n] rt.start, rt.end at entry */
    unsigned *win;                  /* [2n] out: start, end (end <= start: nothing left, trim_and_segment_raw's {0}) */
    float *scratch;                 /* as many floats as x: chunk MADs of read i at scratch + off[i] */
    unsigned trim_start, trim_end, chunk;
    float perc;
    unsigned nread;
};

__device__ __forceinline__ unsigned p0_key(float v) {
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float p0_unkey(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

/* util.c:117-125: position of quantile p among n sorted values */
__device__ __forceinline__ void p0_qpos(float p, unsigned long long n, unsigned long long &idx, float &remf) {
    const float t = p * (float)(n - 1);
    idx = (unsigned long long)t;
    remf = t - (float)idx;
}
__device__ __forceinline__ float p0_interp(float a, float b, float remf) {
    /* util.c:121: (1.0 - remf) * space[idx] + remf * space[idx + 1] -- the first product is double arithmetic, the SECOND a float product (both operands
     * float), rounded to float before it is added (as two double products the result differed by an ulp for remf other than 0 or 0.5) */
    return (float)((1.0 - (double)remf) * (double)a + (double)(remf * b));
}

/* The k-th and (k+1)-th smallest of v[i] = x[i] (ABS = false) or |x[i] - c| (ABS = true), i < m, by the whole workgroup.
 * hist: 264 words of LDS.  Every thread returns the same a and b (b = a when k is the last position). */
template <bool ABS>
__device__ __forceinline__ void p0_select(const float *x, unsigned m, unsigned k, float c, unsigned *hist, float &a, float &b) {
    const unsigned tid = threadIdx.x, nth = SH_P0_THREADS;
    unsigned prefix = 0, rem = k, cnt = 0;
    for (int pass = 0; pass < 4; pass++) {
        const int shift = 24 - 8 * pass;
        const unsigned himask = pass ? (0xffffffffu << (shift + 8)) : 0u;
        for (unsigned i = tid; i < 256; i += nth) hist[i] = 0;
        __syncthreads();
        /* neighbouring samples mostly share their leading digits: a thread adds runs, not single elements */
        unsigned run_b = 0xffffffffu, run_n = 0;
        for (unsigned i = tid; i < m; i += nth) {
            float v = x[i];
            if (ABS) v = fabsf(v - c);
            const unsigned key = p0_key(v + 0.0f);
            if ((key & himask) != prefix) continue;
            const unsigned bkt = (key >> shift) & 255u;
            if (bkt == run_b) run_n++;
            else { if (run_n) atomicAdd(&hist[run_b], run_n); run_b = bkt; run_n = 1; }
        }
        if (run_n) atomicAdd(&hist[run_b], run_n);
        __syncthreads();
        if (tid < 64) {
            const unsigned h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
            const unsigned s = h0 + h1 + h2 + h3;
            unsigned incl = s;
            for (int d = 1; d < 64; d <<= 1) {
                const unsigned o = __shfl_up(incl, d, 64);
                if ((int)tid >= d) incl += o;
            }
            const unsigned excl = incl - s;
            if (rem >= excl && rem < incl) {
                unsigned r = rem - excl, bkt, hc;
                if (r < h0) { bkt = 0; hc = h0; }
                else if ((r -= h0) < h1) { bkt = 1; hc = h1; }
                else if ((r -= h1) < h2) { bkt = 2; hc = h2; }
                else { r -= h2; bkt = 3; hc = h3; }
                hist[256] = 4 * tid + bkt; hist[257] = r; hist[258] = hc;
            }
        }
        __syncthreads();
        prefix |= hist[256] << shift;
        rem = hist[257];
        cnt = hist[258];
        __syncthreads();
    }
    a = p0_unkey(prefix);
    b = a;
    if (rem + 1 >= cnt && k + 1 < m) {        /* the next position holds the smallest value above a */
        if (tid == 0) hist[259] = 0xffffffffu;
        __syncthreads();
        unsigned best = 0xffffffffu;
        for (unsigned i = tid; i < m; i += nth) {
            float v = x[i];
            if (ABS) v = fabsf(v - c);
            const unsigned key = p0_key(v + 0.0f);
            if (key > prefix && key < best) best = key;
        }
        if (best != 0xffffffffu) atomicMin(&hist[259], best);
        __syncthreads();
        const unsigned bk = hist[259];
        if (bk != 0xffffffffu) b = p0_unkey(bk);
        __syncthreads();
    }
}

/* quantile p of x[0..m) (or of |x - c|) by the workgroup: util.c:92-130 for one quantile */
template <bool ABS>
__device__ __forceinline__ float p0_quantile(const float *x, unsigned m, float p, float c, unsigned *hist) {
    unsigned long long idx; float remf;
    p0_qpos(p, m, idx, remf);
    float a, b;
    p0_select<ABS>(x, m, (unsigned)idx, c, hist, a, b);
    return (idx < (unsigned long long)m - 1) ? p0_interp(a, b, remf) : a;
}

/* Quantile p of w[0..m) by ONE wave, m <= SH_P0_WCHUNK, w in LDS (16-byte aligned, readable up to the next multiple of 4: the caller
 * pads with +inf): rank by counting.  A value v occupies the sorted positions [#(u < v), #(u <= v)): the order statistic at position k
 * is the v whose interval holds k -- two compare-and-count instructions pairs per (u, v), no tie-breaking by index (equal values are
 * interchangeable).  A lane takes two elements at a time and walks the values four per LDS read (a broadcast ds_read_b128).
 * res: two words of LDS of this wave. */
__device__ __forceinline__ float p0_wave_quantile(const float *w, unsigned m, float p, float *res) {
    const unsigned lane = threadIdx.x & 63u;
    unsigned long long idx; float remf;
    p0_qpos(p, m, idx, remf);
    const unsigned k = (unsigned)idx, m4 = (m + 3u) & ~3u;
    for (unsigned e0 = lane; e0 < m; e0 += 128) {
        const unsigned e1 = e0 + 64;
        const bool has1 = e1 < m;
        const float v0 = w[e0], v1 = has1 ? w[e1] : INFINITY;
        unsigned lt0 = 0, le0 = 0, lt1 = 0, le1 = 0;
#pragma unroll 4
        for (unsigned j = 0; j < m4; j += 4) {
            const float4 u = *(const float4 *)(w + j);
            lt0 += (u.x < v0) + (u.y < v0) + (u.z < v0) + (u.w < v0);
            le0 += (u.x <= v0) + (u.y <= v0) + (u.z <= v0) + (u.w <= v0);
            lt1 += (u.x < v1) + (u.y < v1) + (u.z < v1) + (u.w < v1);
            le1 += (u.x <= v1) + (u.y <= v1) + (u.z <= v1) + (u.w <= v1);
        }
        if (lt0 <= k && k < le0) res[0] = v0;
        if (lt0 <= k + 1 && k + 1 < le0) res[1] = v0;
        if (has1 && lt1 <= k && k < le1) res[0] = v1;
        if (has1 && lt1 <= k + 1 && k + 1 < le1) res[1] = v1;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const float a = res[0], b = res[1];
    __builtin_amdgcn_wave_barrier();
    return (idx < (unsigned long long)m - 1) ? p0_interp(a, b, remf) : a;
}

/* Up to 128 values sorted by ONE wave in registers: element i of the bitonic network lives in lane i & 63, slot i >> 6 (v0: i < 64, v1: the rest;
 * positions past the chunk hold +inf and sort to the end).  28 compare-exchange steps, 27 of them through ds_bpermute (__shfl_xor), ~280
 * instructions where ranking 100 values by counting takes ~830.  Afterwards sorted position k is (lane k & 63, slot k >> 6). */
__device__ __forceinline__ void p0_wave_sort128(float &v0, float &v1) {
    const unsigned lane = threadIdx.x & 63u;
#pragma unroll
    for (unsigned k = 2; k <= 128; k <<= 1) {
#pragma unroll
        for (unsigned j = k >> 1; j >= 1; j >>= 1) {
            if (j == 64) {                       /* (k = 128: ascending everywhere) partner = the other slot of the same lane */
                const float lo = fminf(v0, v1), hi = fmaxf(v0, v1);
                v0 = lo; v1 = hi;
            } else {
                const float p0 = __shfl_xor(v0, (int)j, 64), p1 = __shfl_xor(v1, (int)j, 64);
                const bool lower = (lane & j) == 0;
                const bool up0 = (lane & k) == 0;                      /* slot 0: i = lane */
                const bool up1 = ((64u + lane) & k) == 0;              /* slot 1: i = 64 + lane */
                v0 = (lower == up0) ? fminf(v0, p0) : fmaxf(v0, p0);
                v1 = (lower == up1) ? fminf(v1, p1) : fmaxf(v1, p1);
            }
        }
    }
}
__device__ __forceinline__ float p0_sorted_at(float v0, float v1, unsigned k) {
    const float a = __shfl(v0, (int)(k & 63u), 64), b = __shfl(v1, (int)(k & 63u), 64);
    return (k >> 6) ? b : a;
}
/* median of a chunk of m <= 128 values (v0, v1 as above), util.c:117-125 / :132-138 */
__device__ __forceinline__ float p0_wave_median128(float v0, float v1, unsigned m) {
    unsigned long long idx; float remf;
    p0_qpos(0.5f, m, idx, remf);
    p0_wave_sort128(v0, v1);
    const float a = p0_sorted_at(v0, v1, (unsigned)idx);
    if (!(idx < (unsigned long long)m - 1)) return a;
    return p0_interp(a, p0_sorted_at(v0, v1, (unsigned)idx + 1), remf);
}

__global__ void __launch_bounds__(SH_P0_THREADS) k_p0(const ShP0Args A) {
    __shared__ __attribute__((aligned(16))) float stage[SH_P0_STAGE];
    __shared__ __attribute__((aligned(16))) float warea[SH_P0_THREADS / 64][SH_P0_WCHUNK];
    __shared__ float wres[SH_P0_THREADS / 64][2];
    __shared__ unsigned hist[264];
    __shared__ unsigned red[2];
    const unsigned tid = threadIdx.x, nth = SH_P0_THREADS, wave = tid >> 6, lane = tid & 63u, nwave = SH_P0_THREADS / 64;
    for (unsigned rd = blockIdx.x; rd < A.nread; rd += gridDim.x) {
        float *xg = A.x + A.off[rd];
        const unsigned n = A.len[rd];
        unsigned start = A.st0[rd], end = A.en0[rd];
        if (end > n) end = n;
        if (start > end) start = end;
        const bool staged = n <= SH_P0_STAGE;
        __syncthreads();
        if (staged) {
            for (unsigned i = tid; i < n; i += nth) stage[i] = xg[i];
            __syncthreads();
        }
        /* the rest of the read's preparation, once for samples in LDS and once for samples in global memory: with ONE pointer that may be
         * either, every access is a flat instruction; inlined per origin, the compiler knows the address space (ds_read / global_load) */
        auto body = [&](const float *xs) __attribute__((always_inline)) {
        /* --- trim_raw_by_mad (scrappie_common.c:39-73) --- */
        const unsigned cs = A.chunk;
        const unsigned nchunk = cs ? (end - start) / cs : 0;     /* chunk = 0 (a division by zero in the reference): no segmentation, the window stays */
        float *madarr = A.scratch + A.off[rd];
        if (cs) end = nchunk * cs;      /* relative to 0, as the reference */
        if (nchunk > 0) {
            if (cs <= 128 && cs > 1) {
                /* the default (100): a wave sorts the chunk in registers, twice (values, then absolute deviations) */
                for (unsigned c = wave; c < nchunk; c += nwave) {
                    const float *xc = xs + start + (size_t)c * cs;
                    const float x0 = lane < cs ? xc[lane] : INFINITY, x1 = 64 + lane < cs ? xc[64 + lane] : INFINITY;
                    const float med = p0_wave_median128(x0, x1, cs);
                    const float m2 = p0_wave_median128(lane < cs ? fabsf(x0 - med) : INFINITY, 64 + lane < cs ? fabsf(x1 - med) : INFINITY, cs);
                    if (lane == 0) madarr[c] = m2 * 1.4826f;
                }
            } else if (cs <= SH_P0_WCHUNK && cs > 1) {
                float *w = warea[wave];
                const unsigned cs4 = (cs + 3u) & ~3u;
                for (unsigned c = wave; c < nchunk; c += nwave) {
                    const float *xc = xs + start + (size_t)c * cs;
                    for (unsigned j = lane; j < cs4; j += 64) w[j] = j < cs ? xc[j] : INFINITY;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const float med = p0_wave_quantile(w, cs, 0.5f, wres[wave]);
                    for (unsigned j = lane; j < cs; j += 64) w[j] = fabsf(w[j] - med);      /* (a lane rewrites its own elements) */
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const float m2 = p0_wave_quantile(w, cs, 0.5f, wres[wave]);
                    if (lane == 0) madarr[c] = m2 * 1.4826f;
                }
            } else if (cs == 1) {
                for (unsigned c = tid; c < nchunk; c += nth) madarr[c] = 0.0f;       /* madf of one value (util.c:161) */
            } else {
                for (unsigned c = 0; c < nchunk; c++) {
                    const float *xc = xs + start + (size_t)c * cs;
                    const float med = p0_quantile<false>(xc, cs, 0.5f, 0.0f, hist);
                    const float m2 = p0_quantile<true>(xc, cs, 0.5f, med, hist);
                    if (tid == 0) madarr[c] = m2 * 1.4826f;
                }
            }
            __threadfence_block();
            __syncthreads();
            float thresh;
            if (nchunk <= SH_P0_WCHUNK) {        /* a few dozen values: one wave ranks them (no histogram passes), the others wait */
                float *w = warea[0];
                const unsigned n4 = (nchunk + 3u) & ~3u;
                for (unsigned j = tid; j < n4; j += nth) w[j] = j < nchunk ? madarr[j] : INFINITY;
                __syncthreads();
                if (wave == 0) { const float t = p0_wave_quantile(w, nchunk, A.perc, wres[0]); if (lane == 0) wres[1][0] = t; }
                __syncthreads();
                thresh = wres[1][0];
            } else thresh = p0_quantile<false>(madarr, nchunk, A.perc, 0.0f, hist);
            if (tid == 0) { red[0] = nchunk; red[1] = 0; }
            __syncthreads();
            unsigned first = nchunk, last1 = 0;        /* first chunk above the threshold; one past the last one above it */
            for (unsigned c = tid; c < nchunk; c += nth)
                if (madarr[c] > thresh) { if (c < first) first = c; if (c + 1 > last1) last1 = c + 1; }
            if (first < nchunk) { atomicMin(&red[0], first); atomicMax(&red[1], last1); }
            __syncthreads();
            start += red[0] * cs;
            end = red[1] * cs;
            __syncthreads();
        }
        /* --- fixed trims (scrappie_common.c:12-16) --- */
        start = (n - start) > A.trim_start ? start + A.trim_start : n;
        end = (end > A.trim_end) ? end - A.trim_end : 0;
        if (tid == 0) { A.win[2 * rd] = start; A.win[2 * rd + 1] = (start >= end) ? start : end; }
        if (start >= end) return;
        /* --- medmad_normalise_array (util.c:190-205) --- */
        const unsigned m = end - start;
        if (m == 1) { if (tid == 0) xg[start] = 0.0f; return; }
        const float xmed = p0_quantile<false>(xs + start, m, 0.5f, 0.0f, hist);
        const float xmad = p0_quantile<true>(xs + start, m, 0.5f, xmed, hist) * 1.4826f;
        for (unsigned i = tid; i < m; i += nth) xg[start + i] = (xs[start + i] - xmed) / xmad;
        };
        if (staged) body(stage);
        else body(xg);
    }
}
