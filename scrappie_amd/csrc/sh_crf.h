/* sh_crf.h -- part of sh_kernels.h (included from there, in this order): globalnorm + CRF Viterbi; decoder-input injection; layout converters.
 * Device code for gfx950 only; see sh_kernels.h for conventions (layouts, split products, citations). */
#ifndef SH_CRF_H
#define SH_CRF_H

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
#ifndef SH_CRF_D
#define SH_CRF_D 8          /* columns in flight per lane (forward and Viterbi passes) */
#endif
#ifndef SH_CRF_W
#define SH_CRF_W 16          /* traceback words in flight (walk back) */
#endif
/* WRITE: the normalised transitions go back to C (the posterior surface returns them); the basecall path only decodes them */
template <bool WRITE>
__global__ __launch_bounds__(128) void k_crf(float *__restrict__ C, ShMeta md,
                                             unsigned char *__restrict__ tbbuf /*[ncb][16][8]*/,
                                             const long long *__restrict__ seq_off,
                                             int *__restrict__ seq, float *__restrict__ score, int npad, int sstride) {
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
    /* Every vector-memory operation of the two block loops is UNCONDITIONAL (round 5).  With the loads under `k < ne` / `t < T` each sat in a
     * block of its own behind an s_cbranch_execz, the compiler could no longer count what is in flight and waited with vmcnt(0) on every block: the
     * ring of columns below never ran ahead, a block cost a whole global-memory round trip (1.70 ms per 10 000 reads x 800 blocks).  So: every lane
     * loads five floats it may read (lanes 5-7: the column's padding, slots 25-31), the loops run to the TILE's block count, and what a lane has no
     * use for is dropped by a select; stores go to slots nobody reads (padding slots; the blocks of this tile's columns past the read's own end). */
    const int Tt = md.tile_T[tile];                /* (uniform) */
    int eo[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int e = st < 5 ? 5 * st + k : (st == 5 ? 25 + k : 28 + ((st - 6) * 5 + k) % 4);      /* <= 31: inside the column's two chunks */
        eo[k] = (e >> 4) * 256 + (((e >> 2) & 3) * 16 + b) * 4 + (e & 3);
    }
    const int epad = 256 + (3 * 16 + b) * 4;       /* slots 28-31 of the read */
    auto fetch = [&](int t, float (&v)[5]) {
        const float *col = C + (boff + min(t, Tt - 1)) * 512;          /* (uniform) */
#pragma unroll
        for (int k = 0; k < 5; k++) v[k] = col[eo[k]];
    };
    auto gather = [&](float mine, float (&p)[5]) {
#pragma unroll
        for (int k = 0; k < 5; k++) p[k] = __shfl(mine, grp + k);
    };
    if (T <= 0) return;
    /* the column of block t + D is fetched while block t is worked on (a block's work is a few hundred cycles,
     * a global load several times that): a ring of D columns in registers */
    constexpr int D = SH_CRF_D;
    float q[D][5];
    float mine = 0.0f;
#pragma unroll
    for (int d = 0; d < D; d++) fetch(d, q[d]);
    for (int t0 = 0; t0 < Tt; t0 += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            float tr[5];
#pragma unroll
            for (int k = 0; k < 5; k++) tr[k] = (k < ne) ? q[d][k] : 0.0f;
            fetch(t0 + d + D, q[d]);
            float p[5];
            gather(mine, p);
            float acc = tr[0] + p[0];
#pragma unroll
            for (int s2 = 1; s2 < 5; s2++) acc = d_lse(acc, tr[s2] + p[s2]);
            mine = (t0 + d < T) ? acc : mine;
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
    for (int t0 = 0; t0 < Tt; t0 += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            const int t = t0 + d;
            /* a tile whose block count is no multiple of D: past its end every lane aims at padding of the last column -- float slots 28-31 and
             * traceback bytes 5-7 of its read -- (the column itself was fetched for these iterations BEFORE it was normalised) */
            const bool in_tile = t < Tt;            /* (uniform) */
            const int tc = min(t, Tt - 1);
            float tr[5];
            float *col = C + (boff + tc) * 512;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const float v = q[d][k] - logZ;                 /* layers.c:881-886 */
                if (WRITE) col[in_tile ? eo[k] : epad + (k & 3)] = v;
                tr[k] = (k < ne) ? v : 0.0f;
            }
            fetch(t + D, q[d]);             /* (blocks t + D > t: never one already normalised) */
            gather(mine, p);
            float best = tr[0] + p[0];
            unsigned from = 0;
#pragma unroll
            for (int fr = 1; fr < 5; fr++) {
                const float sc = tr[fr] + p[fr];
                if (sc > best) { best = sc; from = fr; }   /* decode.c:873 */
            }
            mine = (t < T) ? best : mine;
            tbbuf[((boff + tc) * 16 + b) * 8 + (in_tile ? st : 5 + (st & 1))] = (unsigned char)from;      /* (bytes 5-7 of a block's word and the blocks past the read's end are never looked at) */
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
    out[(long long)T * sstride] = arg;
    const unsigned long long *tb8 = (const unsigned long long *)tbbuf + boff * 16 + b;
    constexpr int W = SH_CRF_W;
    int blk = T;                                    /* blocks blk - 1 .. 0 are still to be resolved */
    /* the ragged head, T mod W blocks, one word at a time ... */
    for (int k = T % W; k > 0; k--) {
        blk--;
        arg = (int)((tb8[(long long)blk * 16] >> (8 * arg)) & 0xffull);
        out[(long long)blk * sstride] = arg;
    }
    /* ... then whole groups of W with nothing conditional inside: the next group's words are in flight while this group's chain runs */
    if (blk > 0) {
        unsigned long long w[W], wn[W];
#pragma unroll
        for (int k = 0; k < W; k++) w[k] = tb8[(long long)(blk - 1 - k) * 16];
        for (; blk > 0; blk -= W) {
            const int nb = max(blk - W, W);         /* (the last group fetches itself again) */
#pragma unroll
            for (int k = 0; k < W; k++) wn[k] = tb8[(long long)(nb - 1 - k) * 16];
#pragma unroll
            for (int k = 0; k < W; k++) {
                arg = (int)((w[k] >> (8 * arg)) & 0xffull);
                out[(long long)(blk - 1 - k) * sstride] = arg;
            }
#pragma unroll
            for (int k = 0; k < W; k++) w[k] = wn[k];
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

/* Launch-group metadata from pinned host memory into device memory by a KERNEL (the device reads the host buffer over
 * PCIe): a copy-engine upload queues behind the previous group's result copies, which wait for its decoder -- and with it
 * the whole prologue of the next group (convolution under the previous group's recurrent layers) would wait too. */
__global__ __launch_bounds__(256) void k_upload_words(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, long long n16) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) dst[i] = src[i];
}

/* scrappie_hip_set_trunk_input: caller-supplied trunk activations (row-major [nblock][S] per read) into the chunk
 * layout S1 reads; blocks past a read's end and padding reads are zero */
__global__ __launch_bounds__(256) void k_inject_trunk(const float *__restrict__ trunk, const unsigned long long *__restrict__ poff /*[npad], ~0 = none*/,
                                                      ShMeta md, int S, float *__restrict__ act) {
    const int tile = blockIdx.x;
    const int Tt = md.tile_T[tile];
    const long long boff = md.tile_boff[tile];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = lane & 15, q = lane >> 4;
    const int rd = tile * 16 + b;
    const int myT = md.rT[rd];
    const unsigned long long off = poff[rd];
    const int NU = S / 16;
    for (int t = blockIdx.y; t < Tt; t += gridDim.y) {
        const bool live = t < myT && off != ~0ull;
        for (int u = wave; u < NU; u += 4) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (live) v = *(const f32x4 *)(trunk + off + (unsigned long long)t * S + 16 * u + 4 * q);
            *(f32x4 *)(act + ((boff + t) * NU + u) * 256 + lane * 4) = v;
        }
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
    if (finalize) v = fin_post(v, d_rcp(sums[(boff + t) * 16 + b]), min_prob, 1.0f - min_prob, want_log);
    dst[(long long)t * out_stride + m] = v;
}

#endif /* SH_CRF_H */
