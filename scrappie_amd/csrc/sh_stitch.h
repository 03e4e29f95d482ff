/* sh_stitch.h -- part of sh_kernels.h (included from there): D2 homopolymer correction + D3 k-mer stitching (and
 * crfpath_to_basecall) on the device, one thread per read, so that only BASES cross PCIe.
 * Device code for gfx950 only; see sh_kernels.h for conventions.
 *
 * Why: with realistic calls (~0.5 bases per block) the host form of these two steps costs 12.5 us per 800-block read
 * on one core of the GPU box (profiles/r3_host_stitch.txt): 125 ms per 10 000-read launch group single-threaded, 64 ms
 * on the two host threads an 8-rank job gets under the box's 16-CPU quota -- twice the 30 ms device step -- and it
 * needs the path (4 B per block) and five posterior rows (20 B per block) on the host.  Here the same integer logic
 * runs behind the traceback walk on the copy stream, under the next launch group's kernels, and the D2H shrinks
 * from 24 to at most 5 bytes per block.
 *
 * The algorithms are those of sh_host.c (homopolymer.c:67-235 on the five-row side buffer; decode.c:367-509;
 * decode.c:895-918) restated as ONE forward pass (see k_stitch for why that is exact).  One step is not integer work: the posterior-mean count of a run rounds sum(pr / (pr + ps))
 * of libm expf values (homopolymer.c:209-215, host: glibc).  The device evaluates the same expression with
 * exp() in double rounded to float (within an ulp of any libm's expf) and, whenever the sum lies within that
 * uncertainty (4e-7 per block of the run) of a rounding boundary, does NOT decide: it flags the read, and the host
 * re-stitches that read with the host code from the path and side rows still on the device (sh_host.c; ~never). */
#ifndef SH_STITCH_H
#define SH_STITCH_H

struct ShStitchArgs {
    const int *seq;               /* [nseq] Viterbi paths (k_backtrace / k_crf), read i at seq_off[i], T_i + 1 entries */
    const long long *seq_off;     /* [npad] */
    const float *hp;              /* [sum T][5] side rows or NULL (no homopolymer pass) */
    const long long *hp_off;      /* [npad] */
    int *pos;                     /* [nseq] or NULL: overlapper's pos[] (same offsets) */
    char *bases;                  /* [sum cap] */
    const long long *bases_off;   /* [npad] */
    int *blen;                    /* [npad] number of bases, -1: no call (every entry a stay) */
    unsigned *redo;               /* [npad] 1: the homopolymer mean sits on a rounding boundary -> host decides */
    int npad, nstate, crf;
    int sstride;                  /* ints between consecutive entries of a read in seq / work / pos (SH_SEQ_STRIDE; 1: contiguous) */
};

__device__ __forceinline__ int st_repeat_kmer(int b, int k) { int y = 0; for (int i = 0; i < k; i++) y = y * 4 + b; return y; }   /* scrappie_seq_helpers.c:115 */
__device__ __forceinline__ int st_kmer_shift(int k1, int k2, int nkmer) {       /* decode.c:367-401: smallest s >= 1 with suffix(k1) == prefix(k2) */
    /* (k <= 5: the five candidates side by side, no loop -- s = k always matches, both sides are empty) */
    const int m = nkmer - 1;
    int s = 5;
    s = ((k1 & (m >> 8)) == (k2 >> 8)) ? 4 : s;
    s = ((k1 & (m >> 6)) == (k2 >> 6)) ? 3 : s;
    s = ((k1 & (m >> 4)) == (k2 >> 4)) ? 2 : s;
    s = ((k1 & (m >> 2)) == (k2 >> 2)) ? 1 : s;
    return s;
}

/* One forward pass per read.  What makes that exact (proved from sh_host.c's two conditions, for ANY path):
 *  - a candidate at index i is decided by p = path[i-1] and q = path[i] alone, and its base is p & 3 (condition 1
 *    wants the last k-1 bases of p equal, condition 2 the last k-2 equal and the one before different: they exclude
 *    each other, and neither can hold for two bases);
 *  - runs are pairwise disjoint and no index inside a run (entries -1 / hk) nor the index after it is a candidate, so
 *    finding and applying runs in one sweep, in place, gives what collecting them base by base first and applying
 *    them afterwards gives (the order of application cannot matter);
 *  - every run owns its p (never a stay, never a homopolymer k-mer, so never inside a run): 2 nrun <= nblock, the
 *    reference's table of nblock / 2 runs never overflows.
 * The stitching consumes the corrected entries as they are produced; a run is replayed when its end is known. */
#ifndef SH_STITCH_VGPR_HALF
#define SH_STITCH_VGPR_HALF 40  /* at most 80 VGPRs (its natural size): fits beside three k_gru_proj waves of 144 on a SIMD */
#endif
/* (returns what it leaves in blen[rd]) */
__device__ __forceinline__ int stitch_read(const ShStitchArgs &a, const ShMeta &md, int rd) {
    const int T = md.rT[rd];
    if (T <= 0) { a.blen[rd] = -1; a.redo[rd] = 0; return -1; }
    const int *seq = a.seq + a.seq_off[rd];
    const long long ss = a.sstride;
#define SQ(x) seq[(long long)(x) * ss]
#define PS(x) pos[(long long)(x) * ss]
    unsigned *out32 = (unsigned *)(a.bases + a.bases_off[rd]);       /* (8-byte aligned: build_group) */
    int *pos = a.pos ? a.pos + a.seq_off[rd] : nullptr;
    /* bases are gathered four to a word: one store per four bases */
    unsigned word = 0;
    int nout = 0;
    auto emit = [&](int base) {
        word |= ((0x54474341u >> (8 * (base & 3))) & 0xffu) << (8 * (nout & 3));      /* 'A' 'C' 'G' 'T' */
        if ((++nout & 3) == 0) { out32[(nout >> 2) - 1] = word; word = 0; }
    };
    if (a.crf) {                                   /* decode.c:895-918: path entries below 4 are bases; pos untouched (Q11) */
        for (int i0 = 0; i0 < T; i0 += 8) {
            int v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = SQ(min(i0 + u, T - 1));
#pragma unroll
            for (int u = 0; u < 8; u++) if (i0 + u < T && v[u] < 4) emit(v[u]);
        }
        if (nout & 3) out32[nout >> 2] = word;
        a.blen[rd] = nout;
        a.redo[rd] = 0u;
        return nout;
    }
    const int nkmer = a.nstate - 1;
    int klen = 0;
    for (int x = nkmer; x > 1; x >>= 2) klen++;
    const int n = T + 1, nblock = T;
    const int fkm1 = 1 << (2 * (klen - 1)), fkm2 = 1 << (2 * (klen - 2));
    const bool hp_on = a.hp != nullptr && nblock / 2 > 0;
    const float *side = hp_on ? a.hp + a.hp_off[rd] * 5 : nullptr;

    /* D3 state (sh_host.c: overlapper; decode.c:449-509) */
    int prev = -1, pp = 0;
    auto put = [&](int k, int cur) {               /* entry k of the corrected path */
        if (cur >= 0) {
            if (prev < 0) { for (int i = klen - 1; i >= 0; i--) emit(cur >> (2 * i)); }
            else {
                const int s = st_kmer_shift(prev, cur, nkmer);
                pp += s;
                for (int i = s - 1; i >= 0; i--) emit(cur >> (2 * i));
            }
            prev = cur;
        }
        if (pos) PS(k) = pp;                         /* (zero up to and including the first k-mer, as calloc + pos[0] = 0 leave it) */
    };
    /* D2 state (sh_host.c: sh_homopolymer_side; homopolymer.c:95-138, :193-228) */
    enum { NORMAL = 0, STAYS2 = 1, RUN = 2 };
    int mode = NORMAL, p = -1, rb = 0, rhk = 0, rfrom = 0, rlen = 0, nvit = 0;
    double nmean = 0.0;
    bool redo = false;
    auto run_add = [&](int k, int cur) {
        const float *s = side + (long long)(k - 1) * 5;               /* block k-1 pairs with path[k] (Q8) */
        const double ps = (double)(float)exp((double)s[4]), pr = (double)(float)exp((double)s[rb]);
        nmean += pr / (pr + ps);
        nvit += (cur == rhk);
        rlen++;
    };
    auto run_start = [&](int k, int cur) { mode = RUN; rfrom = k; rlen = 0; nvit = 0; nmean = 0.0; run_add(k, cur); };
    auto run_finish = [&]() {
        const double r = nmean + 0.5;
        const int newn = (int)r;
        const double fr = r - (double)newn, tol = 4.0e-7 * (double)(rlen + 1);
        if (fr < tol || fr > 1.0 - tol || !(nmean == nmean)) redo = true;
        const bool change = newn != nvit;
        for (int i = 0; i < rlen; i++) put(rfrom + i, change ? (i < newn ? rhk : -1) : SQ(rfrom + i));
        mode = NORMAL;
    };
    auto step = [&](int k, int cur) {
        if (mode == RUN) {
            if (k < nblock && (cur == -1 || cur == rhk)) { run_add(k, cur); p = cur; return; }
            run_finish();
        } else if (mode == STAYS2) {
            if (k < nblock && cur == -1) { put(k, cur); p = cur; return; }
            mode = NORMAL;
            if (cur == rhk && k < nblock - 1) { run_start(k, cur); p = cur; return; }
        }
        if (hp_on && k >= 1 && k < nblock - 2 && p != -1) {
            const int b = p & 3;
            const int hk = st_repeat_kmer(b, klen), hk1 = hk % fkm1, hk2 = hk % fkm2;
            if (cur == -1 || cur == hk) {
                const bool c1 = p != hk && (p % fkm1) == hk1;
                const bool c2 = (p % fkm2) == hk2 && (p % fkm1) != hk1;
                if (c1 || (c2 && cur == hk)) { rb = b; rhk = hk; run_start(k, cur); p = cur; return; }
                if (c2) { rb = b; rhk = hk; mode = STAYS2; put(k, cur); p = cur; return; }
            }
        }
        put(k, cur);
        p = cur;
    };
    for (int k0 = 0; k0 < n; k0 += 8) {
        int v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = SQ(min(k0 + u, n - 1));       /* eight loads in flight */
#pragma unroll
        for (int u = 0; u < 8; u++) if (k0 + u < n) step(k0 + u, v[u]);
    }
    if (mode == RUN) run_finish();
    if (nout & 3) out32[nout >> 2] = word;
    a.redo[rd] = redo ? 1u : 0u;
    a.blen[rd] = prev < 0 ? -1 : nout;
#undef SQ
#undef PS
    return prev < 0 ? -1 : nout;
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(SH_STITCH_VGPR_HALF))) void k_stitch(ShStitchArgs a, ShMeta md) {
    const int rd = blockIdx.x * blockDim.x + threadIdx.x;
    if (rd >= a.npad) return;
    (void)stitch_read(a, md, rd);
}

/* Results of a launch group into pinned host memory, written by the device itself (a wave per read copies exactly that
 * read's bases; lane 0 the per-read words).  Not a copy-engine transfer: those are queued in order at enqueue time and
 * would sit, waiting for this group's decoder, in front of the NEXT group's uploads. */
struct ShResultArgs {
    const char *d_bases; char *h_bases; const long long *bases_off;     /* offsets are multiples of 16 */
    const int *d_blen; int *h_blen;
    const unsigned *d_redo; unsigned *h_redo;
    const float *d_score; float *h_score;
    const unsigned *d_bad; unsigned *h_bad;
    const unsigned *d_err; unsigned *h_err;
    int npad;
};
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(16))) void k_results_out(ShResultArgs a) {
    const int rd = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (rd >= a.npad) return;
    const int len = a.d_blen[rd];
    if (lane == 0) {
        a.h_blen[rd] = len; a.h_redo[rd] = a.d_redo[rd]; a.h_score[rd] = a.d_score[rd]; a.h_bad[rd] = a.d_bad[rd];
        if (rd == 0) *a.h_err = *a.d_err;
    }
    if (len <= 0) return;
    const u32x4 *src = (const u32x4 *)(a.d_bases + a.bases_off[rd]);
    u32x4 *dst = (u32x4 *)(a.h_bases + a.bases_off[rd]);
    for (int i = lane; i < (len + 15) / 16; i += 64) dst[i] = src[i];
}

/* The three steps behind the decoder as ONE kernel on the copy stream: a workgroup is one wave and owns 64 reads --
 * lane = read for the walk back and the stitching (the path goes through HBM as before: the walk writes it back to front, the stitching
 * reads it front to back), then the wave as a whole copies each of its reads' bases to pinned host memory (k_results_out's form). */
struct ShWalkArgs { const unsigned *tb; const int *tb_end; const int *final_state; const long long *seq_off; int *seq; int NQ; };
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(SH_STITCH_VGPR_HALF))) void k_walk_stitch_out(ShWalkArgs w, ShStitchArgs a, ShResultArgs r, ShMeta md) {
    const int rd0 = blockIdx.x * 64, lane = threadIdx.x;
    const int rd = rd0 + lane;
    int mylen = -1;
    if (rd < a.npad) {
        backtrace_read(w.tb, w.tb_end, w.final_state, md, w.seq_off, w.seq, rd, w.NQ, a.sstride);
        mylen = stitch_read(a, md, rd);
        r.h_blen[rd] = mylen; r.h_redo[rd] = a.redo[rd]; r.h_score[rd] = r.d_score[rd]; r.h_bad[rd] = r.d_bad[rd];
        if (rd == 0) *r.h_err = *r.d_err;
    }
    __syncthreads();                               /* the lanes' bases (global stores) are the wave's to read */
    const int nrd = min(64, a.npad - rd0);
    for (int k = 0; k < nrd; k++) {
        const int len = __shfl(mylen, k);
        if (len <= 0) continue;
        const u32x4 *src = (const u32x4 *)(r.d_bases + r.bases_off[rd0 + k]);
        u32x4 *dst = (u32x4 *)(r.h_bases + r.bases_off[rd0 + k]);
        for (int i = lane; i < (len + 15) / 16; i += 64) dst[i] = src[i];
    }
}

#endif /* SH_STITCH_H */
