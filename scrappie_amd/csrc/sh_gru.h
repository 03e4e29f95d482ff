/* sh_gru.h -- part of sh_kernels.h (included from there, in this order): the GRU layer: one tile per workgroup, lane-scheduled, split products, projection + recurrence in one kernel.
 * Device code for gfx950 only; see sh_kernels.h for conventions (layouts, split products, citations). */
#ifndef SH_GRU_H
#define SH_GRU_H

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

/* the lane schedule over PAIRS of tiles (k_gru_proj32, sh_gru32.h: an experiments-build kernel; the host builds the schedule either way) */
struct ShGruPairs {
    const int *lane_off;         /* [gridDim.x + 1] */
    const ShGruSegD *seg;        /* {pair, first step, end step, 0} (sh_sched.h over pairs, one lane per workgroup) */
    const int *pair_tile;        /* [npair][2]: the pair's tiles; second = -1: none */
    float *hstate;               /* [npair][3][16][64]: state handed from the lane that ran a pair's first steps */
    unsigned *flag;              /* [npair] arrival counters */
    unsigned *err;               /* the launch group's error word */
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
            if (!sh_wait_flag(L.flag + tile, (unsigned)NU, L.flag + L.ntile) && lane == 0)      /* give up loudly instead of hanging the device */
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
            if (!sh_wait_flag(L.flag + tile, (unsigned)NU, L.flag + L.ntile) && lane == 0)      /* give up loudly instead of hanging the device */
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

#ifndef SH_REC45_PRIO
#define SH_REC45_PRIO 0     /* s_setprio of recurrence waves 4, 5 alone: measured no effect at 1, 2, 3 */
#endif
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
#ifndef SH_PROJ_INPLACE
#define SH_PROJ_INPLACE 1   /* the projection team's queue of input chunks is refilled in place (two steps per trip) in every variant, not only where the layer computes
                               its own input: shifting it (`xq1 = xq2; xq2 = fetch()`) makes the compiler wait for the chunk fetched a moment before, on every step */
#endif
#ifndef SH_ABL
#define SH_ABL 0            /* timing ablations of k_gru_proj (tools/ab.sh); results are invalid unless 0 */
#endif
__device__ __forceinline__ f32x4 abl_logistic4(f32x4 a) { return (SH_ABL & 1) ? a * (0.25f * SH_OINV) + 0.5f : d_logistic4_acc(a); }
__device__ __forceinline__ f32x4 abl_tanh4(f32x4 a) { return (SH_ABL & 1) ? a * (0.5f * SH_OINV) : d_tanh4_acc(a); }

#ifndef SH_GRU_VGPR_HALF
#define SH_GRU_VGPR_HALF 72     /* amdgpu_num_vgpr counts pairs of registers on gfx90a+: 72 -> at most 144 VGPRs per wave (no scratch; the compiler takes
                                   164 of the 168 three waves per SIMD allow when left alone), so that three waves per SIMD (432 of 512) leave 80
                                   registers for helper waves -- k_backtrace, k_stitch, the next group's k_conv_act_bg, k_results_out -- which then run
                                   BESIDE a recurrent layer instead of delaying its workgroups.  80 (160): 27.97, 76: 27.49, 72: 27.51 ms per step */
#endif
/* Layer 1 computing its own input (round 3): the projection team turns the raw signal into the convolution's output
 * chunk by chunk, exactly as k_conv_mfma does -- the same MFMAs in the same order, the same partial windows, the same
 * activation: identical bits -- instead of fetching it from HBM.  KST = k steps of 4 taps (0: the input is read from
 * `in`), ACT = 0 elu / 1 tanh. */
#ifndef SH_FUSE_ABL
#define SH_FUSE_ABL 0       /* timing ablations of k_gru_conv (1: no sample loads, 2: no conv MFMAs, 4: no partial-window path, 8: no activation); results invalid unless 0 */
#endif
struct ShConvFuse {
    const float *sig;        /* the launch group's signals */
    const float *W;          /* [WL][F] taps */
    const float *bias;       /* [F] */
    const int *edge;         /* [npad][4] (host, run_pipeline): mask of the read's last 32 columns whose regular window does not exist
                                (bit j: column T - 1 - j), first column with a right-edge partial window, its w, N */
    unsigned *bad;           /* [npad] */
    ShConvGeom g;
};

#ifndef SH_RESID_LDS
#define SH_RESID_LDS 1     /* 0: the recurrence waves fetch the residual column themselves, a step ahead (the form before profiles/r5_resid_lds.txt; A/B builds) */
#endif
template <int NU, int NT, bool RESID, bool STAMP, int KST, int ACT>
__device__ __forceinline__ void gru_proj_body(const float *__restrict__ in, float *__restrict__ out,
                                              const float *__restrict__ resid,
                                              const unsigned *__restrict__ iWp, const float *__restrict__ ibfrag,
                                              const unsigned *__restrict__ sWp, const unsigned *__restrict__ sW2p,
                                              const ShMeta &md, int backward, const ShGruLanes &L,
                                              unsigned long long *dbg, const ShConvFuse &cf) {
    constexpr bool CONV = KST > 0;
    static_assert(NU % 2 == 0, "k steps of 32 units");
    constexpr int KS = NU / 2;
    constexpr int PBUF = KS * 2 * 64 * 4;          /* one operand as fp16 pieces, in 32-bit words: [ks][piece][lane][4] */
    constexpr int XBUF = 3 * NU * 256;             /* one block's gate inputs, accumulator layout [gate][u][lane][4] */
    constexpr int TBUF = 4 * PBUF + 2 * XBUF;      /* words per tile slot: h | r*h | in[2] | x[2] */
    /* residual layers: the projection wave that cuts unit tile u of a block's input column into pieces also leaves the column itself
     * (fp32, the output's own layout) in a ring of three blocks behind the tiles' slots, where recurrence wave u picks it up two steps
     * later -- instead of a second fetch from memory by the chain's wave (eight registers held across the whole step, and a load that a
     * step's time does not always cover).  Wherever the ring fits beside the slots (S = 96: 156 of 160 KB) */
    constexpr int RBUF = NU * 256;
    constexpr bool RLDS = RESID && !CONV && SH_RESID_LDS && (NT * (TBUF + 3 * RBUF) * 4 <= 160 * 1024);
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
    auto lds_raw = [&](int tl, int slot) { return (float *)(ldsw + NT * TBUF + (tl * 3 + slot) * RBUF) + (u * 64 + lane) * 4; };

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
        /* CONV: what the lane's read (b = lane & 15) of each tile slot needs to find its windows */
        int e_irr[NT], e_pw[NT], e_N[NT], e_T[NT];          /* e_pw: first partial-window column * 16 + its w */
        unsigned e_soff[NT];           /* the read's first sample in floats from cf.sig (the host takes this path below 2^30 only) */
        float ca[CONV ? KST : 1];      /* A operand of k step ks: tap 4 ks + (lane >> 4) of filter 16 u + (lane & 15) */
        f32x4 cb = {0.f, 0.f, 0.f, 0.f};    /* bias of filters 16 u + 4 (lane >> 4) .. + 3 */
        auto read_consts = [&](int tl) {
            if (CONV && c[tl].ok) {
                const int rd = c[tl].tile * 16 + (lane & 15);
                const int4 ev = ((const int4 *)cf.edge)[rd];
                e_irr[tl] = ev.x; e_pw[tl] = ev.y * 16 + ev.z; e_N[tl] = ev.w;
                e_T[tl] = md.rT[rd];
                e_soff[tl] = (unsigned)md.sig_off[rd];
            }
        };
        if (CONV) {
#pragma unroll
            for (int ks = 0; ks < (CONV ? KST : 1); ks++)
                ca[ks] = (4 * ks + (lane >> 4) < cf.g.WL) ? cf.W[(4 * ks + (lane >> 4)) * cf.g.F + 16 * u + (lane & 15)] : 0.0f;
            cb = *(const f32x4 *)(cf.bias + 16 * u + 4 * (lane >> 4));
#pragma unroll
            for (int tl = 0; tl < NT; tl++) { e_irr[tl] = 0; e_pw[tl] = 0; e_N[tl] = 0; e_T[tl] = 0; e_soff[tl] = 0u; }
        }
        /* a queue entry: the input chunk itself, or (CONV) the samples of its window as the B operand + what the
         * lane's read has to say about this column (0: past its end; 1: a column; w + 2: the partial window w ends here) */
        struct XQ { f32x4 v; float xs[CONV ? KST : 1]; int meta; int tile; };
        auto fetch = [&](ShLaneCursor &cc, int tl) {
            XQ x;
            x.meta = 0; x.tile = 0;
            if constexpr (!CONV) {
                const long long col = cc.ok ? column(cc) : 0;
                x.v = gload(in + (col * NU + u) * 256);
            } else {
                const int tb = cc.ok ? (backward ? cc.Tt - 1 - cc.s : cc.s) : 0;       /* the block within its tile */
                const bool live = cc.ok && tb < e_T[tl];
                const int j = e_T[tl] - 1 - tb;                                          /* >= 0 where live */
                const bool regular = live && !(j < 32 && ((e_irr[tl] >> (j & 31)) & 1));
                const int base = tb * cf.g.st - cf.g.padL + (lane >> 4);
                typedef __attribute__((address_space(1))) float *gsf;
                gsf sb = (gsf)cf.sig; asm volatile("" : "+s"(sb));
                int okbits = 0;
#pragma unroll
                for (int ks = 0; ks < KST; ks++) {
                    const int idx = base + 4 * ks;
                    const bool ok = regular && 4 * ks + (lane >> 4) < cf.g.WL && idx >= 0 && idx < e_N[tl];
                    /* unconditional (the same loads in flight on every path), and NOT looked at here: the value is masked where
                     * the chunk is built, two blocks later -- touching it now would wait for the load just issued, on every step */
                    x.xs[ks] = (SH_FUSE_ABL & 1) ? (float)idx : *(sb + (e_soff[tl] + (unsigned)(ok ? idx : 0)));
                    okbits |= ok ? (256 << ks) : 0;
                }
                const int jw = tb - (e_pw[tl] >> 4);
                const int wp = (e_pw[tl] & 15) + jw * cf.g.st;
                x.meta = (live ? ((jw >= 0 && wp < cf.g.padR) ? wp + 2 : 1) : 0) | okbits;     /* bits 0-7: the column; 8 + ks: sample ks is real */
                x.tile = cc.tile;
            }
            if (cc.ok) {
                cc.s++;
                if (cc.s == cc.s1) { cc.sgi++; enter(cc); read_consts(tl); }
            }
            return x;
        };
        /* CONV: the chunk of a queue entry, as k_conv_mfma computes it (sh_conv_affine.h) */
        auto chunk_of = [&](const XQ &x) {
            if constexpr (!CONV) return x.v;
            else {
                f32x4 acc = cb;
#pragma unroll
                for (int ks = 0; ks < KST; ks++) { if (SH_FUSE_ABL & 2) acc[0] += ca[ks] * x.xs[ks]; else acc = mfma4(ca[ks], ((x.meta >> (8 + ks)) & 1) ? x.xs[ks] : 0.0f, acc); }
                const int rd = x.tile * 16 + (lane & 15);
                const int colcode = x.meta & 255;
                if (!(SH_FUSE_ABL & 4) && colcode >= 2) {          /* rare: the lanes of reads that end here (layers.c:227-241) */
                    const int N = md.rN[rd], w = colcode - 2;
                    const float *xp = cf.sig + md.sig_off[rd] + (N - cf.g.WL + 1 + w);
                    const int f0 = 16 * u + 4 * (lane >> 4);
                    for (int tap = 0; tap < cf.g.WL - w - 1; tap++) {
                        const f32x4 wv = *(const f32x4 *)(cf.W + tap * cf.g.F + f0);
                        acc += wv * xp[tap];
                    }
                }
                if (colcode == 0) acc = f32x4{0.f, 0.f, 0.f, 0.f};
                else if (!(SH_FUSE_ABL & 8)) {
                    bool out_of_range = false;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float v = ACT ? d_tanh(acc[r]) : d_elu(acc[r]);
                        out_of_range |= !(__builtin_fabsf(v) < SH_ACT_LIMIT) | !(__builtin_fabsf(acc[r]) <= 3.0e38f);
                        acc[r] = (v == v) ? __builtin_amdgcn_fmed3f(v, -SH_ACT_LIMIT, SH_ACT_LIMIT) : 0.0f;
                    }
                    if (out_of_range && cf.bad) cf.bad[rd] = 1u;
                }
                return acc;
            }
        };
        XQ xq1[NT], xq2[NT];
        /* a block's 27 MFMAs: the candidate rows (9) in interval A, where the recurrence team issues 18 per
         * wave and tile, the update and reset rows (18) in interval B, where it issues 9 */
        /* (the affine kernels' order: bit-identical to them; the gate inputs stay in accumulator units) */
        auto project_h = [&](const unsigned *ibuf, float *xdst) {       /* (straight into the ring: nobody reads the slot of block it + 1 before the barrier) */
            ShSplit ip[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ks++) ip[ks] = pieces(ibuf, ks);
            const f32x4 dst = (SH_ABL & 8) ? bh : split_dot<KS>(w2, ip, bh);
            *(f32x4 *)(xdst + ((2 * NU + u) * 64 + lane) * 4) = dst;
            if ((SH_ABL & 16) && !(u == 2 || u == 3)) {     /* timing emulation of a 4 : 1 split of the projection's m-tiles between the waves of SIMDs 2, 3 and 0, 1 */
                const f32x4 extra = split_dot<KS>(w0, ip, bh);
                asm volatile("" :: "v"(extra));
            }
        };
        auto project_zr = [&](const unsigned *ibuf, float *xdst) {
            f32x4 cz = bz, cr = br;
            if (!(SH_ABL & 8) && !((SH_ABL & 16) && (u == 2 || u == 3))) {
                ShSplit ip[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ks++) ip[ks] = pieces(ibuf, ks);
                split_dot2<KS>(w0, w1, ip, cz, cr);
            }
            *(f32x4 *)(xdst + (u * 64 + lane) * 4) = cz;
            *(f32x4 *)(xdst + ((NU + u) * 64 + lane) * 4) = cr;
        };
        /* prologue: block 0's gate inputs, block 1 as pieces */
#pragma unroll
        for (int tl = 0; tl < NT; tl++) {
            enter(c[tl]);
            read_consts(tl);
            const XQ xin = fetch(c[tl], tl);
            xq1[tl] = fetch(c[tl], tl); xq2[tl] = fetch(c[tl], tl);
            publish(lds_in(tl, 0), chunk_of(xin));
            if constexpr (RLDS) *(f32x4 *)lds_raw(tl, 0) = chunk_of(xin);
        }
        lds_barrier();
#pragma unroll
        for (int tl = 0; tl < NT; tl++) {
            project_h(lds_in(tl, 0), lds_x(tl, 0));
            project_zr(lds_in(tl, 0), lds_x(tl, 0));
            publish(lds_in(tl, 1), chunk_of(xq1[tl]));
            if constexpr (RLDS) *(f32x4 *)lds_raw(tl, 1) = chunk_of(xq1[tl]);
            xq1[tl] = xq2[tl];
            xq2[tl] = fetch(c[tl], tl);
        }
        lds_barrier();
        if (STAMP) pt0 = __builtin_readcyclecounter();
        int wslot = 2;                                  /* (block it + 2) % 3 */
        if constexpr (CONV || SH_PROJ_INPLACE) {
            /* The queue does not shift here: the entry just turned into pieces is refilled in place (two steps per trip, the
             * entries' roles fixed at compile time).  With `xq1 = xq2; xq2 = fetch()` the compiler kept the freshly loaded
             * samples in other registers and moved them at the loop's end -- a wait for loads issued a moment before, on
             * every step (4.3 instead of 2.9 ms for the layer). */
            auto step = [&](int it, const int par, XQ (&e)[NT]) {      /* par = it & 1 */
                const int np = par ^ 1;
#pragma unroll
                for (int tl = 0; tl < NT; tl++) project_h(lds_in(tl, np), lds_x(tl, np));          /* interval A: block it + 1 */
                PSTAMP(pa);
                lds_barrier();
                PSTAMP(pb);
#pragma unroll
                for (int tl = 0; tl < NT; tl++) {
                    publish(lds_in(tl, par), chunk_of(e[tl]));                                     /* block it + 2 as pieces */
                    if constexpr (RLDS) *(f32x4 *)lds_raw(tl, wslot) = chunk_of(e[tl]);
                    e[tl] = fetch(c[tl], tl);                                                      /* block it + 4 */
                }
                if constexpr (RLDS) wslot = wslot == 2 ? 0 : wslot + 1;
                __builtin_amdgcn_sched_barrier(0);
                if (STAMP) { qt1 = __builtin_readcyclecounter(); q1 += qt1 - pt0; }               /* (projection waves: the chunk + fetch part of B) */
#pragma unroll
                for (int tl = 0; tl < NT; tl++) project_zr(lds_in(tl, np), lds_x(tl, np));         /* interval B */
                PSTAMP(pc);
                lds_barrier();
                PSTAMP(pd);
            };
            for (int it = 0; it < nit; it += 2) {
                step(it, 0, xq1);
                if (it + 1 < nit) step(it + 1, 1, xq2);
            }
        } else
        for (int it = 0; it < nit; it++) {
            const int np = (it + 1) & 1;
            if (SH_PDELAY_A) __builtin_amdgcn_s_sleep(SH_PDELAY_A);
#pragma unroll
            for (int tl = 0; tl < NT; tl++) project_h(lds_in(tl, np), lds_x(tl, np));          /* interval A: block it + 1 */
            PSTAMP(pa);
            lds_barrier();
            PSTAMP(pb);
            if (SH_PROJ_VALU_FIRST) {
#pragma unroll
                for (int tl = 0; tl < NT; tl++) {
                    publish(lds_in(tl, it & 1), chunk_of(xq1[tl]));                            /* block it + 2 as pieces */
                    if constexpr (RLDS) *(f32x4 *)lds_raw(tl, wslot) = chunk_of(xq1[tl]);
                    xq1[tl] = xq2[tl];
                    xq2[tl] = fetch(c[tl], tl);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (SH_PDELAY_B) __builtin_amdgcn_s_sleep(SH_PDELAY_B);
#pragma unroll
            for (int tl = 0; tl < NT; tl++) project_zr(lds_in(tl, np), lds_x(tl, np));           /* interval B */
            if (!SH_PROJ_VALU_FIRST) {
#pragma unroll
                for (int tl = 0; tl < NT; tl++) {
                    publish(lds_in(tl, it & 1), chunk_of(xq1[tl]));                            /* block it + 2 as pieces */
                    if constexpr (RLDS) *(f32x4 *)lds_raw(tl, wslot) = chunk_of(xq1[tl]);
                    xq1[tl] = xq2[tl];
                    xq2[tl] = fetch(c[tl], tl);
                }
            }
            if constexpr (RLDS) wslot = wslot == 2 ? 0 : wslot + 1;
            PSTAMP(pc);
            lds_barrier();
            PSTAMP(pd);
        }
        PDUMP();
        return;
    }

    /* ---------------- recurrence team ---------------- */
    if (SH_REC_PRIO) __builtin_amdgcn_s_setprio(SH_REC_PRIO);
    /* issue arbitration is by priority, then by age: the SECOND recurrence wave of SIMDs 0 / 1 (waves 4, 5; the younger one)
     * finishes its intervals 500-900 cycles after the first (stamps) */
    if (SH_REC45_PRIO && NU == 6 && wave >= 4) __builtin_amdgcn_s_setprio(SH_REC45_PRIO);
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
            if (!sh_wait_flag(L.flag + c[tl].tile, (unsigned)NU, L.flag + L.ntile) && lane == 0)      /* give up loudly instead of hanging the device */
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
    /* element offset of the column a tile slot is at (the same in the layer's input and output): kept up by one scalar addition
     * per step instead of being rebuilt from the cursor (a dozen scalar instructions per tile and step on the chain's wave) */
    long long oix[NT];
    const long long dstep = backward ? -(long long)(NU * 256) : (long long)(NU * 256);
    auto cur_off = [&](int tl) {
        const long long v = c[tl].ok ? (column(c[tl]) * NU + u) * 256 : (long long)(u * 256);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (long long)(((unsigned long long)hi << 32) | lo);                       /* (wave-uniform: scalar registers) */
    };
#pragma unroll
    for (int tl = 0; tl < NT; tl++) {
        enter(c[tl]);
        take_over(tl);
        oix[tl] = cur_off(tl);
        publish(lds_h(tl), h[tl]);
    }
    lds_barrier();                                  /* (prologue of the projection team) */
    lds_barrier();
    if (STAMP) pt0 = __builtin_readcyclecounter();
    /* rnnrf (networks.c:583): the layer's input column is added to its output; it comes through the LDS ring (RLDS, above),
     * else this wave fetches it a step ahead */
    f32x4 rs[NT];
    auto resid_fetch = [&](int tl) {
        rs[tl] = gload(resid + oix[tl]);
    };
    if (RESID && !RLDS) {
#pragma unroll
        for (int tl = 0; tl < NT; tl++) resid_fetch(tl);
    }
    int rslot = 0;                                  /* it % 3 */
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
            if constexpr (RLDS) rs[tl] = *(const f32x4 *)lds_raw(tl, rslot);            /* (behind the candidate's products, in front of the tanh) */
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
                const long long oidx = RESID ? oix[tl] : ((long long)(c[tl].boff + t) * NU + u) * 256;       /* uniform */
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
                    if (RESID) oix[tl] = cur_off(tl);
                } else if (RESID) oix[tl] += dstep;
            }
            if (STAMP) __builtin_amdgcn_sched_barrier(0);
            QSTAMP(q4);                                        /* output store, lane bookkeeping */
            if (RESID && !RLDS) resid_fetch(tl);                                       /* the next step's column */
            publish(lds_h(tl), h[tl]);
            if (STAMP) __builtin_amdgcn_sched_barrier(0);
            QSTAMP(q5);                                        /* cut into pieces + LDS write */
        }
        if constexpr (RLDS) rslot = rslot == 2 ? 0 : rslot + 1;
        PSTAMP(pc);
        lds_barrier();
        PSTAMP(pd);
    }
    PDUMP();
#undef PSTAMP
#undef PDUMP
#undef QSTAMP
}

template <int NU, int NT, bool RESID, bool STAMP = false>
__global__ __launch_bounds__(128 * NU) __attribute__((amdgpu_num_vgpr(SH_GRU_VGPR_HALF))) void k_gru_proj(const float *__restrict__ in, float *__restrict__ out,
                                                       const float *__restrict__ resid,
                                                       const unsigned *__restrict__ iWp, const float *__restrict__ ibfrag,
                                                       const unsigned *__restrict__ sWp, const unsigned *__restrict__ sW2p,
                                                       ShMeta md, int backward, ShGruLanes L,
                                                       unsigned long long *dbg = nullptr) {
    gru_proj_body<NU, NT, RESID, STAMP, 0, 0>(in, out, resid, iWp, ibfrag, sWp, sW2p, md, backward, L, dbg, ShConvFuse{});
}

/* The residual layers of rnnrf_r94 (networks.c:583-607: every layer's output + its input).  The residual column costs the
 * recurrence waves four more live registers per tile.  The compiler keeps 16 bytes of scratch per lane for this variant whatever
 * the cap (144 ... 168 registers: 1-5 values spilled; fetching the column one interval ahead instead of one step: 53): the
 * values are stored once in front of the step loop and reloaded only on the segment-change path (a tile's end: once per ~800
 * steps) -- the 10 scratch instructions of round 4's disassembly, none of them in the steady-state step.  The four extra
 * registers of the 148-register cap (3 x 148 = 444 of the SIMD's 512: the 64-register helper kernels still fit beside) are
 * what the layer gained: 3.37 -> 3.23 ms per launch (profiles/r5_bench_rnnrf_r94.json). */
#ifndef SH_GRU_RES_VGPR_HALF
#define SH_GRU_RES_VGPR_HALF 74
#endif
template <int NU, int NT>
__global__ __launch_bounds__(128 * NU) __attribute__((amdgpu_num_vgpr(SH_GRU_RES_VGPR_HALF))) void k_gru_proj_res(const float *__restrict__ in, float *__restrict__ out,
                                                       const float *__restrict__ resid,
                                                       const unsigned *__restrict__ iWp, const float *__restrict__ ibfrag,
                                                       const unsigned *__restrict__ sWp, const unsigned *__restrict__ sW2p,
                                                       ShMeta md, int backward, ShGruLanes L) {
    gru_proj_body<NU, NT, true, false, 0, 0>(in, out, resid, iWp, ibfrag, sWp, sW2p, md, backward, L, nullptr, ShConvFuse{});
}

#ifdef SH_EXPERIMENTS
/* the first layer of the rgrgr models with the convolution inside (96 filters = 96 units; KST * 4 >= WL taps).  No helper
 * kernel has to fit beside it (the traceback walk and k_stitch of the previous group run beside the layers after it), so
 * it may take the 168 VGPRs three waves per SIMD allow */
#ifndef SH_GRU_CONV_VGPR_HALF
#define SH_GRU_CONV_VGPR_HALF 84
#endif
template <int NT, int KST, int ACT>
__global__ __launch_bounds__(768) __attribute__((amdgpu_num_vgpr(SH_GRU_CONV_VGPR_HALF))) void k_gru_conv(float *__restrict__ out,
                                                       const unsigned *__restrict__ iWp, const float *__restrict__ ibfrag,
                                                       const unsigned *__restrict__ sWp, const unsigned *__restrict__ sW2p,
                                                       ShMeta md, int backward, ShGruLanes L, ShConvFuse cf) {
    gru_proj_body<6, NT, false, false, KST, ACT>(nullptr, out, nullptr, iWp, ibfrag, sWp, sW2p, md, backward, L, nullptr, cf);
}

template <int NT, int KST, int ACT>
__global__ __launch_bounds__(768) __attribute__((amdgpu_num_vgpr(SH_GRU_CONV_VGPR_HALF))) void k_gru_conv_stamp(float *__restrict__ out,
                                                       const unsigned *__restrict__ iWp, const float *__restrict__ ibfrag,
                                                       const unsigned *__restrict__ sWp, const unsigned *__restrict__ sW2p,
                                                       ShMeta md, int backward, ShGruLanes L, ShConvFuse cf, unsigned long long *dbg) {
    gru_proj_body<6, NT, false, true, KST, ACT>(nullptr, out, nullptr, iWp, ibfrag, sWp, sW2p, md, backward, L, dbg, cf);
}

#endif /* SH_EXPERIMENTS */

#endif /* SH_GRU_H */
