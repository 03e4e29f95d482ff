/* sh_lstm.h -- part of sh_kernels.h (included from there, in this order): events: feature columns and the peephole LSTM.
 * Device code for gfx950 only; see sh_kernels.h for conventions (layouts, split products, citations). */
#ifndef SH_LSTM_H
#define SH_LSTM_H

/* ------------------------------------------------------------------ */
/* (f).4  events: feature columns -> chunk layout, and the peephole LSTM  */
/* ------------------------------------------------------------------ */
/* feature3 columns (12 floats per event: networks.c:155-157) of the reads of a tile into one
 * 16-unit chunk per column block (units 12..15 zero; the weights are padded to match) */
__global__ __launch_bounds__(256) void k_feat_in(const float *__restrict__ feat, ShMeta md, int nfeat,
                                                 float *__restrict__ act, long long ncb_total, unsigned *__restrict__ bad /*[npad]*/) {
    bool out_of_range = false;
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
            /* operand range of the split products (see k_conv_act): studentised features stay below ~50 */
#pragma unroll
            for (int k = 0; k < 4; k++) {
                out_of_range |= !(__builtin_fabsf(v[k]) < SH_ACT_LIMIT);
                v[k] = (v[k] == v[k]) ? __builtin_amdgcn_fmed3f(v[k], -SH_ACT_LIMIT, SH_ACT_LIMIT) : 0.0f;
            }
        }
        *(f32x4 *)(act + (boff + t) * 256 + lane * 4) = v;
    }
    if (out_of_range && bad) bad[rd] = 1u;
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
            if (!sh_wait_flag(L.flag + tile, (unsigned)NU, L.flag + L.ntile) && lane == 0)      /* give up loudly instead of hanging the device */
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
/* L1 + lstm_step in one kernel (the layers whose input is as wide as   */
/* the state): k_gru_proj's two teams with ONE barrier per step -- the   */
/* LSTM has no second phase.  One tile of 16 reads per workgroup.  Wave  */
/* u of the recurrence team owns unit tile u of the four gates (96 VGPRs */
/* of sW pieces for S = 96) and the cell state; wave u of the projection */
/* team holds the same rows of iW, turns the input column of the NEXT    */
/* block into that block's gate inputs -- in accumulator units, in a     */
/* two-slot LDS ring -- and publishes the column after that as pieces.   */
/* The 4S gate inputs per read and block (12.3 GB written and read back  */
/* per layer and direction by k_affine + k_lstm_lanes) never exist in    */
/* HBM.  Lane schedule and hand-over (h and c) as in k_lstm_lanes.       */
/* ------------------------------------------------------------------ */
template <int NU, int NUI>      /* NUI: 16-row chunks of the layer input (NU, or 1 for the feature columns of the first level) */
__global__ __launch_bounds__(128 * NU) void k_lstm_proj(const float *__restrict__ in, float *__restrict__ out,
                                                       const unsigned *__restrict__ iWp, const float *__restrict__ ibfrag,
                                                       const unsigned *__restrict__ sWp, const float *__restrict__ pfrag,
                                                       ShMeta md, int backward, ShGruLanes L) {
    static_assert(NU % 2 == 0, "k steps of 32 units");
    constexpr int KS = NU / 2;
    constexpr int KSI = (NUI + 1) / 2;             /* k steps of the input contraction (a missing half chunk stays zero) */
    static_assert(NUI == NU || NUI == 1, "input as wide as the state, or one chunk");
    constexpr int PBUF = KS * 2 * 64 * 4;          /* one operand as fp16 pieces, in 32-bit words */
    constexpr int XBUF = 4 * NU * 256;             /* one block's gate inputs [gate][u][lane][4] */
    extern __shared__ __attribute__((aligned(16))) unsigned ldsw[];
    unsigned *lds_h = ldsw;                        /* [2][PBUF] */
    unsigned *lds_in = ldsw + 2 * PBUF;            /* [2][PBUF] */
    float *lds_x = (float *)(ldsw + 4 * PBUF);     /* [2][XBUF] */
    float *peep = lds_x + 2 * XBUF;                /* [3 * NU * 256] */
    const int lane = threadIdx.x & 63;
    const unsigned lofs = (unsigned)lane * 4u;
    typedef __attribute__((address_space(1))) float *gf32;
    typedef __attribute__((address_space(1))) f32x4 *gf32x4;
    auto gload = [&](const float *base) { gf32 b = (gf32)base; asm volatile("" : "+s"(b)); return *(gf32x4)(b + lofs); };
    auto gstore = [&](float *base, f32x4 v) { gf32 b = (gf32)base; asm volatile("" : "+s"(b)); *(gf32x4)(b + lofs) = v; };
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool rec = wave < NU;
    const int u = rec ? wave : wave - NU;

    if (NUI < 2 * KSI) {                            /* the half chunk nobody publishes: zero, once */
        for (int i = threadIdx.x; i < 2 * PBUF; i += 128 * NU) lds_in[i] = 0u;
        __syncthreads();
    }
    if (rec) {
#pragma unroll
        for (int g = 0; g < 3; g++)
            *(f32x4 *)(peep + ((g * NU + u) * 64 + lane) * 4) = *(const f32x4 *)(pfrag + ((long long)(g * NU + u) * 64 + lane) * 4);
    }
    const int wofs = (((u >> 1) * 2) * 64 + lane) * 4 + (u & 1) * 2;
    auto publish = [&](unsigned *buf, f32x4 v) {
        unsigned a1, a2, b1, b2;
        split_pair(v[0], v[1], a1, a2);
        split_pair(v[2], v[3], b1, b2);
        *(uint2 *)(buf + wofs) = make_uint2(a1, b1);
        *(uint2 *)(buf + wofs + 256) = make_uint2(a2, b2);
    };

    ShLaneCursor c = {};
    c.sgi = __builtin_amdgcn_readfirstlane(L.lane_off[blockIdx.x]);
    c.sge = __builtin_amdgcn_readfirstlane(L.lane_off[blockIdx.x + 1]);
    int nit = 0;
    for (int i = c.sgi; i < c.sge; i++) nit += L.seg[i].s1 - L.seg[i].s0;
    nit = __builtin_amdgcn_readfirstlane(nit);
    if (nit == 0) return;
    auto enter = [&](ShLaneCursor &cc) {
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
        ShSplit w[4][KSI];                          /* this wave's rows of iW as pieces */
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
            for (int ks = 0; ks < KSI; ks++) {
                w[g][ks] = load_pieces(iWp + ((long long)(g * NU + u) * KSI + ks) * 512, lane);
                asm volatile("" : "+v"(w[g][ks].p1), "+v"(w[g][ks].p2));
            }
        f32x4 bias[4];
#pragma unroll
        for (int g = 0; g < 4; g++) { bias[g] = *(const f32x4 *)(ibfrag + ((g * NU + u) * 64 + lane) * 4); asm volatile("" : "+v"(bias[g])); }
        auto fetch = [&]() {          /* unconditional (see k_gru_proj) */
            const long long col = c.ok ? column(c) : 0;
            const f32x4 v = gload(in + (col * NUI + (u < NUI ? u : 0)) * 256);
            if (c.ok) {
                c.s++;
                if (c.s == c.s1) { c.sgi++; enter(c); }
            }
            return v;
        };
        auto project = [&](const unsigned *ibuf, float *xdst) {
            ShSplit ip[KSI];
#pragma unroll
            for (int ks = 0; ks < KSI; ks++) ip[ks] = load_pieces(ibuf + ks * 512, lane);
            f32x4 a0 = bias[0], a1 = bias[1], a2 = bias[2], a3 = bias[3];
            split_dot2<KSI>(w[0], w[1], ip, a0, a1);
            split_dot2<KSI>(w[2], w[3], ip, a2, a3);
            *(f32x4 *)(xdst + ((0 * NU + u) * 64 + lane) * 4) = a0;
            *(f32x4 *)(xdst + ((1 * NU + u) * 64 + lane) * 4) = a1;
            *(f32x4 *)(xdst + ((2 * NU + u) * 64 + lane) * 4) = a2;
            *(f32x4 *)(xdst + ((3 * NU + u) * 64 + lane) * 4) = a3;
        };
        enter(c);
        f32x4 xq1, xq2;
        {
            const f32x4 x0 = fetch();
            xq1 = fetch(); xq2 = fetch();
            if (u < NUI) publish(lds_in, x0);
        }
        lds_barrier();
        project(lds_in, lds_x);                                /* block 0 */
        if (u < NUI) publish(lds_in + PBUF, xq1);              /* block 1 as pieces */
        xq1 = xq2;
        xq2 = fetch();
        lds_barrier();
        /* the queue is refilled in place, two steps per trip with the entries' roles fixed (see k_gru_proj: shifting it makes the compiler move the chunk
         * fetched a moment before at the loop's end, i.e. wait for it, on every step) */
        auto step = [&](int it, f32x4 &e) {
            const int np = (it + 1) & 1;
            if (u < NUI) publish(lds_in + (it & 1) * PBUF, e);       /* block it + 2 as pieces (block it's were last read a step ago) */
            e = fetch();                                             /* block it + 4 */
            project(lds_in + np * PBUF, lds_x + np * XBUF);    /* block it + 1 */
            lds_barrier();
        };
        for (int it = 0; it < nit; it += 2) {
            step(it, xq1);
            if (it + 1 < nit) step(it + 1, xq2);
        }
        return;
    }

    /* ---------------- recurrence team ---------------- */
    ShSplit w[4][KS];                               /* this wave's rows of sW as pieces */
#pragma unroll
    for (int g = 0; g < 4; g++)
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            w[g][ks] = load_pieces(sWp + ((long long)(g * NU + u) * KS + ks) * 512, lane);
            asm volatile("" : "+v"(w[g][ks].p1), "+v"(w[g][ks].p2));
        }
    int myT = 0;
    f32x4 h = {0.f, 0.f, 0.f, 0.f}, cs = h;
    auto take_over = [&]() {
        h = (f32x4){0.f, 0.f, 0.f, 0.f}; cs = h;
        myT = 0;
        if (!c.ok) return;
        myT = md.rT[c.tile * 16 + (lane & 15)];
        if (c.s > 0) {                              /* continuation of a tile begun on another lane */
            if (!sh_wait_flag(L.flag + c.tile, (unsigned)NU, L.flag + L.ntile) && lane == 0)
                __hip_atomic_store(L.flag + L.ntile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const float *hs = L.hstate + ((long long)c.tile * 2 * NU + u) * 256 + lane * 4;       /* [h | c] */
#pragma unroll
            for (int k = 0; k < 4; k++) {
                h[k] = __hip_atomic_load(hs + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                cs[k] = __hip_atomic_load(hs + NU * 256 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        asm volatile("" : "+v"(myT), "+v"(h), "+v"(cs));
    };
    enter(c);
    take_over();
    publish(lds_h, h);
    lds_barrier();                                  /* (prologue of the projection team) */
    lds_barrier();
    for (int it = 0; it < nit; it++) {
        const int par = it & 1;
        ShSplit hp[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ks++) hp[ks] = load_pieces(lds_h + par * PBUF + ks * 512, lane);
        const float *xs = lds_x + par * XBUF;
        f32x4 ai = *(const f32x4 *)(xs + ((0 * NU + u) * 64 + lane) * 4), au = *(const f32x4 *)(xs + ((1 * NU + u) * 64 + lane) * 4);
        f32x4 af = *(const f32x4 *)(xs + ((2 * NU + u) * 64 + lane) * 4), ao = *(const f32x4 *)(xs + ((3 * NU + u) * 64 + lane) * 4);
        /* the four gates' chains interleaved: a dependent MFMA never waits for its predecessor */
#pragma unroll
        for (int ks = 0; ks < KS; ks++) { ai = mfma16(w[0][ks].p1, hp[ks].p2, ai); au = mfma16(w[1][ks].p1, hp[ks].p2, au); af = mfma16(w[2][ks].p1, hp[ks].p2, af); ao = mfma16(w[3][ks].p1, hp[ks].p2, ao); }
#pragma unroll
        for (int ks = 0; ks < KS; ks++) { ai = mfma16(w[0][ks].p2, hp[ks].p1, ai); au = mfma16(w[1][ks].p2, hp[ks].p1, au); af = mfma16(w[2][ks].p2, hp[ks].p1, af); ao = mfma16(w[3][ks].p2, hp[ks].p1, ao); }
#pragma unroll
        for (int ks = 0; ks < KS; ks++) { ai = mfma16(w[0][ks].p1, hp[ks].p1, ai); au = mfma16(w[1][ks].p1, hp[ks].p1, au); af = mfma16(w[2][ks].p1, hp[ks].p1, af); ao = mfma16(w[3][ks].p1, hp[ks].p1, ao); }
        const int t = backward ? c.Tt - 1 - c.s : c.s;
        const bool active = t < myT;
        const f32x4 ti = d_tanh4_acc(ai);
        const f32x4 pu = *(const f32x4 *)(peep + (u * 64 + lane) * 4);
        const f32x4 pf = *(const f32x4 *)(peep + ((NU + u) * 64 + lane) * 4);
        const f32x4 po = *(const f32x4 *)(peep + ((2 * NU + u) * 64 + lane) * 4);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float forget = d_logistic(af[k] * SH_OINV + cs[k] * pf[k]) * cs[k];       /* layers.c:811-813 */
            const float update = d_logistic(au[k] * SH_OINV + cs[k] * pu[k]) * ti[k];       /* :815-817 */
            const float ns = forget + update;
            const float ho = d_logistic(ao[k] * SH_OINV + ns * po[k]) * d_tanh(ns);         /* :820-825 */
            cs[k] = active ? ns : 0.0f;
            h[k] = active ? ho : 0.0f;
        }
        gstore(out + ((long long)(c.boff + t) * NU + u) * 256, h);
        c.s++;
        if (c.s == c.s1) {                                   /* segment done */
            if (c.s1 < c.Tt) {                               /* the tile continues on another lane */
                float *hs = L.hstate + ((long long)c.tile * 2 * NU + u) * 256 + lane * 4;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    __hip_atomic_store(hs + k, h[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(hs + NU * 256 + k, cs[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                if (lane == 0) __hip_atomic_fetch_add(L.flag + c.tile, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            c.sgi++;
            enter(c);
            take_over();
        }
        publish(lds_h + (par ^ 1) * PBUF, h);
        lds_barrier();
    }
}

#endif /* SH_LSTM_H */
