/* sh_gru_mx.h -- part of sh_kernels.h (included behind sh_gru.h): k_gru_mx, one recurrent layer (projection + recurrence, S = 96) with the
 * matrix work and the elementwise work on DIFFERENT waves.  Device code for gfx950 only; conventions, layouts and citations as sh_gru.h.
 *
 * k_gru_proj's budget (profiles/r5_gru_stamps.txt): the wave that owns a unit tile's chain issues its MFMAs and its activations in ONE
 * in-order instruction stream -- it sits in front of MFMAs it cannot issue (the pipe is shared three ways) and only then starts on the
 * logistic / cut / publish of the values it already has; matrix pipe 46 % busy.  Here, per workgroup of two 16-read tiles:
 *
 *   waves 0-5    RECURRENCE products only: wave u holds the update / reset / candidate rows of unit tile u (72 VGPRs of fp16 pieces), starts
 *                its accumulators from the gate inputs in the LDS ring and writes the finished pre-activations back IN PLACE
 *   waves 6-11   PROJECTION products only: wave u holds the rows of iW for unit tile u and turns the NEXT block's input pieces into gate inputs
 *                (the ring's other slot), spread over the phases in which the recurrence waves of its SIMD leave the pipe alone
 *   waves 12-15  everything else, no MFMA: wave v owns tile v >> 1, unit tiles 3 (v & 1) .. + 2: the fp32 state, logistic(r) * h, logistic(z),
 *                tanh, blend, the output store, the cut into fp16 pieces of h / r*h / the input column, the lane schedule's bookkeeping and
 *                the hand-over of a tile's state between workgroups
 *
 * Four LDS-only barriers per step (16 waves):
 *   P1  rec: r and z products of block t       proj: candidate rows of t + 1 (waves alone on their SIMD's recurrence side)   elementwise: input column t + 2 -> pieces, fetch t + 4
 *   P2  elementwise: logistic(r) * h -> pieces  proj: update / reset rows, tile 0
 *   P3  rec: candidate products on r * h        proj: candidate rows, tile 1                                                    elementwise: logistic(z)
 *   P4  elementwise: tanh, blend, store, h -> pieces                                                                             proj: update / reset rows, tile 1
 * Same operations on the same values in the same order as k_gru_proj (and as k_affine_lds + k_gru_split): identical bits.
 * 16 waves x 128 VGPRs, 120 KB LDS (the layout of k_gru_proj).  Built for NU = 6, two tiles per workgroup; other launches run k_gru_proj. */
#ifndef SH_GRU_MX_H
#define SH_GRU_MX_H

template <bool RESID, bool STAMP = false>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_vgpr(64))) void k_gru_mx(const float *__restrict__ in, float *__restrict__ out,
                                                       const float *__restrict__ resid,
                                                       const unsigned *__restrict__ iWp, const float *__restrict__ ibfrag,
                                                       const unsigned *__restrict__ sWp, const unsigned *__restrict__ sW2p,
                                                       ShMeta md, int backward, ShGruLanes L, unsigned long long *dbg = nullptr) {
    constexpr int NU = 6, NT = 2, KS = 3;
    /* cycle stamps (STAMP): per wave, cycles of work / of waiting at the barrier behind it, for each of the four phases */
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st0 = 0;
#define MXS(i) do { if (STAMP) { const unsigned long long t_ = __builtin_readcyclecounter(); st[i] += t_ - st0; st0 = t_; } } while (0)
#define MXBAR(i) do { MXS(2 * (i)); lds_barrier(); MXS(2 * (i) + 1); } while (0)
#define MXDUMP() do { if (STAMP && dbg && lane == 0) { unsigned long long *d_ = dbg + ((long long)blockIdx.x * 16 + wave) * 10; for (int i_ = 0; i_ < 8; i_++) d_[i_] = st[i_]; d_[8] = (unsigned long long)nit; } } while (0)
    constexpr int PBUF = KS * 2 * 64 * 4;          /* one operand as fp16 pieces, in 32-bit words: [ks][piece][lane][4] */
    constexpr int XBUF = 3 * NU * 256;             /* one block's gate inputs / pre-activations, accumulator layout [gate][u][lane][4] */
    constexpr int TBUF = 4 * PBUF + 2 * XBUF;      /* words per tile slot: h | r*h | in[2] | x[2] */
    extern __shared__ __attribute__((aligned(16))) unsigned ldsw[];
    const int lane = threadIdx.x & 63;
    const unsigned lofs = (unsigned)lane * 4u;
    typedef __attribute__((address_space(1))) float *gf32;
    typedef __attribute__((address_space(1))) f32x4 *gf32x4;
    auto gload = [&](const float *base) { gf32 b = (gf32)base; asm volatile("" : "+s"(b)); return *(gf32x4)(b + lofs); };
    auto gstore = [&](float *base, f32x4 v) { gf32 b = (gf32)base; asm volatile("" : "+s"(b)); *(gf32x4)(b + lofs) = v; };
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto pieces = [&](const unsigned *buf, int ks) { return load_pieces(buf + ks * 512, lane); };
    auto lds_h = [&](int tl) { return ldsw + tl * TBUF; };
    auto lds_rh = [&](int tl) { return ldsw + tl * TBUF + PBUF; };
    auto lds_in = [&](int tl, int par) { return ldsw + tl * TBUF + (2 + par) * PBUF; };
    auto lds_x = [&](int tl, int par) { return (float *)(ldsw + tl * TBUF + 4 * PBUF + par * XBUF); };

    /* every wave knows how many steps the workgroup takes */
    int sgi0[NT], sge0[NT], my_it[NT], nit = 0;
#pragma unroll
    for (int tl = 0; tl < NT; tl++) {
        const int ln = blockIdx.x * NT + tl;
        sgi0[tl] = __builtin_amdgcn_readfirstlane(L.lane_off[ln]);
        sge0[tl] = __builtin_amdgcn_readfirstlane(L.lane_off[ln + 1]);
        int n = 0;
        for (int i = sgi0[tl]; i < sge0[tl]; i++) n += L.seg[i].s1 - L.seg[i].s0;
        my_it[tl] = __builtin_amdgcn_readfirstlane(n);
        nit = max(nit, my_it[tl]);
    }
    if (nit == 0) return;                                     /* (uniform over the workgroup) */

    if (wave < 2 * NU) {
        /* ---------------- the two MFMA teams ---------------- */
        const bool rec = wave < NU;
        const int u = rec ? wave : wave - NU;
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
#pragma unroll
            for (int ks = 0; ks < KS; ks++)
                asm volatile("" : "+v"(w0[ks].p1), "+v"(w0[ks].p2), "+v"(w1[ks].p1), "+v"(w1[ks].p2), "+v"(w2[ks].p1), "+v"(w2[ks].p2));
        }
        const int oz = (u * 64 + lane) * 4, orr = ((NU + u) * 64 + lane) * 4, oh = ((2 * NU + u) * 64 + lane) * 4;
        if (rec) {
            lds_barrier();                                  /* (prologue of the other teams) */
            lds_barrier();
            if (STAMP) st0 = __builtin_readcyclecounter();
            for (int it = 0; it < nit; it++) {
                const int par = it & 1;
                /* P1: reset gate first (the elementwise team waits for it), then the update gate; both back into the ring slot they came from */
#pragma unroll
                for (int tl = 0; tl < NT; tl++) {       /* (a tile at a time: the h pieces of both tiles at once do not fit 128 registers beside the weights) */
                    float *xs = lds_x(tl, par);
                    ShSplit hp[KS];
#pragma unroll
                    for (int ks = 0; ks < KS; ks++) hp[ks] = pieces(lds_h(tl), ks);
                    *(f32x4 *)(xs + orr) = split_dot<KS>(w1, hp, *(const f32x4 *)(xs + orr));
                    *(f32x4 *)(xs + oz) = split_dot<KS>(w0, hp, *(const f32x4 *)(xs + oz));
                }
                MXBAR(0);
                MXBAR(1);                                   /* P2: the elementwise team's */
                /* P3: candidate on the r * h pieces */
#pragma unroll
                for (int tl = 0; tl < NT; tl++) {
                    float *xs = lds_x(tl, par);
                    ShSplit rp[KS];
#pragma unroll
                    for (int ks = 0; ks < KS; ks++) rp[ks] = pieces(lds_rh(tl), ks);
                    *(f32x4 *)(xs + oh) = split_dot<KS>(w2, rp, *(const f32x4 *)(xs + oh));
                }
                MXBAR(2);
                MXBAR(3);                                   /* P4 */
            }
            MXDUMP();
            return;
        }
        /* projection: gate inputs of block it + 1 into the ring's other slot */
        f32x4 bz = *(const f32x4 *)(ibfrag + (u * 64 + lane) * 4);
        f32x4 br = *(const f32x4 *)(ibfrag + ((NU + u) * 64 + lane) * 4);
        f32x4 bh = *(const f32x4 *)(ibfrag + ((2 * NU + u) * 64 + lane) * 4);
        asm volatile("" : "+v"(bz), "+v"(br), "+v"(bh));
        auto project_h = [&](int tl, int slot) {
            ShSplit ip[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ks++) ip[ks] = pieces(lds_in(tl, slot), ks);
            *(f32x4 *)(lds_x(tl, slot) + oh) = split_dot<KS>(w2, ip, bh);
        };
        auto project_zr = [&](int tl, int slot) {
            f32x4 cz = bz, cr = br;
            ShSplit ip[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ks++) ip[ks] = pieces(lds_in(tl, slot), ks);
            split_dot2<KS>(w0, w1, ip, cz, cr);
            *(f32x4 *)(lds_x(tl, slot) + oz) = cz;
            *(f32x4 *)(lds_x(tl, slot) + orr) = cr;
        };
        /* SIMDs 0 / 1 host two recurrence waves: the projection waves there (waves 8, 9) keep out of P1 / P3 */
        const bool spread = !(wave == NU + 2 || wave == NU + 3);
        lds_barrier();                                      /* block 0's input pieces are in */
#pragma unroll
        for (int tl = 0; tl < NT; tl++) { project_h(tl, 0); project_zr(tl, 0); }
        lds_barrier();
        if (STAMP) st0 = __builtin_readcyclecounter();
        for (int it = 0; it < nit; it++) {
            const int np = (it + 1) & 1;
            if (spread) project_h(0, np);
            MXBAR(0);
            if (!spread) project_h(0, np);
            project_zr(0, np);
            MXBAR(1);
            if (spread) project_h(1, np);
            MXBAR(2);
            if (!spread) project_h(1, np);
            project_zr(1, np);
            MXBAR(3);
        }
        MXDUMP();
        return;
    }

    /* ---------------- the elementwise team: wave v owns tile v >> 1, unit tiles 3 (v & 1) .. + 2 ---------------- */
    const int v = wave - 2 * NU, tl = v >> 1, u0 = 3 * (v & 1);
    auto wofs = [&](int u) { return (((u >> 1) * 2) * 64 + lane) * 4 + (u & 1) * 2; };
    auto publish = [&](unsigned *buf, int u, f32x4 x) {
        unsigned a1, a2, b1, b2;
        split_pair(x[0], x[1], a1, a2); split_pair(x[2], x[3], b1, b2);
        *(uint2 *)(buf + wofs(u)) = make_uint2(a1, b1);
        *(uint2 *)(buf + wofs(u) + 256) = make_uint2(a2, b2);
    };
    /* two cursors over the tile's lane: one runs ahead with the input fetch, one with the state and the output */
    ShLaneCursor ci = {}, co = {};
    ci.sgi = co.sgi = sgi0[tl]; ci.sge = co.sge = sge0[tl];
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
    struct X3 { f32x4 a[3]; };
    auto fetch = [&]() {                                    /* the input column at the fetch cursor (three unit tiles), then step it */
        X3 x;
        const long long col = ci.ok ? column(ci) : 0;       /* (unconditional loads: past the lane's end the layer's first column again) */
#pragma unroll
        for (int k = 0; k < 3; k++) x.a[k] = gload(in + (col * NU + u0 + k) * 256);
        if (ci.ok) {
            ci.s++;
            if (ci.s == ci.s1) { ci.sgi++; enter(ci); }
        }
        return x;
    };
    f32x4 h[3], z[3];
    int myT = 0;
    auto take_over = [&]() {                                /* initial state of the (new) current segment */
#pragma unroll
        for (int k = 0; k < 3; k++) h[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        myT = co.ok ? md.rT[co.tile * 16 + (lane & 15)] : 0;
        if (!co.ok) return;
        if (co.s > 0) {                                     /* continuation of a tile begun on another lane */
            if (!sh_wait_flag(L.flag + co.tile, (unsigned)NU, L.flag + L.ntile) && lane == 0)
                __hip_atomic_store(L.flag + L.ntile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float *hs = L.hstate + ((long long)co.tile * NU + u0 + k) * 256 + lane * 4;
#pragma unroll
                for (int j = 0; j < 4; j++) h[k][j] = __hip_atomic_load(hs + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        asm volatile("" : "+v"(myT), "+v"(h[0]), "+v"(h[1]), "+v"(h[2]));
    };
    /* prologue: block 0 as pieces, the state as pieces; then block 1 as pieces while the projection makes block 0's gate inputs */
    enter(ci); enter(co);
    take_over();
    {
        const X3 x0 = fetch();
#pragma unroll
        for (int k = 0; k < 3; k++) { publish(lds_in(tl, 0), u0 + k, x0.a[k]); publish(lds_h(tl), u0 + k, h[k]); }
    }
    X3 xq1 = fetch(), xq2 = fetch();                        /* blocks 1, 2 */
    lds_barrier();
#pragma unroll
    for (int k = 0; k < 3; k++) publish(lds_in(tl, 1), u0 + k, xq1.a[k]);
    xq1 = xq2; xq2 = fetch();                               /* blocks 2, 3 */
    lds_barrier();
    f32x4 rs[3];
    auto resid_fetch = [&]() {
        const long long col = co.ok ? column(co) : 0;
#pragma unroll
        for (int k = 0; k < 3; k++) rs[k] = gload(resid + (col * NU + u0 + k) * 256);
    };
    if (RESID) resid_fetch();
    if (STAMP) st0 = __builtin_readcyclecounter();
    for (int it = 0; it < nit; it++) {
        const int par = it & 1;
        const float *xs = lds_x(tl, par);
        /* P1: block it + 2 as pieces (into the slot block it's pieces have left), block it + 4 on its way */
#pragma unroll
        for (int k = 0; k < 3; k++) publish(lds_in(tl, par), u0 + k, xq1.a[k]);
        xq1 = xq2; xq2 = fetch();
        MXBAR(0);
        /* P2: logistic(r) * h -> pieces (layers.c:515) */
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const f32x4 cr = *(const f32x4 *)(xs + ((NU + u0 + k) * 64 + lane) * 4);
            publish(lds_rh(tl), u0 + k, d_logistic4_acc(cr) * h[k]);
        }
        MXBAR(1);
        /* P3: the update gate (its pre-activation has been in the ring since P1) */
#pragma unroll
        for (int k = 0; k < 3; k++) z[k] = d_logistic4_acc(*(const f32x4 *)(xs + ((u0 + k) * 64 + lane) * 4));
        MXBAR(2);
        /* P4: tanh, blend (layers.c:525), output, the state as pieces */
        const bool live = it < my_it[tl];                                           /* (wave-uniform) */
        const int t = backward ? co.Tt - 1 - co.s : co.s;
        const bool active = t < myT;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const f32x4 hbar = d_tanh4_acc(*(const f32x4 *)(xs + ((2 * NU + u0 + k) * 64 + lane) * 4));
            const f32x4 hn = z[k] * h[k] + (1.0f - z[k]) * hbar;
#pragma unroll
            for (int j = 0; j < 4; j++) h[k][j] = active ? hn[j] : 0.0f;
        }
        if (live) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                f32x4 o = h[k];
                if (RESID) o += rs[k];                                               /* networks.c:583 */
                gstore(out + ((long long)(co.boff + t) * NU + u0 + k) * 256, o);
            }
            co.s++;
            if (co.s == co.s1) {                                                    /* segment done */
                if (co.s1 < co.Tt) {                                                /* the tile continues on another lane */
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        float *hs = L.hstate + ((long long)co.tile * NU + u0 + k) * 256 + lane * 4;
#pragma unroll
                        for (int j = 0; j < 4; j++) __hip_atomic_store(hs + j, h[k][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    if (lane == 0) __hip_atomic_fetch_add(L.flag + co.tile, 3u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
                co.sgi++;
                enter(co);
                take_over();
            }
        }
        if (RESID) resid_fetch();
#pragma unroll
        for (int k = 0; k < 3; k++) publish(lds_h(tl), u0 + k, h[k]);
        MXBAR(3);
    }
    MXDUMP();
#undef MXS
#undef MXBAR
#undef MXDUMP
}

#endif /* SH_GRU_MX_H */
