/* sh_gru_free.h -- part of sh_kernels.h (included after sh_gru.h): k_gru_proj with synchronisation finer than the
 * workgroup barrier.
 *
 * k_gru_proj keeps both teams on two s_barriers per step, so on every SIMD the matrix-heavy and the transcendental-
 * heavy phases of all waves line up.  Here the step loop has NO s_barrier at all:
 *   - the projection team runs free, up to RING = 3 blocks ahead of the recurrence, paced by two LDS counters
 *     (FULL: gate inputs of a block written by all its waves; SYNCB: the recurrence is done with a ring slot) and
 *     synchronises its own waves (the input column travels through LDS as pieces, one chunk cut by each wave)
 *     with a third (PCNT);
 *   - the six recurrence waves synchronise among themselves with counters (SYNCA after r*h is published, SYNCB
 *     after h is) instead of s_barrier -- gfx950 has no partial / named barriers, and s_barrier would drag the
 *     projection waves back into lock step.
 * Counters are monotonic (never reset): "wave w has done step k" is count >= NU * (k + 1).  A wave's LDS
 * operations execute in order, so data written before the counter increment is visible to whoever sees the
 * increment.  The arithmetic -- every split product, every accumulator's order of MFMAs -- is k_gru_proj's: results
 * are bit-identical (tests/test_gpu_parity.py).  Same lane schedule, segments and state hand-over. */
#ifndef SH_GRU_FREE_H
#define SH_GRU_FREE_H

template <int NU, int NT, bool RESID, bool STAMP = false>
__global__ __launch_bounds__(128 * NU) void k_gru_free(const float *__restrict__ in, float *__restrict__ out,
                                                       const float *__restrict__ resid,
                                                       const unsigned *__restrict__ iWp, const float *__restrict__ ibfrag,
                                                       const unsigned *__restrict__ sWp, const unsigned *__restrict__ sW2p,
                                                       ShMeta md, int backward, ShGruLanes L,
                                                       unsigned long long *dbg = nullptr) {
    static_assert(NU % 2 == 0, "k steps of 32 units");
    constexpr int KS = NU / 2, RING = 3;
    constexpr int PBUF = KS * 2 * 64 * 4;          /* one operand as fp16 pieces, in 32-bit words: [ks][piece][lane][4] */
    constexpr int XBUF = 3 * NU * 256;             /* one block's gate inputs, accumulator layout [gate][u][lane][4] */
    constexpr int TBUF = 4 * PBUF + RING * XBUF;   /* words per tile slot: h | r*h | in[2] | x[RING] */
    enum { C_FULL = 0, C_SYNCA = 1, C_SYNCB = 2, C_PCNT = 3 };
    extern __shared__ __attribute__((aligned(16))) unsigned ldsw[];
    volatile unsigned *cnt = (volatile unsigned *)(ldsw + NT * TBUF);
    unsigned long long s_work = 0, s_w0 = 0, s_w1 = 0, s_w2 = 0, s_t0 = 0, s_t1;
#define FSTAMP(acc) do { if (STAMP) { s_t1 = __builtin_readcyclecounter(); acc += s_t1 - s_t0; s_t0 = s_t1; } } while (0)
    const int lane = threadIdx.x & 63;
    const unsigned lofs = (unsigned)lane * 4u;
    typedef __attribute__((address_space(1))) float *gf32;
    typedef __attribute__((address_space(1))) f32x4 *gf32x4;
    auto gload = [&](const float *base) { gf32 b = (gf32)base; asm volatile("" : "+s"(b)); return *(gf32x4)(b + lofs); };
    auto gstore = [&](float *base, f32x4 v) { gf32 b = (gf32)base; asm volatile("" : "+s"(b)); *(gf32x4)(b + lofs) = v; };
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool rec = wave < NU;
    const int u = rec ? wave : wave - NU;
    /* counter increment after this wave's LDS writes (in order behind them), and the wait for a count */
    auto arrive = [&](int k) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add((unsigned *)&cnt[k], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto wait_for = [&](int k, unsigned need) {
        for (;;) {
            const unsigned v = __builtin_amdgcn_readfirstlane(cnt[k]);
            if ((int)(v - need) >= 0) break;
            __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
    };

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
    const int wofs = (((u >> 1) * 2) * 64 + lane) * 4 + (u & 1) * 2;
    auto publish = [&](unsigned *buf, f32x4 v) {
        unsigned a1, a2, b1, b2;
        split_pair(v[0], v[1], a1, a2); split_pair(v[2], v[3], b1, b2);
        *(uint2 *)(buf + wofs) = make_uint2(a1, b1);
        *(uint2 *)(buf + wofs + 256) = make_uint2(a2, b2);
    };
    auto pieces = [&](const unsigned *buf, int ks) { return load_pieces(buf + ks * 512, lane); };
    auto lds_h = [&](int tl) { return ldsw + tl * TBUF; };
    auto lds_rh = [&](int tl) { return ldsw + tl * TBUF + PBUF; };
    auto lds_in = [&](int tl, int par) { return ldsw + tl * TBUF + (2 + par) * PBUF; };
    auto lds_x = [&](int tl, int slot) { return (float *)(ldsw + tl * TBUF + 4 * PBUF + slot * XBUF); };

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
    if (threadIdx.x < 16) cnt[threadIdx.x] = 0u;
    __syncthreads();                                          /* the only workgroup barrier: counters are zero */
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

#ifndef SH_FREE_REC_PRIO
#define SH_FREE_REC_PRIO 0      /* s_setprio of the recurrence waves (the projection waves stay at 0) */
#endif
    if (SH_FREE_REC_PRIO && rec) __builtin_amdgcn_s_setprio(SH_FREE_REC_PRIO);
    if (!rec) {
        /* ---------------- projection team: free running, at most RING blocks ahead ---------------- */
        f32x4 bz = *(const f32x4 *)(ibfrag + (u * 64 + lane) * 4);
        f32x4 br = *(const f32x4 *)(ibfrag + ((NU + u) * 64 + lane) * 4);
        f32x4 bh = *(const f32x4 *)(ibfrag + ((2 * NU + u) * 64 + lane) * 4);
        asm volatile("" : "+v"(bz), "+v"(br), "+v"(bh));
        auto fetch = [&](ShLaneCursor &cc) {       /* (unconditional load: see k_gru_proj) */
            const long long col = cc.ok ? column(cc) : 0;
            const f32x4 v = gload(in + (col * NU + u) * 256);
            if (cc.ok) {
                cc.s++;
                if (cc.s == cc.s1) { cc.sgi++; enter(cc); }
            }
            return v;
        };
        f32x4 q0[NT], q1[NT], q2[NT];              /* this wave's chunk of blocks j, j + 1, j + 2 */
#pragma unroll
        for (int tl = 0; tl < NT; tl++) {
            enter(c[tl]);
            q0[tl] = fetch(c[tl]); q1[tl] = fetch(c[tl]); q2[tl] = fetch(c[tl]);
        }
        if (STAMP) s_t0 = __builtin_readcyclecounter();
        int slot = 0;
        for (int j = 0; j < nit; j++) {
#pragma unroll
            for (int tl = 0; tl < NT; tl++) {
                publish(lds_in(tl, j & 1), q0[tl]);
                q0[tl] = q1[tl]; q1[tl] = q2[tl];
                q2[tl] = fetch(c[tl]);
            }
            arrive(C_PCNT);
            FSTAMP(s_work);
            wait_for(C_PCNT, (unsigned)NU * (unsigned)(j + 1));                 /* the whole column of block j is in LDS */
            FSTAMP(s_w0);
            if (j >= RING) wait_for(C_SYNCB, (unsigned)NU * (unsigned)(j - RING + 2));       /* the recurrence is done with this ring slot */
            FSTAMP(s_w1);
#pragma unroll
            for (int tl = 0; tl < NT; tl++) {
                ShSplit ip[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ks++) ip[ks] = pieces(lds_in(tl, j & 1), ks);
                /* (the affine kernels' order of products on every accumulator: bit-identical to them) */
                const f32x4 ah = split_dot<KS>(w2, ip, bh);
                f32x4 cz = bz, cr = br;
                split_dot2<KS>(w0, w1, ip, cz, cr);
                float *xdst = lds_x(tl, slot);
                *(f32x4 *)(xdst + (u * 64 + lane) * 4) = cz;
                *(f32x4 *)(xdst + ((NU + u) * 64 + lane) * 4) = cr;
                *(f32x4 *)(xdst + ((2 * NU + u) * 64 + lane) * 4) = ah;
            }
            arrive(C_FULL);
            slot = (slot == RING - 1) ? 0 : slot + 1;
            FSTAMP(s_work);
        }
        if (STAMP && dbg && lane == 0) { unsigned long long *d_ = dbg + ((long long)blockIdx.x * 2 * NU + wave) * 16; d_[0] = s_work; d_[1] = s_w0; d_[2] = s_w1; d_[3] = 0; d_[4] = nit; }
        return;
    }

    /* ---------------- recurrence team ---------------- */
    static_assert(NT <= 2, "two block counts per register");
    unsigned myT2 = 0;
    f32x4 h[NT];
    auto take_over = [&](int tl) {                  /* initial state of lane tl's (new) current segment */
        h[tl] = (f32x4){0.f, 0.f, 0.f, 0.f};
        int mt = 0;
        if (c[tl].ok) mt = md.rT[c[tl].tile * 16 + (lane & 15)];
        if (NT == 1) myT2 = (unsigned)mt;
        else myT2 = tl ? ((myT2 & 0xffffu) | ((unsigned)mt << 16)) : ((myT2 & 0xffff0000u) | ((unsigned)mt & 0xffffu));
        if (!c[tl].ok) return;
        if (c[tl].s > 0) {                          /* continuation of a tile begun on another lane */
            if (!sh_wait_flag(L.flag + c[tl].tile, (unsigned)NU, L.flag + L.ntile) && lane == 0)
                __hip_atomic_store(L.flag + L.ntile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const float *hs = L.hstate + ((long long)c[tl].tile * NU + u) * 256 + lane * 4;
#pragma unroll
            for (int k = 0; k < 4; k++) h[tl][k] = __hip_atomic_load(hs + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("" : "+v"(myT2), "+v"(h[tl][0]), "+v"(h[tl][1]), "+v"(h[tl][2]), "+v"(h[tl][3]));
    };
#pragma unroll
    for (int tl = 0; tl < NT; tl++) {
        enter(c[tl]);
        take_over(tl);
        publish(lds_h(tl), h[tl]);
    }
    arrive(C_SYNCB);                                /* round 0 of SYNCB: the initial state is published */
    f32x4 rs[NT];
    auto resid_fetch = [&](int tl) {
        const long long col = c[tl].ok ? column(c[tl]) : 0;
        rs[tl] = gload(resid + (col * NU + u) * 256);
    };
    if (RESID) {
#pragma unroll
        for (int tl = 0; tl < NT; tl++) resid_fetch(tl);
    }
    if (STAMP) s_t0 = __builtin_readcyclecounter();
    int slot = 0;
    for (int it = 0; it < nit; it++) {
        const unsigned need = (unsigned)NU * (unsigned)(it + 1);
        wait_for(C_FULL, need);                     /* this block's gate inputs are in the ring */
        FSTAMP(s_w0);
        wait_for(C_SYNCB, need);                    /* every recurrence wave has published h of the previous step */
        FSTAMP(s_w2);
        /* phase A: reset and update gates on the h pieces; r*h -> LDS (reset gate first: see k_gru_proj) */
        f32x4 cr[NT], cz[NT], z[NT];
        {
            ShSplit hp[NT][KS];
#pragma unroll
            for (int tl = 0; tl < NT; tl++) {
                const float *xs = lds_x(tl, slot);
                cz[tl] = *(const f32x4 *)(xs + (u * 64 + lane) * 4);
                cr[tl] = *(const f32x4 *)(xs + ((NU + u) * 64 + lane) * 4);
#pragma unroll
                for (int ks = 0; ks < KS; ks++) hp[tl][ks] = pieces(lds_h(tl), ks);
                cr[tl] = split_dot<KS>(w1, hp[tl], cr[tl]);
            }
#pragma unroll
            for (int tl = 0; tl < NT; tl++) cz[tl] = split_dot<KS>(w0, hp[tl], cz[tl]);
#pragma unroll
            for (int tl = 0; tl < NT; tl++) publish(lds_rh(tl), d_logistic4_acc(cr[tl]) * h[tl]);      /* layers.c:515 */
        }
        arrive(C_SYNCA);
        FSTAMP(s_work);
        wait_for(C_SYNCA, need);                    /* r*h of every unit tile is in LDS */
        FSTAMP(s_w1);
        /* phase B: candidate on the r*h pieces, blend, publish */
        f32x4 ch[NT];
#pragma unroll
        for (int tl = 0; tl < NT; tl++) {
            ShSplit rp[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ks++) rp[ks] = pieces(lds_rh(tl), ks);
            ch[tl] = split_dot<KS>(w2, rp, *(const f32x4 *)(lds_x(tl, slot) + ((2 * NU + u) * 64 + lane) * 4));
        }
#pragma unroll
        for (int tl = 0; tl < NT; tl++) z[tl] = d_logistic4_acc(cz[tl]);
#pragma unroll
        for (int tl = 0; tl < NT; tl++) {
            const bool live = it < my_it[tl];                                          /* (wave-uniform) */
            const int t = backward ? c[tl].Tt - 1 - c[tl].s : c[tl].s;
            const bool active = t < (int)(NT == 1 ? myT2 : (tl ? (myT2 >> 16) : (myT2 & 0xffffu)));
            {
                const f32x4 hbar = d_tanh4_acc(ch[tl]);
                const f32x4 hn = z[tl] * h[tl] + (1.0f - z[tl]) * hbar;                /* layers.c:525 */
#pragma unroll
                for (int k = 0; k < 4; k++) h[tl][k] = active ? hn[k] : 0.0f;
            }
            if (live) {
                f32x4 o = h[tl];
                const long long oidx = ((long long)(c[tl].boff + t) * NU + u) * 256;       /* uniform */
                if (RESID) o += rs[tl];                                               /* networks.c:583 */
                gstore(out + oidx, o);
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
            if (RESID) resid_fetch(tl);                                                /* the next step's column */
            publish(lds_h(tl), h[tl]);
        }
        arrive(C_SYNCB);
        slot = (slot == RING - 1) ? 0 : slot + 1;
        FSTAMP(s_work);
    }
    if (STAMP && dbg && lane == 0) { unsigned long long *d_ = dbg + ((long long)blockIdx.x * 2 * NU + wave) * 16; d_[0] = s_work; d_[1] = s_w0; d_[2] = s_w1; d_[3] = s_w2; d_[4] = nit; }
#undef FSTAMP
}

#endif /* SH_GRU_FREE_H */
