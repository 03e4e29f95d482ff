/* sh_gru32x2.h -- part of sh_kernels.h (experiments build, after sh_gru32.h): k_gru_proj32 with TWO tiles of 32 reads per workgroup, in
 * opposite phases.  layers.c:373-527, scrappie_matrix.c:323, layers.c:303.
 *
 * WHY.  profiles/r4_gru32_profile.txt: k_gru_proj32 issues 38 % fewer VALU instructions and has 36 % less LDS activity than
 * k_gru_proj and takes the same 3.1 ms -- a recurrent layer is bound by the serial chain of a step (reset-gate products -> logistic
 * -> r*h -> [barrier] -> candidate products -> tanh, blend -> h -> [barrier]), which on one tile leaves a chain wave with nothing to
 * issue while its own 18 dependent MFMAs run, twice per step.  Here a workgroup steps two tiles half a step apart: every barrier
 * interval carries phase A (reset gate) of one tile and phase B (candidate, blend) of the other, so a chain wave has the other
 * tile's VALU work to issue under a chain, and the G / C / L waves' work per interval is one tile's, as before.  Arithmetic, per
 * tile, is k_gru_proj32's instruction for instruction: identical bits (tests/test_gru32.py).
 *
 * INTERVALS.  k = -1, 0, 1, ...:   O(k): slot 1 in phase A of its step k,     slot 0 in phase B of its step k
 *                                  E(k): slot 0 in phase A of its step k + 1, slot 1 in phase B of its step k
 * (a phase of step -1 or past the slot's last step does nothing but keep the counters and barriers going).  With P the slot in
 * phase A and Q the slot in phase B of an interval:
 *   R_j  P: x_r(P) + sW_r . h(P) -> logistic -> r*h(P) -> pieces;   Q: x_c(Q) + sW2 . r*h(Q), z(Q), blend -> h(Q) -> HBM, pieces
 *   G_j  P: update gate of P's step = its projection (kept since P's last phase B) + sW_z . h(P) -> LDS
 *        Q: projection rows z, r of Q's NEXT block from its input column -> x_r(Q) to LDS, x_z(Q) kept
 *   C    Q: candidate projection of Q's next block; it overwrites x_c(Q), which the chain waves read at the start of this very
 *           interval: they count themselves off in LDS once their reads are in, C checks the count before it writes
 *   L    P: the input column of P's next block (fetched two intervals ago) cut into pieces -> IN(P); fetches the one after
 * Every buffer is written in one interval and read in the next one of the same slot, so all of them are single: 72 KB of LDS per
 * slot, 144 KB + the bias table per workgroup.
 */
#ifndef SH_GRU32X2_H
#define SH_GRU32X2_H

#define SH_G32X2_SLOT_WORDS (3 * 3072 + 9 * 1024)
#define SH_G32X2_LDS_WORDS (2 * SH_G32X2_SLOT_WORDS + 288 + 16 + 3 * 2 * 512)
#ifndef SH_G32X2_RPRIO
#define SH_G32X2_RPRIO 0
#endif

template <bool RESID, bool STAMP>
__global__ __launch_bounds__(512) void k_gru_proj32x2(const float *__restrict__ in, float *__restrict__ out,
                                                      const unsigned *__restrict__ iWp, const float *__restrict__ ibias,
                                                      const unsigned *__restrict__ sWp, const unsigned *__restrict__ sW2p,
                                                      ShMeta md, int backward, ShGruPairs L /* two lanes per workgroup */, unsigned long long *dbg) {
    extern __shared__ __attribute__((aligned(16))) unsigned ldsw[];
    auto Hs = [&](int s) { return ldsw + s * SH_G32X2_SLOT_WORDS; };
    auto RHs = [&](int s) { return ldsw + s * SH_G32X2_SLOT_WORDS + 3072; };
    auto INs = [&](int s) { return ldsw + s * SH_G32X2_SLOT_WORDS + 6144; };
    auto ring = [&](int s, int gate, int j) { return (float *)(ldsw + s * SH_G32X2_SLOT_WORDS + 9216) + (gate * 3 + j) * 1024; };
    float *const BIAS = (float *)(ldsw + 2 * SH_G32X2_SLOT_WORDS);
    unsigned *const CNT = ldsw + 2 * SH_G32X2_SLOT_WORDS + 288;      /* [slot]: chain waves that have read x_c of the slot, ever */
    unsigned *const WLDS = CNT + 16;                                 /* [chain wave][r | c][2 pieces][64 lanes][4]: the chain waves' weights of their last k step
                                                                        (their registers hold five k steps of each gate; the sixth does not fit beside two tiles' state) */
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    unsigned long long sa = 0, sb = 0, st0 = 0, st1;
#define XSTAMP(acc) do { if (STAMP) { st1 = __builtin_readcyclecounter(); acc += st1 - st0; st0 = st1; } } while (0)

    /* the two lanes of the schedule this workgroup steps */
    int sg0[2], sg1[2], nsteps[2], nit = 0;
#pragma unroll
    for (int s = 0; s < 2; s++) {
        sg0[s] = __builtin_amdgcn_readfirstlane(L.lane_off[blockIdx.x * 2 + s]);
        sg1[s] = __builtin_amdgcn_readfirstlane(L.lane_off[blockIdx.x * 2 + s + 1]);
        int n = 0;
        for (int i = sg0[s]; i < sg1[s]; i++) n += L.seg[i].s1 - L.seg[i].s0;
        nsteps[s] = __builtin_amdgcn_readfirstlane(n);
        nit = max(nit, nsteps[s]);
    }
    if (nit == 0) return;
    if (threadIdx.x < 288) BIAS[threadIdx.x] = ibias[threadIdx.x];
    if (threadIdx.x < 2) CNT[threadIdx.x] = 0u;

    const int half = (lane >> 4) & 1;
    const unsigned lanepart = (unsigned)(((lane >> 5) * 16 + (lane & 15)) * 16);
    /* per slot: the lane's walk over its segments (wave-uniform) and the per-lane view of the current pair (sh_gru32.h) */
    ShPairCursor c[2];
    int hT[2] = {0, 0}, myT[2] = {0, 0};
    unsigned voff[2] = {0u, 0u};
    auto enter = [&](const int s) {
        ShPairCursor &cc = c[s];
        cc.ok = cc.sgi < cc.sge;
        if (cc.ok) {
            const ShGruSegD sg = L.seg[cc.sgi];
            cc.pair = __builtin_amdgcn_readfirstlane(sg.tile);
            cc.s = __builtin_amdgcn_readfirstlane(sg.s0);
            cc.s1 = __builtin_amdgcn_readfirstlane(sg.s1);
            const int tA = __builtin_amdgcn_readfirstlane(L.pair_tile[2 * cc.pair]);
            const int tB = __builtin_amdgcn_readfirstlane(L.pair_tile[2 * cc.pair + 1]);
            const int T0 = __builtin_amdgcn_readfirstlane(md.tile_T[tA]);
            const int T1 = tB >= 0 ? __builtin_amdgcn_readfirstlane(md.tile_T[tB]) : 0;
            cc.Tt = max(T0, T1);
            {
                const unsigned long long b0 = (unsigned long long)md.tile_boff[tA];
                cc.boff0 = (long long)(((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(b0 >> 32)) << 32) | __builtin_amdgcn_readfirstlane((unsigned)b0));
            }
            long long boff1 = cc.boff0;
            if (tB >= 0) {
                const unsigned long long b1 = (unsigned long long)md.tile_boff[tB];
                boff1 = (long long)(((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(b1 >> 32)) << 32) | __builtin_amdgcn_readfirstlane((unsigned)b1));
            }
            const bool second = half && T1 > 0;
            hT[s] = half ? T1 : T0;
            myT[s] = (half ? tB >= 0 : true) ? md.rT[(half ? tB : tA) * 16 + (lane & 15)] : 0;
            voff[s] = lanepart + (second ? (unsigned)(boff1 - cc.boff0) * (unsigned)SH_G32_COLB : 0u);
        }
    };
    auto block_off = [&](const int s, int t) {
        const int lim = hT[s] > 0 ? hT[s] - 1 : 0;
        return (unsigned)min(t, lim) * (unsigned)SH_G32_COLB + voff[s];
    };
#pragma unroll
    for (int s = 0; s < 2; s++) { c[s].sgi = sg0[s]; c[s].sge = sg1[s]; c[s].ok = false; c[s].pair = 0; c[s].s = 0; c[s].s1 = 0; c[s].Tt = 0; c[s].boff0 = 0; }

    if (wave < 3) {
        /* ------------------------------ R_j: the chains of both slots ------------------------------ */
        const int j = wave;
        if (SH_G32X2_RPRIO) __builtin_amdgcn_s_setprio(SH_G32X2_RPRIO);
        ShSplit wr[6], wc[6];
        int kofs[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const int ks = (2 * j + i) % 6;
            kofs[i] = ks * 512;
            wr[i] = load_pieces(sWp + ((3 + j) * 6 + ks) * 512, lane);
            wc[i] = load_pieces(sW2p + (j * 6 + ks) * 512, lane);
        }
        {   /* the last k step's weights go to LDS */
            unsigned *wl = WLDS + j * 1024 + lane * 4;
            *(u32x4 *)wl = __builtin_bit_cast(u32x4, wr[5].p1); *(u32x4 *)(wl + 256) = __builtin_bit_cast(u32x4, wr[5].p2);
            *(u32x4 *)(wl + 512) = __builtin_bit_cast(u32x4, wc[5].p1); *(u32x4 *)(wl + 768) = __builtin_bit_cast(u32x4, wc[5].p2);
        }
#pragma unroll
        for (int ks = 0; ks < 5; ks++) asm volatile("" : "+v"(wr[ks].p1), "+v"(wr[ks].p2), "+v"(wc[ks].p1), "+v"(wc[ks].p2));
        f32x4 h[2][4];
        ShSplit own[2];                                    /* the h pieces this wave published last: those of the slot that is in phase A next */
        auto publish_own = [&](unsigned *buf, const f32x4 (&v)[4]) {
            u32x4 p1[2], p2[2];
            cut16(v, p1, p2);
            put16(buf, j, lane, p1, p2);
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) { own[s2].p1 = __builtin_bit_cast(f16x8, p1[s2]); own[s2].p2 = __builtin_bit_cast(f16x8, p2[s2]); }
        };
        auto take_over = [&](const int s) {
#pragma unroll
            for (int g = 0; g < 4; g++) h[s][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (c[s].ok && c[s].s > 0) {
                if (!sh_wait_flag(L.flag + c[s].pair, 3u, L.err) && lane == 0)
                    __hip_atomic_store(L.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                const float *hs = L.hstate + ((long long)c[s].pair * 3 + j) * 1024 + lane * 4;
#pragma unroll
                for (int g = 0; g < 4; g++)
#pragma unroll
                    for (int k = 0; k < 4; k++) h[s][g][k] = __hip_atomic_load(hs + g * 256 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("" : "+v"(h[s][0]), "+v"(h[s][1]), "+v"(h[s][2]), "+v"(h[s][3]), "+v"(hT[s]), "+v"(myT[s]), "+v"(voff[s]));
        };
#pragma unroll
        for (int s = 0; s < 2; s++) {
            enter(s);
            take_over(s);
            publish_own(Hs(s), h[s]);
        }
        lds_barrier();
        if (STAMP) st0 = __builtin_readcyclecounter();
        /* one interval: P in phase A of its step kP, Q in phase B of its step kQ (compile-time slots) */
        bool own_ok = false;                               /* `own` holds the h pieces of the slot about to enter phase A */
        auto body = [&](auto Pc, auto Qc, auto LPc, auto LQc) {
            constexpr int P = decltype(Pc)::value, Q = decltype(Qc)::value;
            constexpr bool liveP = decltype(LPc)::value, liveQ = decltype(LQc)::value;
            /* P's operands first; Q's are read WHILE P's dependent products issue (a k step's pieces leave their registers as its three
             * products are issued, the next operand takes them): both chains' operands at once do not fit the register file */
            f32x16 accP = {}, accQ = {}, za = {};
            ShSplit hp[6], rp[6];
            ShSplit wr5, wc5;
            if (liveP) {
                accP = acc_read(ring(P, 1, j), lane);
                if (own_ok) { hp[0] = own[0]; hp[1] = own[1]; }
                else { hp[0] = load_pieces(Hs(P) + kofs[0], lane); hp[1] = load_pieces(Hs(P) + kofs[1], lane); }
#pragma unroll
                for (int i = 2; i < 6; i++) hp[i] = load_pieces(Hs(P) + kofs[i], lane);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 6; ks++) {
                if (liveP && ks == 2) wr5 = load_pieces(WLDS + j * 1024, lane);
                if (liveP) accP = split_k32(ks == 5 ? wr5 : wr[ks], hp[ks], accP);
                if (liveQ) {
                    if (ks == 0) accQ = acc_read(ring(Q, 2, j), lane);
                    if (ks == 1) rp[0] = load_pieces(RHs(Q) + kofs[0], lane);       /* (Q's operands arrive as P's products free registers) */
                    if (ks == 2) rp[1] = load_pieces(RHs(Q) + kofs[1], lane);
                    if (ks == 3) rp[2] = load_pieces(RHs(Q) + kofs[2], lane);
                    if (ks == 4) rp[3] = load_pieces(RHs(Q) + kofs[3], lane);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            /* x_c(Q) is in: C may overwrite it with the next block's (counted whether the slot is live or not) */
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(CNT + Q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_sched_barrier(0);
            /* Q's 18 dependent products, each followed by a slice of P's activations (a wave issues in order: behind a dependent MFMA that
             * cannot issue yet nothing else of the wave does, so the VALU work meant to run under the chain has to sit BETWEEN its MFMAs).
             * The slices are d_logistic4_acc(accP) * h and cut16 (sh_kernels.h, sh_gru32.h) taken apart: the same operations, the same bits */
            f32x4 tP[4], rh[4];
            u32x4 cp1[2], cp2[2];
#define SH_X2_SLICE(I)                                                                                              \
            if (liveP) {                                                                                            \
                if ((I) < 4) tP[(I) & 3] = grp16<(I) & 3>(accP) * (-1.44269504088896341f * SH_OINV);                 \
                else if ((I) < 8) { _Pragma("unroll") for (int k = 0; k < 4; k++) tP[(I) & 3][k] = __builtin_amdgcn_exp2f(tP[(I) & 3][k]); } \
                else if ((I) < 12) { tP[(I) & 3] = 1.0f + tP[(I) & 3]; _Pragma("unroll") for (int k = 0; k < 4; k++) tP[(I) & 3][k] = d_rcp(tP[(I) & 3][k]); } \
                else if ((I) < 16) rh[(I) & 3] = tP[(I) & 3] * h[P][(I) & 3];                                        \
                else {                                                                                              \
                    constexpr int s2 = (I) & 1;                                                                     \
                    unsigned a1, a2, b1, b2, c1, c2, d1, d2;                                                        \
                    split_pair32(rh[2 * s2][0], rh[2 * s2][1], a1, a2);                                             \
                    split_pair32(rh[2 * s2][2], rh[2 * s2][3], b1, b2);                                             \
                    split_pair32(rh[2 * s2 + 1][0], rh[2 * s2 + 1][1], c1, c2);                                     \
                    split_pair32(rh[2 * s2 + 1][2], rh[2 * s2 + 1][3], d1, d2);                                     \
                    cp1[s2] = (u32x4){a1, b1, c1, d1};                                                              \
                    cp2[s2] = (u32x4){a2, b2, c2, d2};                                                              \
                }                                                                                                   \
            }
#define SH_X2_STEP(KS, PR, I)                                                                                       \
            if (liveQ && (KS) == 2 && (PR) == 2) wc5 = load_pieces(WLDS + j * 1024 + 512, lane);                      \
            if (liveQ) { const ShSplit &wq = (KS) == 5 ? wc5 : wc[(KS) == 5 ? 0 : (KS)];                               \
                         accQ = (PR) == 0 ? mfma32(wq.p1, rp[KS].p2, accQ) : (PR) == 1 ? mfma32(wq.p2, rp[KS].p1, accQ) : mfma32(wq.p1, rp[KS].p1, accQ); } \
            if (liveQ && (KS) == 0 && (PR) == 2) rp[4] = load_pieces(RHs(Q) + kofs[4], lane);                         \
            if (liveQ && (KS) == 1 && (PR) == 2) rp[5] = load_pieces(RHs(Q) + kofs[5], lane);                         \
            SH_X2_SLICE(I)                                                                                          \
            __builtin_amdgcn_sched_barrier(0);
            SH_X2_STEP(0, 0, 0) SH_X2_STEP(0, 1, 1) SH_X2_STEP(0, 2, 2) SH_X2_STEP(1, 0, 3) SH_X2_STEP(1, 1, 4) SH_X2_STEP(1, 2, 5)
            SH_X2_STEP(2, 0, 6) SH_X2_STEP(2, 1, 7) SH_X2_STEP(2, 2, 8) SH_X2_STEP(3, 0, 9) SH_X2_STEP(3, 1, 10) SH_X2_STEP(3, 2, 11)
            SH_X2_STEP(4, 0, 12) SH_X2_STEP(4, 1, 13) SH_X2_STEP(4, 2, 14) SH_X2_STEP(5, 0, 15) SH_X2_STEP(5, 1, 16) SH_X2_STEP(5, 2, 17)
#undef SH_X2_STEP
#undef SH_X2_SLICE
            if (liveQ) za = acc_read(ring(Q, 0, j), lane);
            if (liveP) {                                   /* phase A of P done (layers.c:511-515): r*h as pieces */
                put16(RHs(P), j, lane, cp1, cp2);
            }
            if (liveQ) {                                   /* phase B of Q: layers.c:517-525 */
                const int t = backward ? c[Q].Tt - 1 - c[Q].s : c[Q].s;
                f32x4 rs[4];
                if (RESID) {
                    const float *rb = in + c[Q].boff0 * 1536 + j * 512;
                    const unsigned o = block_off(Q, t);
#pragma unroll
                    for (int g = 0; g < 4; g++) rs[g] = gload_so(rb, o, (g >> 1) * 1024 + (g & 1) * 512);
                }
                const bool active = t < myT[Q];
#define SH_X2_BLEND(G)                                                                     \
                {                                                                          \
                    const f32x4 z = g32_logistic(grp16<G>(za));                            \
                    f32x4 hn;                                                              \
                    f32x4 y = grp16<G>(accQ) * (-2.0f * 1.44269504088896341f * SH_OINV);   \
                    _Pragma("unroll") for (int k = 0; k < 4; k++) y[k] = d_rcp(1.0f + __builtin_amdgcn_exp2f(y[k])); \
                    _Pragma("unroll") for (int k = 0; k < 4; k++) {                        \
                        const float hbar = __builtin_fmaf(2.0f, y[k], -1.0f);              \
                        hn[k] = __builtin_fmaf(z[k], h[Q][G][k] - hbar, hbar);             \
                    }                                                                      \
                    _Pragma("unroll") for (int k = 0; k < 4; k++) h[Q][G][k] = active ? hn[k] : 0.0f; \
                }
                SH_X2_BLEND(0) SH_X2_BLEND(1) SH_X2_BLEND(2) SH_X2_BLEND(3)
#undef SH_X2_BLEND
                if (t < hT[Q]) {
                    float *ob = out + (c[Q].boff0 + t) * 1536 + j * 512;
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        f32x4 o = h[Q][g];
                        if (RESID) o += rs[g];
                        gstore_so(ob, voff[Q], (g >> 1) * 1024 + (g & 1) * 512, o);
                    }
                }
                c[Q].s++;
                if (c[Q].s == c[Q].s1) {                   /* segment done */
                    if (c[Q].s1 < c[Q].Tt) {               /* the pair continues on another lane */
                        float *hs = L.hstate + ((long long)c[Q].pair * 3 + j) * 1024 + lane * 4;
#pragma unroll
                        for (int g = 0; g < 4; g++)
#pragma unroll
                            for (int k = 0; k < 4; k++) __hip_atomic_store(hs + g * 256 + k, h[Q][g][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                        if (lane == 0) __hip_atomic_fetch_add(L.flag + c[Q].pair, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    c[Q].sgi++;
                    enter(Q);
                    take_over(Q);
                }
                publish_own(Hs(Q), h[Q]);
            }
            own_ok = liveQ;
            XSTAMP(sa);
            lds_barrier();
            XSTAMP(sb);
        };
        auto interval = [&](auto Pc, auto Qc, const int kP, const int kQ) {
            constexpr int P = decltype(Pc)::value, Q = decltype(Qc)::value;
            const bool lp = kP >= 0 && kP < nsteps[P], lq = kQ >= 0 && kQ < nsteps[Q];
            if (lp && lq) body(Pc, Qc, std::true_type{}, std::true_type{});
            else if (lp) body(Pc, Qc, std::true_type{}, std::false_type{});
            else if (lq) body(Pc, Qc, std::false_type{}, std::true_type{});
            else body(Pc, Qc, std::false_type{}, std::false_type{});
        };
        for (int k = -1; k < nit; k++) {
            interval(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, k, k);            /* O(k) */
            interval(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, k + 1, k);        /* E(k) */
        }
    } else if (wave >= 4 && wave < 7) {
        /* ------------------------------ G_j ------------------------------ */
        const int j = wave - 4;
        ShSplit wz[6], wrr[6], uz[6];
#pragma unroll
        for (int ks = 0; ks < 6; ks++) {
            wz[ks] = load_pieces(iWp + (j * 6 + ks) * 512, lane);
            wrr[ks] = load_pieces(iWp + ((3 + j) * 6 + ks) * 512, lane);
            uz[ks] = load_pieces(sWp + (j * 6 + ks) * 512, lane);
        }
#pragma unroll
        for (int ks = 0; ks < 6; ks++) asm volatile("" : "+v"(wz[ks].p1), "+v"(wz[ks].p2), "+v"(wrr[ks].p1), "+v"(wrr[ks].p2), "+v"(uz[ks].p1), "+v"(uz[ks].p2));
        f32x16 xz[2] = {};
        lds_barrier();
        if (STAMP) st0 = __builtin_readcyclecounter();
        auto interval = [&](auto Pc, auto Qc, const int kP, const int kQ) {
            constexpr int P = decltype(Pc)::value, Q = decltype(Qc)::value;
            const bool liveP = kP >= 0 && kP < nsteps[P];
            const bool projQ = kQ >= -1 && kQ + 1 < nsteps[Q];           /* Q's next block exists */
            ShSplit q[12];
            f32x16 xr = {};
            auto item = [&](const int i) { return load_pieces((i < 6 ? Hs(P) + i * 512 : INs(Q) + (i - 6) * 512), lane); };
#pragma unroll
            for (int i = 0; i < SH_G32_D; i++) q[i] = item(i);
#pragma unroll
            for (int i = 0; i < 12; i++) {
                if (i + SH_G32_D < 12) q[i + SH_G32_D] = item(i + SH_G32_D);
                if (i < 6) { if (liveP) xz[P] = split_k32(uz[i], q[i], xz[P]); }
                else if (projQ) {
                    xz[Q] = split_k32(wz[i - 6], q[i], xz[Q]);
                    xr = split_k32(wrr[i - 6], q[i], xr);
                }
                if (i == 5) {
                    if (liveP) acc_write(ring(P, 0, j), lane, xz[P]);
                    if (projQ) { xz[Q] = bias_read(BIAS, j, lane); xr = bias_read(BIAS, 3 + j, lane); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (projQ) acc_write(ring(Q, 1, j), lane, xr);
            XSTAMP(sa);
            lds_barrier();
            XSTAMP(sb);
        };
        for (int k = -1; k < nit; k++) {
            interval(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, k, k);
            interval(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, k + 1, k);
        }
    } else if (wave == 3) {
        /* ------------------------------ C ------------------------------ */
        ShSplit w[3][6];
#pragma unroll
        for (int m = 0; m < 3; m++)
#pragma unroll
            for (int ks = 0; ks < 6; ks++) w[m][ks] = load_pieces(iWp + ((6 + m) * 6 + ks) * 512, lane);
#pragma unroll
        for (int m = 0; m < 3; m++)
#pragma unroll
            for (int ks = 0; ks < 6; ks++) asm volatile("" : "+v"(w[m][ks].p1), "+v"(w[m][ks].p2));
        unsigned nQ[2] = {0u, 0u};                         /* phase-B intervals of each slot so far */
        lds_barrier();
        if (STAMP) st0 = __builtin_readcyclecounter();
        auto interval = [&](auto Qc, const int kQ) {
            constexpr int Q = decltype(Qc)::value;
            const bool projQ = kQ >= -1 && kQ + 1 < nsteps[Q];
            nQ[Q]++;
            if (projQ) {
                f32x16 a0 = bias_read(BIAS, 6, lane), a1 = bias_read(BIAS, 7, lane), a2 = bias_read(BIAS, 8, lane);
                ShSplit q[6];
#pragma unroll
                for (int ks = 0; ks < SH_G32_D; ks++) q[ks] = load_pieces(INs(Q) + ks * 512, lane);
#pragma unroll
                for (int ks = 0; ks < 6; ks++) {
                    if (ks + SH_G32_D < 6) q[ks + SH_G32_D] = load_pieces(INs(Q) + (ks + SH_G32_D) * 512, lane);
                    const ShSplit &ip = q[ks];
                    a0 = mfma32(w[0][ks].p1, ip.p2, a0); a1 = mfma32(w[1][ks].p1, ip.p2, a1); a2 = mfma32(w[2][ks].p1, ip.p2, a2);
                    a0 = mfma32(w[0][ks].p2, ip.p1, a0); a1 = mfma32(w[1][ks].p2, ip.p1, a1); a2 = mfma32(w[2][ks].p2, ip.p1, a2);
                    a0 = mfma32(w[0][ks].p1, ip.p1, a0); a1 = mfma32(w[1][ks].p1, ip.p1, a1); a2 = mfma32(w[2][ks].p1, ip.p1, a2);
                    __builtin_amdgcn_sched_barrier(0);
                }
                /* the chain waves have read this interval's x_c(Q): each of the three counts itself off once per phase-B interval */
                while (__hip_atomic_load(CNT + Q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 3u * nQ[Q]) __builtin_amdgcn_s_sleep(1);
                acc_write(ring(Q, 2, 0), lane, a0);
                acc_write(ring(Q, 2, 1), lane, a1);
                acc_write(ring(Q, 2, 2), lane, a2);
            }
            XSTAMP(sa);
            lds_barrier();
            XSTAMP(sb);
        };
        for (int k = -1; k < nit; k++) {
            interval(std::integral_constant<int, 0>{}, k);
            interval(std::integral_constant<int, 1>{}, k);
        }
    } else {
        /* ------------------------------ L ------------------------------ */
        struct Q3 { f32x4 v[3][4]; };
        int lastT[2] = {0, 0};
        auto fetch = [&](const int s) {
            Q3 q;
            if (c[s].ok) lastT[s] = backward ? c[s].Tt - 1 - c[s].s : c[s].s;
            const float *cb = in + c[s].boff0 * 1536;
            const unsigned o = block_off(s, lastT[s]);
#pragma unroll
            for (int jj = 0; jj < 3; jj++)
#pragma unroll
                for (int g = 0; g < 4; g++) q.v[jj][g] = gload_so(cb, o, (2 * jj + (g >> 1)) * 1024 + (g & 1) * 512);
            if (c[s].ok) {
                c[s].s++;
                if (c[s].s == c[s].s1) { c[s].sgi++; enter(s); }
            }
            return q;
        };
        auto cut_put = [&](unsigned *buf, const Q3 &q) {
#pragma unroll
            for (int jj = 0; jj < 3; jj++) {
                u32x4 p1[2], p2[2];
                cut16(q.v[jj], p1, p2);
                put16(buf, jj, lane, p1, p2);
            }
        };
        enter(0); enter(1);
        Q3 e0 = fetch(0);                                  /* slot 0: block 0 -> IN now, block 1 in flight */
        cut_put(INs(0), e0);
        e0 = fetch(0);
        Q3 e1 = fetch(1);                                  /* slot 1: block 0, written in O(-1) */
        lds_barrier();
        if (STAMP) st0 = __builtin_readcyclecounter();
        /* P in phase A of step kP: the slot's input pieces become those of block kP + 1 (read in P's phase B, the next interval) */
        for (int k = -1; k < nit; k++) {
            cut_put(INs(1), e1);                           /* O(k): P = slot 1 */
            e1 = fetch(1);
            XSTAMP(sa);
            lds_barrier();
            XSTAMP(sb);
            cut_put(INs(0), e0);                           /* E(k): P = slot 0, step k + 1 */
            e0 = fetch(0);
            XSTAMP(sa);
            lds_barrier();
            XSTAMP(sb);
        }
    }
    if (STAMP && dbg && lane == 0) {
        unsigned long long *d = dbg + ((long long)blockIdx.x * 8 + wave) * 16;
        d[0] = sa; d[1] = sb; d[2] = 0; d[3] = 0; d[4] = (unsigned long long)(2 * (nit + 1));
    }
#undef XSTAMP
}

#endif /* SH_GRU32X2_H */
