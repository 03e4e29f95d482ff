/* sh_decode.h -- part of sh_kernels.h (included from there, in this order): transducer Viterbi, traceback, and S1 inside the decoder (k_ff_viterbi).
 * Device code for gfx950 only; see sh_kernels.h for conventions (layouts, split products, citations). */
#ifndef SH_DECODE_H
#define SH_DECODE_H

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
    int dump_final;               /* test hook: a tile's LAST piece also leaves its final scores, start and end state in vstate */
};

__device__ __forceinline__ void argmax_merge(float &v, int &i, float ov, int oi) {
    /* keep the larger value; on equal values the lower index (first wins).
     * Written as selects: as an if() this compiles to exec-mask branches. */
    const bool take = (ov > v) | ((ov == v) & (oi < i));
    v = take ? ov : v;
    i = take ? oi : i;
}

/* (value, index) of the larger value over the four rows of a wave, lower index on equal values, in every lane */
__device__ __forceinline__ void rows_argmax(float &bv, int &bi) {
    float ov = __shfl_xor(bv, 16); int oi = __shfl_xor(bi, 16);
    argmax_merge(bv, bi, ov, oi);
    ov = __shfl_xor(bv, 32); oi = __shfl_xor(bi, 32);
    argmax_merge(bv, bi, ov, oi);
}

#ifndef SH_FVT_GAP
#define SH_FVT_GAP 4.76837158203125e-07f       /* 2^-21: see the one-addition path of k_ff_viterbi */
#endif
/* max of a quad's four new scores for the end state's scan.  Written as v_max3_f32 + v_max_f32: as nested __builtin_fmaxf the compiler cannot see that the
 * values (merged from the two update paths) are canonical and puts a v_max_f32 x, x, x in front of two of them -- five instructions where two do (round 6:
 * 16 of a decoder wave's ~640 VALU instructions per block were those; the kernel is bound by instruction issue: profiles/r6_dual_issue.txt).  Scores are
 * finite (-1e30 at worst), so the two forms agree bit for bit. */
#ifndef SH_MAX4_ASM
#define SH_MAX4_ASM 1        /* 0: the nested __builtin_fmaxf form of rounds 1-5 (A/B: profiles/r6_decoder_issue.txt) */
#endif
__device__ __forceinline__ float d_max4(const f32x4 v) {
#if !SH_MAX4_ASM
    return __builtin_fmaxf(__builtin_fmaxf(v[0], v[1]), __builtin_fmaxf(v[2], v[3]));
#endif
    float t;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(v[0]), "v"(v[1]), "v"(v[2]));
    asm("v_max_f32 %0, %1, %2" : "=v"(t) : "v"(t), "v"(v[3]));
    return t;
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
            if (!sh_wait_flag(a.flag + tile, (unsigned)ord, a.err)) __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
        const float rmf = d_rcp(sum_nx) * mpm1;           /* fin_log's factor */

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
                const float ve = d_max4(ns) - a.local_pen;
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
    if (a.dump_final && a.vstate) {
        float *vst = a.vstate + (long long)tile * (NH * 16 + 32);
#pragma unroll
        for (int i = 0; i < PPT; i++) {
            const int Q = qq + QSTR * i;
            *(f32x4 *)(vst + (Q * 16 + b) * 4) = *(const f32x4 *)(cur + (Q * 16 + b) * 4);
        }
        if (qq == 0) { vst[NH * 16 + b] = pstart; vst[NH * 16 + 16 + b] = pend; }
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
#ifndef SH_FV_STAY_WAVE
#define SH_FV_STAY_WAVE 3   /* the wave that computes the stay state's m-tile (any: its row sum is group 8 whoever adds it; an OLDER wave of its SIMD -- the younger ones set the pace of phase C: -1.4 %) */
#endif
#ifndef SH_FV_LOG_IN_B
#define SH_FV_LOG_IN_B 0    /* 1: S2 (fin_log) of a block's emissions in phase B instead of phase C -- measured SLOWER (12.98 against 12.80 ms: phase B is on the block's critical path, and its LDS round trips do not leave the VALU as idle as its length suggests) */
#endif
#ifndef SH_FV_YOUNG_PRIO
#define SH_FV_YOUNG_PRIO 0
#endif
#ifndef SH_FV_FLIP_PRIO
#define SH_FV_FLIP_PRIO 3   /* 1: the two waves of a SIMD take priority in turns, quad by quad; n >= 2: the younger wave has it for its first n quads of a block, the older one (by age) after that.  0: 12.70, 1: 12.54, 3: 12.52, 4: 12.61, 5: 12.65 ms */
#endif
#ifndef SH_FV_ABL_BANK
#define SH_FV_ABL_BANK 0    /* timing ablation (results invalid unless 0): the two dwords of every ds_read2st64_b32 of the block loop (phase B's suffix scans,
                               phase C's four prefix scores per quad) moved 32 banks apart -- another read's scores, the same instructions: what the
                               bank conflicts the PMC reports cost */
#endif
#ifndef SH_FV_WAHEAD
#define SH_FV_WAHEAD 1      /* m-tiles the S1 weight stream runs ahead of the MFMAs, in the same two register buffers: 1 = tile i + 1 is requested before tile i's products
                               (into the other buffer), 2 = tile i + 2 right behind them (into the buffer they have just read: both buffers are then live through the
                               quad's VALU work -- 64 bytes of scratch, 12.6 against 11.85 ms -- unless SH_FV_BP_REREAD makes room, and that costs 1.0 ms for an
                               LDS round trip in front of every m-tile's products while the doubled distance wins back 0.1: the stream's cost is its volume through
                               the CU's one vector-memory path, not its latency; profiles/r4_decoder_ablations.txt) */
#endif
#ifndef SH_FV_SGB
#define SH_FV_SGB 0         /* n > 0: a scheduling pattern for every quad of the update loop -- each of the nine S1 MFMAs followed by n VALU instructions (left to itself the
                               compiler issues them in clumps at the top of the quad, where the wave waits out the matrix pipe) */
#endif
#ifndef SH_FV_BP_REREAD
#define SH_FV_BP_REREAD (SH_FV_WAHEAD == 2)     /* 1: the next block's trunk column (B operand of the S1 products, 24 VGPRs) is read from LDS again for every m-tile instead of living in
                               registers through the block: room for the second buffer of the weight stream */
#endif
#ifndef SH_FV_ABL
#define SH_FV_ABL 0         /* timing ablations of k_ff_viterbi's block loop (results invalid unless 0): 1 no MFMAs (the next block's logits = the bias), 2 no weight stream
                               (the first m-tile's registers serve all), 4 no transcendentals (v_exp_f32 / v_log_f32 replaced by a multiplication), 8 no traceback store,
                               16 no phase B (suffix maxima not scanned) */
#endif
#ifndef SH_FV_PK
#define SH_FV_PK 0          /* 1: the fast path's additions, fin_log's multiply-add and scaling, and the scaling before v_exp_f32 as packed f32 (same bits) */
#endif
#ifndef SH_FV_SB
#define SH_FV_SB 1          /* scheduling barrier after every SH_FV_SB quads of k_ff_viterbi's update loop (0: none) */
#endif
struct ShFfArgs {
    const float *in;              /* trunk output [ncb][6][64][4] */
    const unsigned *wpiece;       /* S1 weights as pieces [65][3][2][64][4] */
    const float *bfrag;           /* bias fragments x 2^14 [65][64][4] */
    float in_div, out_div;        /* softmax_with_temperature's two divisions (1: none) */
    int no_clamp;                 /* the logits are provably inside exp_ps's clamp (Model::ff_no_clamp and no temperatures, no trunk hook): k_ff_viterbi_teams only */
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
    const float lbound = (mp > 0.0f) ? 1.0e-3f - __logf(mp) : INFINITY;      /* |log-posterior| <= this */
    unsigned long long vA = 0, vB = 0, vC = 0, vD = 0, vt0 = 0, vt1;
#if SH_FV_YOUNG_PRIO
    if (wave >= 4) __builtin_amdgcn_s_setprio(SH_FV_YOUNG_PRIO);      /* experiment: the younger wave of every SIMD at a static higher priority */
#endif

    /* this wave's rows of the S1 weights: 48 KB of fp16 pieces, streamed from L2 once per block, one m-tile (24
     * VGPRs) ahead of the MFMAs that use it.  (The whole matrix is 394 KB -- more than the CU's LDS, and with the
     * decoder's state more than its register file; tools/l2_stream_probe.hip: all 256 CUs re-reading it
     * concurrently take 2.9 us per pass, 35 TB/s aggregate, against ~5 us of decoding per block.) */
    const unsigned *wmine = f.wpiece + (long long)(PPT * wave) * KS * 512;
    ShSplit W[2][KS];
    /* global addresses as (wave-uniform 64-bit base in scalar registers) + (32-bit lane offset): one VGPR serves all */
    const unsigned lofs = (unsigned)lane * 4u, tofs = (unsigned)lane;
    /* (ONE running scalar base, advanced by a tile per call -- the calls come in cyclic tile order 0, 1, .. 7, 0 ..:
     * 48 precomputed bases did not fit the scalar registers and came back through v_readlane, 116 VALU slots per block) */
    typedef const __attribute__((address_space(1))) unsigned *gu32;
    typedef const __attribute__((address_space(1))) u32x4 *gu32x4;
    gu32 wp = (gu32)wmine;
    auto w_load = [&](int i) {
        static_assert(KS == 3, "two bases per tile: immediate offsets reach 4095 bytes");
        gu32 b0 = wp, b1 = wp + 1024;
        asm volatile("" : "+s"(b0), "+s"(b1));       /* ... kept scalar and opaque: else loop-invariant 64-bit VGPR addresses are formed (and spilled) */
        W[i & 1][0].p1 = __builtin_bit_cast(f16x8, *(gu32x4)(b0 + lofs));
        W[i & 1][0].p2 = __builtin_bit_cast(f16x8, *(gu32x4)(b0 + 256 + lofs));
        W[i & 1][1].p1 = __builtin_bit_cast(f16x8, *(gu32x4)(b0 + 512 + lofs));
        W[i & 1][1].p2 = __builtin_bit_cast(f16x8, *(gu32x4)(b0 + 768 + lofs));
        W[i & 1][2].p1 = __builtin_bit_cast(f16x8, *(gu32x4)(b1 + lofs));
        W[i & 1][2].p2 = __builtin_bit_cast(f16x8, *(gu32x4)(b1 + 256 + lofs));
        wp = (i == PPT - 1) ? (gu32)wmine : wp + KS * 512;
        asm volatile("" : "+s"(wp));
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
            if (!sh_wait_flag(a.flag + tile, (unsigned)ord, a.err)) __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    auto e_of = [&](float acc) { return (SH_FV_ABL & 4) ? acc * 1.0e-6f + 1.0f : DIV ? d_exp((acc * SH_OINV) / f.out_div) : d_exp_acc(acc); };   /* no max subtraction (Q2) */
    /* the stay state's m-tile (row 1024 and 15 rows of padding, whose results are masked: only the lanes that
     * hold row 0 of the A operand need real weights -- 384 bytes, kept in LDS): wave 7 */
    auto stay_group = [&](const ShSplit (&bp)[KS], int buf, bool reread = false) {
        if (wave == SH_FV_STAY_WAVE) {
            ShSplit Ws[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                const u32x4 a1 = *(const u32x4 *)(sStay + ((ks * 2 + 0) * 4 + q) * 4), a2 = *(const u32x4 *)(sStay + ((ks * 2 + 1) * 4 + q) * 4);
                const u32x4 z = {0u, 0u, 0u, 0u};
                Ws[ks].p1 = __builtin_bit_cast(f16x8, b == 0 ? a1 : z);
                Ws[ks].p2 = __builtin_bit_cast(f16x8, b == 0 ? a2 : z);
            }
            f32x4 acc = *(const f32x4 *)(sBias + (PPT * NW) * 16 + 4 * q);
            if (SH_FV_BP_REREAD && reread) {
                ShSplit bq[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ks++) bq[ks] = load_pieces(xp + (buf * KS + ks) * 512, lane);
                acc = split_dot<KS>(Ws, bq, acc);
            } else
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
        if (SH_FV_WAHEAD == 2) w_load(1);
#pragma unroll
        for (int i = 0; i < PPT; i++) {
            f32x4 acc = *(const f32x4 *)(sBias + (PPT * wave + i) * 16 + 4 * q);
            if (SH_FV_WAHEAD == 1) w_load((i + 1) & (PPT - 1));
            acc = split_dot<KS>(W[i & 1], bp, acc);
            if (SH_FV_WAHEAD == 2) w_load((i + 2) & (PPT - 1));
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
        /* (the piece's last block is a second copy of this code behind the loop: there the thread index is taken afresh -- kept live
         * across the loop for it, its multiple was the one value the default instantiation spilled: 8 bytes of scratch) */
        int tidb = tid;
        if (!more) { tidb = (int)threadIdx.x; asm volatile("" : "+v"(tidb)); }
        for (int p = tidb; p < NSKIP * 16 && !((SH_FV_ABL & 16) && t > s0); p += NTH) {
            const int j = p >> 4, bb = p & 15;
            float v = cur[((j >> 2) * 16 + bb) * 4 + (j & 3)];
            int ri = 0;
#pragma unroll
            for (int r = 1; r < 16; r++) {
                const int s = r * NSKIP + j;
                const float c = cur[((s >> 2) * 16 + (SH_FV_ABL_BANK ? ((bb + 8 * (r & 1)) & 15) : bb)) * 4 + (s & 3)];
                const bool up = v < c;
                v = up ? c : v;
                ri = up ? r : ri;
            }
            skv[p] = v; ski[p] = ri;
        }
        if (SLIP) {
            for (int p = tidb; p < NSLIP * 16; p += NTH) {
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
#if SH_FV_LOG_IN_B
        {   /* S2 of this block's emissions (normalise, floor, log: three instructions and a v_log_f32 per state) needs the
             * row sums only, not the skip maxima: it runs HERE, under phase B's LDS round trips, instead of in phase C,
             * which is bound by VALU issue.  In place: e[] holds log-posteriors from here on. */
            float totb = 0.0f;
#pragma unroll
            for (int w = 0; w < NW + 1; w++) totb += gsum[(par * (NW + 1) + w) * 16 + b];
            const bool actb = t < myT;
            const float rmb = actb ? d_rcp(totb) * mpm1 : 0.0f, mpb = actb ? mp : 0.0f;
#pragma unroll
            for (int i = 0; i < PPT; i++) {
#pragma unroll
                for (int k = 0; k < 4; k++) e[i][k] = fin_log(e[i][k], rmb, mpb);
            }
        }
#endif
        VSTAMP(vA);
        __syncthreads();
        VSTAMP(vB);

        /* phase C: update my states, cur -> nxt; block t+1's emissions alongside */
        float tot = 0.0f;
#pragma unroll
        for (int w = 0; w < NW + 1; w++) tot += gsum[(par * (NW + 1) + w) * 16 + b];
        const float rmf = d_rcp(tot) * mpm1;                        /* fin_log's factor; v_rcp_f32 in every consumer of the row sum (k_viterbi, k_post_out): the forms keep identical bits */
        const float stay_lp = fin_log(gsum[(par * (NW + 1) + NW) * 16 + b], rmf, mp);
        const bool active = t < myT;
        const unsigned long long actmask = __builtin_amdgcn_ballot_w64(active);
        if (a.hp_side && active && tid < 16) (a.hp_side + (hpo + t) * 5)[4] = stay_lp;
        /* a read past its end keeps its scores: see k_viterbi */
        const float rm = active ? rmf : 0.0f;
        const float mpx = active ? mp : 0.0f;
        (void)rm; (void)mpx;
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
        if (more && !SH_FV_BP_REREAD) {
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
            for (int r = 0; r < 4; r++) sc4_n[r] = cur[(((r * NQ + Q) >> 2) * 16 + (SH_FV_ABL_BANK ? ((b + 8 * (r & 1)) & 15) : b)) * 4 + (Q & 3)];
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
            if (SH_FV_WAHEAD == 1 && !(SH_FV_ABL & 2)) w_load((i + 1) & (PPT - 1));                    /* the next m-tile's weights (after the last: the first, for the next block) */
            if (more && SH_FV_BP_REREAD) {
#pragma unroll
                for (int ks = 0; ks < KS; ks++) bp[ks] = load_pieces(xp + ((par ^ 1) * KS + ks) * 512, lane);
            }
            if (more && !(SH_FV_ABL & 1)) accn = split_dot<KS>(W[i & 1], bp, accn);      /* tile i of block t+1: 9 MFMAs, under the VALU work below */
            if (SH_FV_WAHEAD == 2 && !(SH_FV_ABL & 2)) w_load((i + 2) & (PPT - 1));                    /* the m-tile after next, into the registers those products have read */
            if (AHEAD && i + 1 < PPT) q_fetch(i + 1);
            f32x4 l4;
#if SH_FV_LOG_IN_B
            l4 = e[i];
#else
#if SH_FV_PK
            l4 = fin_log4_pk(e[i], rm, mpx);
#else
#pragma unroll
            for (int k = 0; k < 4; k++) l4[k] = (SH_FV_ABL & 4) ? __builtin_fmaf(e[i][k], rm, mpx) * -3.0f : fin_log(e[i][k], rm, mpx);
#endif
#endif
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
            f32x4 ns;
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
             * maximum m, PROVIDED no other candidate x < m rounds to the same sum.  Two reals further apart than an ulp
             * of the larger one round to different floats; a sum l + x has |l + x| <= |l|max + |x|, its ulp is at most
             * 2^-23 of that, and the runner-up md = m - (m - md): so m - md > 2^-23 (|l|max + |m|) (1 + 2^-23) is enough.
             * Quads where the runner-up is within SH_FVT_GAP (|l|max + |m|) = 2^-21 (...) of m -- four times that bound,
             * which also covers the rounding of the test's own three operations -- or an emission is -inf, IN ANY LANE,
             * take the reference's compare-by-compare form below; the others get by with 5 instead of 13 operations per
             * state.  (2^-22 would do by the argument above; measured on the bench workload it changes nothing --
             * 9.71-9.81 against 9.69-9.72 ms -- so the margin stays: profiles/r6_decoder_issue.txt.)  Reads past their end have
             * l = -inf: every move loses against stay in either form, so they do not count. */
            bool fast = false;
            float m = 0.f;
            unsigned cm = cstart;
            if (!SLIP && SKIP0) {
                m = __builtin_fmaxf(__builtin_fmaxf(sv, kv), pstart);
                const float md = __builtin_amdgcn_fmed3f(sv, kv, pstart);
                /* max |l| of the quad: l = log(min_prob + ..) lies in [log min_prob, ~0], so -log min_prob bounds it without
                 * looking (min_prob = 0: the bound is infinite and every quad takes the compare-by-compare form) */
                /* lanes whose gap is NOT clear (<=, or unordered: NaN / inf), as a lane mask straight from the comparison */
                const unsigned long long unclear = __builtin_amdgcn_fcmpf(m - md, (lbound + __builtin_fabsf(m)) * SH_FVT_GAP, 13 /* ULE */);
                fast = (unclear & actmask) == 0;
                cm = (kv == m) ? cskip : cm;
                cm = (sv == m) ? cstep : cm;
            }
            if (fast && SH_FV_PK) {
                const f32x2 s01 = (f32x2){pv[0], pv[1]} + stay_v, s23 = (f32x2){pv[2], pv[3]} + stay_v;      /* stay  :180 */
                const f32x2 m01 = (f32x2){l4[0], l4[1]} + m, m23 = (f32x2){l4[2], l4[3]} + m;                /* the best move into the state */
                SH_CODE_LT(0, codes, s01[0], m01[0], cm); ns[0] = __builtin_fmaxf(s01[0], m01[0]);
                SH_CODE_LT(1, codes, s01[1], m01[1], cm); ns[1] = __builtin_fmaxf(s01[1], m01[1]);
                SH_CODE_LT(2, codes, s23[0], m23[0], cm); ns[2] = __builtin_fmaxf(s23[0], m23[0]);
                SH_CODE_LT(3, codes, s23[1], m23[1], cm); ns[3] = __builtin_fmaxf(s23[1], m23[1]);
            } else if (fast) {
#define SH_FV_FAST(E)                                                                                           \
                {                                                                                               \
                    const float sc = pv[E] + stay_v;        /* stay  :180 */                                    \
                    const float mv = l4[E] + m;             /* the best move into the state */                  \
                    SH_CODE_LT(E, codes, sc, mv, cm);                                                           \
                    ns[E] = __builtin_fmaxf(sc, mv);                                                            \
                }
                SH_FV_FAST(0) SH_FV_FAST(1) SH_FV_FAST(2) SH_FV_FAST(3)
#undef SH_FV_FAST
            } else { SH_FV_STATE(0) SH_FV_STATE(1) SH_FV_STATE(2) SH_FV_STATE(3) }
#undef SH_FV_STATE
            *(f32x4 *)(nxt + (Q * 16 + b) * 4) = ns;
            if (!(SH_FV_ABL & 8)) (a.tb + (cb * NQ + 32 * wave + 4 * i) * 16)[tofs] = codes;   /* also for reads past their end (never read back): no branch */
            else asm volatile("" :: "v"(codes));
            {   /* next block's end-state scan, per quad: this thread meets its quads in increasing index order,
                 * so a strict compare keeps the first maximum */
                const float ve = d_max4(ns) - a.local_pen;
                bi = (ve > bv) ? Q : bi;
                bv = __builtin_fmaxf(bv, ve);
            }
            if (more) {                                     /* block t's emissions of this quad are used up: in place */
#if SH_FV_PK && SH_FAST_MATH
                if (!DIV) {
                    f32x4 c;
#pragma unroll
                    for (int r = 0; r < 4; r++) c[r] = __builtin_amdgcn_fmed3f(accn[r], -88.3762626647949f * SH_OSCALE, 88.3762626647949f * SH_OSCALE);
                    const f32x2 c01 = (f32x2){c[0], c[1]} * (1.44269504088896341f * SH_OINV), c23 = (f32x2){c[2], c[3]} * (1.44269504088896341f * SH_OINV);
                    e[i] = (f32x4){__builtin_amdgcn_exp2f(c01[0]), __builtin_amdgcn_exp2f(c01[1]), __builtin_amdgcn_exp2f(c23[0]), __builtin_amdgcn_exp2f(c23[1])};
                } else
#endif
#pragma unroll
                for (int r = 0; r < 4; r++) e[i][r] = e_of(accn[r]);
                part += (e[i][0] + e[i][1]) + (e[i][2] + e[i][3]);
            }
            if (SH_FV_SGB && more) {
#pragma unroll
                for (int k = 0; k < 9; k++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          /* one MFMA */
                    __builtin_amdgcn_sched_group_barrier(0x002, SH_FV_SGB, 0);  /* n VALU */
                }
            }
            if (SH_FV_SB && (i % SH_FV_SB) == SH_FV_SB - 1) __builtin_amdgcn_sched_barrier(0);          /* quads one after the other: register budget */
#if SH_FV_FLIP_PRIO == 1        /* in turns, quad by quad */
            if (wave >= 4) { if (i & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
            else { if (i & 1) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1); }
#elif SH_FV_FLIP_PRIO >= 2      /* the younger wave has priority for its first SH_FV_FLIP_PRIO quads of a block, the older one (by age) after that */
            if (wave >= 4) { if (i == PPT - 1) __builtin_amdgcn_s_setprio(1); else if (i == SH_FV_FLIP_PRIO - 1) __builtin_amdgcn_s_setprio(0); }
#endif
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
            stay_group(bp, par ^ 1, true);
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
        if (a.dump_final && a.vstate) {
            float *vst = a.vstate + (long long)tile * (NH * 16 + 32);
#pragma unroll
            for (int i = 0; i < PPT; i++) {
                const int Q = 32 * wave + 4 * i + q;
                *(f32x4 *)(vst + (Q * 16 + b) * 4) = *(const f32x4 *)(cur + (Q * 16 + b) * 4);
            }
            if (tid < 16) { vst[NH * 16 + b] = pstart; vst[NH * 16 + 16 + b] = pend; }
        }
    }
    if (a.dbg && lane == 0) { unsigned long long *d = a.dbg + ((long long)blockIdx.x * NW + wave) * 8; d[0] = vA; d[1] = vB; d[2] = vC; d[3] = vD; d[4] = (unsigned long long)(s1 - s0); }
}

/* viterbi_local_backtrace (decode.c:58-98) of one read by one thread */
__device__ __forceinline__ void backtrace_read(const unsigned *__restrict__ tb, const int *__restrict__ tb_end,
                            const int *__restrict__ final_state, const ShMeta &md,
                            const long long *__restrict__ seq_off, int *__restrict__ seq,
                            int rd, int NQ, int sstride) {
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
        if (state >= 0) { out[(long long)(ri + 1) * sstride] = last; last = state; }
        else out[(long long)(ri + 1) * sstride] = -1;
    }
    out[0] = last;
    for (int i = 0; i < T; i++) { if (out[(long long)i * sstride] == NH) out[(long long)i * sstride] = -1; else break; }
    for (int i = T; i >= 0; i--) { if (out[(long long)i * sstride] == NH + 1) out[(long long)i * sstride] = -1; else break; }
}
/* ... one thread per read */
__global__ __attribute__((amdgpu_num_vgpr(16))) void k_backtrace(const unsigned *__restrict__ tb, const int *__restrict__ tb_end,
                            const int *__restrict__ final_state, ShMeta md,
                            const long long *__restrict__ seq_off, int *__restrict__ seq,
                            int npad, int NQ, int sstride) {
    const int rd = blockIdx.x * blockDim.x + threadIdx.x;
    if (rd >= npad) return;
    backtrace_read(tb, tb_end, final_state, md, seq_off, seq, rd, NQ, sstride);
}

/* the same walk for lane 0 of every tile only (tiles whose sixteen lanes run ONE read: the per-read decode_transducer, coalesced) */
__global__ __attribute__((amdgpu_num_vgpr(16))) void k_backtrace_lane0(const unsigned *__restrict__ tb, const int *__restrict__ tb_end,
                            const int *__restrict__ final_state, ShMeta md,
                            const long long *__restrict__ seq_off, int *__restrict__ seq, int ntile, int NQ) {
    const int tile = blockIdx.x * blockDim.x + threadIdx.x;
    if (tile >= ntile) return;
    const int rd = tile * 16;
    const int T = md.rT[rd];
    if (T <= 0) return;
    const long long boff = md.tile_boff[tile];
    const int NH = 4 * NQ;
    int *out = seq + seq_off[rd];
    const unsigned char *tbb = (const unsigned char *)tb;
    int last = final_state[rd];
    for (int ri = T - 1; ri >= 0; ri--) {
        int state;
        if (last < NH) {
            const unsigned code = tbb[(((boff + ri) * NQ + (last >> 2)) * 16) * 4 + (last & 3)];
            if (code == SH_TB_STAY) state = -1;
            else if (code < SH_TB_SKIP) state = (int)(code - SH_TB_STEP) * (NH / 4) + (last >> 2);
            else if (code < SH_TB_SLIP) state = (int)(code - SH_TB_SKIP) * (NH / 16) + (last >> 4);
            else if (code < SH_TB_START) state = (int)(code - SH_TB_SLIP) * (NH / 64) + (last >> 6);
            else state = NH;
        } else if (last == NH) {
            state = NH;                                    /* decode.c:328 */
        } else {
            state = tb_end[(boff + ri) * 16];
        }
        if (state >= 0) { out[ri + 1] = last; last = state; }
        else out[ri + 1] = -1;
    }
    out[0] = last;
    for (int i = 0; i < T; i++) { if (out[i] == NH) out[i] = -1; else break; }
    for (int i = T; i >= 0; i--) { if (out[i] == NH + 1) out[i] = -1; else break; }
}

#endif /* SH_DECODE_H */
