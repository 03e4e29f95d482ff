/* sh_decode_teams.h -- part of sh_kernels.h (included from there, behind sh_decode.h): S1 + S2 + D1 forward in one kernel on TWO TEAMS of waves.
 * Device code for gfx950 only; see sh_kernels.h for conventions (layouts, split products, citations).
 *
 * k_ff_viterbi (sh_decode.h) lets each of its eight waves do everything: stream its rows of the S1 weights, multiply, exponentiate AND decode
 * -- 256 VGPRs a wave, two waves per SIMD, 28 % of wave time in s_waitcnt vmcnt on a weight stream only the products need
 * (profiles/r4_decoder_ablations.txt).  Here the work is divided the way k_gru_proj divides a recurrent layer:
 *
 *  * S1 TEAM, waves 8-11 (one per SIMD).  Producer p owns m-tiles 16p .. 16p+15 of the 1040 x 96 output layer (producer 3 also the stay
 *    state's tile): their fp16 pieces stream from L2 through a ring of SH_FVT_WBUF register buffers, three m-tiles ahead of the products; the
 *    products, the clamp + v_exp_f32 and the row sums are those of k_ff_viterbi, operation for operation (same association of the row sums:
 *    groups of SH_SUM_GROUP m-tiles), so every S1 form gives identical bits.  Block t+1's exp values go into a ONE-BLOCK LDS ring (64 KB,
 *    [m-tile][lane][4] = the MFMA result image) while block t is decoded.
 *  * DECODER TEAM, waves 0-7 (two per SIMD): no MFMA, no weight stream, no vector-memory wait in the block loop (its only global traffic is
 *    the traceback store).  It takes a block's emissions out of the ring into registers in phase B -- behind the barrier the ring is free
 *    for the next block -- and updates the scores IN PLACE: one 64 KB score buffer instead of two.
 *
 * What makes the in-place update possible is the division of the states.  Thread (wave w, q, read b) owns the eight quads
 *      Q = 64 rl + 32 c + 4 w + q,   rl < 4, c < 2        (a quad = the four one-base extensions of one 4-mer)
 * i.e. m-tiles 16 rl + 8 c + w (lane = 16 q + b: the MFMA result layout again).  Then
 *    step  into quad Q comes from states r4 * 256 + Q, r4 < 4: their maximum M1[Q] is computed by Q's own thread in phase B, BEFORE anything is
 *          overwritten, and stays in its registers (decode.c:186-210);
 *    skip  into quad Q' comes from states r * 64 + j, r < 16, j = Q' >> 2 (decode.c:228-262), and with r = 4 r4 + rl that is the maximum over rl
 *          of M1[64 rl + j] -- for j = 32 c + 4 w + q all four are this thread's own.  So the skip maxima cost three merges per thread and
 *          value, not a second scan of the scores, and only they (4 KB + prefixes) cross threads through LDS;
 *    stay  reads the thread's own state.
 * Phase C therefore reads no other thread's score: it may overwrite its own.  Ties: the step maximum keeps the lowest r4 (strict compare
 * in scan order), the skip merge the lowest r = 4 r4 + rl among equal values -- the reference's first-maximum order (Q7).
 * The end state's traceback entry, the one place that reads another thread's quad, is fetched in phase B as well.
 *
 * Two LDS-only barriers per block, shared by both teams (gfx950 has no named barriers):
 *    phase B   decoders: ring -> registers, step / skip maxima      producers: cut the next trunk column into pieces (waves 8-10)
 *    phase C   decoders: S2 + update in place + traceback           producers: S1 of block t+1 -> ring, row sums
 * LDS: scores 64 KB + ring 64 KB + skip maxima 8 KB + bias 4 KB + trunk pieces 6 KB + sums / end-state scan 3 KB = 149 KB.
 * 12 waves x <= 168 VGPRs, three waves per SIMD.  Not built for the slip move (k_ff_viterbi keeps that). */
#ifndef SH_DECODE_TEAMS_H
#define SH_DECODE_TEAMS_H

#ifndef SH_FVT_WBUF
#define SH_FVT_WBUF 4        /* register buffers of the S1 weight stream (24 VGPRs each): the stream runs SH_FVT_WBUF - 1 m-tiles ahead of the products */
#endif
#ifndef SH_FVT_PROD_PRIO
#define SH_FVT_PROD_PRIO 1   /* s_setprio of the S1 team: its waves are the youngest of their SIMDs (issue arbitration: priority, then age) and a late producer holds up the barrier */
#endif
#ifndef SH_FVT_ABL
#define SH_FVT_ABL 0         /* timing ablations (results invalid unless 0): 1 producers idle (the ring keeps the first block's emissions), 2 no phase B scans, 4 no traceback store */
#endif
#ifndef SH_FVT_BAR3
#define SH_FVT_BAR3 0        /* 1: a third barrier per block, right behind the decoders' ring drain: the ring is free for the S1 team from there, i.e. the producers also work
                                through the decoders' scans (SH_FVT_NB1 m-tiles in front of the scan barrier) */
#endif
#ifndef SH_FVT_NB1
#define SH_FVT_NB1 3
#endif
#ifndef SH_FVT_FLIP
#define SH_FVT_FLIP 0        /* n > 0: the younger decoder wave of a SIMD (waves 4-7) has priority for its first n quads of a block, the older one (by age) after that */
#endif
#ifndef SH_FVT_STAMP
#ifdef SH_EXPERIMENTS
#define SH_FVT_STAMP 1       /* per-wave cycle stamps of the phases (SH_VIT_STAMP=1): the experiments build only -- their accumulators are ten scalar registers */
#else
#define SH_FVT_STAMP 0
#endif
#endif
#define SH_FVT_NTH 768
#define SH_FVT_LDS_FLOATS (1024 * 16 + 64 * 256 + 2 * 64 * 16 + 2 * 2 * 8 * 16 + 2 * 9 * 16 + 65 * 16 + 3 * 512 + 3 * 2 * 4 * 4)

template <bool SKIP0, bool DIV>
__global__ __launch_bounds__(SH_FVT_NTH) void k_ff_viterbi_teams(ShFfArgs f, ShVitArgs a, ShMeta md) {
    constexpr int NCW = 8, NPW = 4, PPT = 8, NQ = 256, NH = 1024, KS = 3, KQ = 6, TPP = 16, NG = 9;
    static_assert(2 * SH_SUM_GROUP == TPP, "a producer's tiles are two row-sum groups");
    static_assert(SH_FVT_WBUF >= 2 && SH_FVT_WBUF <= 4 && TPP % SH_FVT_WBUF == 0, "weight ring");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sc = smem;                                   /* scores, ONE buffer: quad Q of read b at (Q * 16 + b) * 4 */
    float *ring = sc + NH * 16;                         /* a block's exp values [m-tile < 64][lane][4] */
    float *skk = ring + 64 * 256;                       /* skip maxima [suffix j < 64][read][value, prefix] */
    float *redv = skk + 2 * 64 * 16;                    /* end-state scan [2][NCW][16] */
    int *redi = (int *)(redv + 2 * NCW * 16);
    float *gsum = (float *)(redi + 2 * NCW * 16);       /* row-sum groups [2][NG][16] (group 8 = the stay state's exp value) */
    float *sBias = gsum + 2 * NG * 16;                  /* bias x 2^14 by state row [65 * 16] */
    unsigned *xp = (unsigned *)(sBias + 65 * 16);       /* the next trunk column as pieces [KS][2][64][4] */
    unsigned *sStay = xp + KS * 512;                    /* row 1024 of the weights as pieces [KS][2][4 k groups][4] */

    const int tid = threadIdx.x, lane = tid & 63, b = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool decoder = wave < NCW;
    const float mp = a.min_prob, mpm1 = 1.0f - a.min_prob;
    const float lbound = (mp > 0.0f) ? 1.0e-3f - __logf(mp) : INFINITY;      /* |log-posterior| <= this */
    unsigned long long vA = 0, vB = 0, vC = 0, vD = 0, vt0 = 0, vt1;
#undef VSTAMP
#if SH_FVT_STAMP
#define VSTAMP(acc) do { if (a.dbg) { vt1 = __builtin_readcyclecounter(); acc += vt1 - vt0; vt0 = vt1; } } while (0)
#else
#define VSTAMP(acc) do { } while (0)
#endif

    if (tid < KS * 2 * 4 * 4) sStay[tid] = f.wpiece[(long long)(64) * KS * 512 + (tid >> 4) * 256 + ((tid >> 2) & 3) * 64 + (tid & 3)];
    for (int j = tid; j < 65 * 16; j += SH_FVT_NTH) sBias[j] = f.bfrag[((j >> 4) * 64 + ((j >> 2) & 3) * 16) * 4 + (j & 3)];

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

    /* the decoder team's quads: i = 2 rl + c, in increasing order of Q */
    const int cw = decoder ? wave : 0;
    float *myq = sc + cw * 256 + lane * 4;                       /* quad i of this thread: + (16 rl + 8 c) * 256 */
    const float *mye = ring + cw * 256 + lane * 4;               /* ... its emissions in the ring */
    const float *mysrc = sc + cw * 64 + b * 4 + q;               /* state r4 * 256 + Q: + (r4 * 64 + 16 rl + 8 c) * 64 */
#define SH_FVT_MT(i) (16 * ((i) >> 1) + 8 * ((i) & 1))           /* m-tile of quad i, less the wave */

    if (s0 == 0) {
        /* decode.c:155-159 */
        if (decoder) {
#pragma unroll
            for (int i = 0; i < PPT; i++) *(f32x4 *)(myq + SH_FVT_MT(i) * 256) = (f32x4){-SH_BIG, -SH_BIG, -SH_BIG, -SH_BIG};
            if (lane < 16) { redv[wave * 16 + b] = -SH_BIG - a.local_pen; redi[wave * 16 + b] = 4 * wave; }
        }
    } else {
        /* the tile's earlier blocks ran on another workgroup: take over its state */
        if (tid == 0) {
            if (!sh_wait_flag(a.flag + tile, (unsigned)ord, a.err)) __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (decoder) {
            const float *vst = a.vstate + (long long)tile * (NH * 16 + 32);
            float bv = -INFINITY;
            int bi = 0x7fffffff;
#pragma unroll
            for (int i = 0; i < PPT; i++) {
                const int Q = 4 * (SH_FVT_MT(i) + cw) + q;
                const f32x4 pv = *(const f32x4 *)(vst + (Q * 16 + b) * 4);
                *(f32x4 *)(myq + SH_FVT_MT(i) * 256) = pv;
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
            if (lane < 16) { redv[((s0 & 1) * NCW + wave) * 16 + b] = bv; redi[((s0 & 1) * NCW + wave) * 16 + b] = bi; }
        }
    }

    if (!decoder) {
        /* ================================================================== */
        /* S1 team                                                              */
        /* ================================================================== */
        const int pw = wave - NCW;
#if SH_FVT_PROD_PRIO
        __builtin_amdgcn_s_setprio(SH_FVT_PROD_PRIO);
#endif
        /* this wave's rows of the S1 weights: 96 KB of fp16 pieces per block, from L2, through SH_FVT_WBUF register buffers.  Global addresses
         * as (ONE running wave-uniform base in scalar registers, advanced by a tile per call -- the calls come in cyclic tile order) + (32-bit
         * lane offset): see k_ff_viterbi */
        const unsigned *wmine = f.wpiece + (long long)(TPP * pw) * KS * 512;
        ShSplit W[SH_FVT_WBUF][KS];
        const unsigned lofs = (unsigned)lane * 4u;
        typedef const __attribute__((address_space(1))) unsigned *gu32;
        typedef const __attribute__((address_space(1))) u32x4 *gu32x4;
        gu32 wp = (gu32)wmine;
        auto w_load = [&](int k) {
            static_assert(KS == 3, "two bases per tile: immediate offsets reach 4095 bytes");
            gu32 b0 = wp, b1 = wp + 1024;
            asm volatile("" : "+s"(b0), "+s"(b1));
            W[k % SH_FVT_WBUF][0].p1 = __builtin_bit_cast(f16x8, *(gu32x4)(b0 + lofs));
            W[k % SH_FVT_WBUF][0].p2 = __builtin_bit_cast(f16x8, *(gu32x4)(b0 + 256 + lofs));
            W[k % SH_FVT_WBUF][1].p1 = __builtin_bit_cast(f16x8, *(gu32x4)(b0 + 512 + lofs));
            W[k % SH_FVT_WBUF][1].p2 = __builtin_bit_cast(f16x8, *(gu32x4)(b0 + 768 + lofs));
            W[k % SH_FVT_WBUF][2].p1 = __builtin_bit_cast(f16x8, *(gu32x4)(b1 + lofs));
            W[k % SH_FVT_WBUF][2].p2 = __builtin_bit_cast(f16x8, *(gu32x4)(b1 + 256 + lofs));
            wp = (k == TPP - 1) ? (gu32)wmine : wp + KS * 512;
            asm volatile("" : "+s"(wp));
        };
        /* the trunk column of a block, k step min(pw, 2), as raw fp32 (the load is unconditional: straight-line vmcnt accounting) ... */
        const int xks = pw < KS ? pw : KS - 1;
        f32x4 xr0, xr1;
        auto xraw_load = [&](int t) {
            const float *p = f.in + ((boff + min(t, s1 - 1)) * KQ + 2 * xks) * 256;       /* uniform */
            xr0 = *(const f32x4 *)(p + lofs);
            xr1 = *(const f32x4 *)(p + 256 + lofs);
        };
        /* ... cut into pieces for the team */
        auto xp_publish = [&]() {
            if (pw < KS) {
                f32x4 v0 = xr0, v1 = xr1;
                if (DIV) { v0 = v0 / f.in_div; v1 = v1 / f.in_div; }      /* shift_scale_matrix_inplace: division (Q5); x / 1 = x */
                const ShSplit sp = split8(v0, v1);
                unsigned *d = xp + pw * 512 + lane * 4;
                *(u32x4 *)d = __builtin_bit_cast(u32x4, sp.p1);
                *(u32x4 *)(d + 256) = __builtin_bit_cast(u32x4, sp.p2);
            }
        };
        auto e_of = [&](float acc) { return DIV ? d_exp((acc * SH_OINV) / f.out_div) : d_exp_acc(acc); };   /* no max subtraction (Q2) */
        auto group_out = [&](float part, int buf, int g) {
            float v = part;
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (lane < 16) gsum[(buf * NG + g) * 16 + b] = v;
        };
        /* one block's S1: 16 m-tiles -> ring, two row-sum groups; producer 3: the stay state's tile (row 1024 and 15 rows of padding, whose
         * results are masked: only the lanes that hold row 0 of the A operand need real weights -- 384 bytes, kept in LDS) */
        float *myring = ring + (TPP * pw) * 256 + lane * 4;
        const float *mybias = sBias + (TPP * pw) * 16 + 4 * q;
        /* one block's S1: 16 m-tiles -> ring, two row-sum groups; producer 3: the stay state's tile.  INLOOP: called from the block loop (with
         * SH_FVT_BAR3 the scan barrier then falls behind tile SH_FVT_NB1 - 1). */
        auto s1_block = [&](int buf, auto inloop_c) {
            constexpr bool INLOOP = decltype(inloop_c)::value;
            ShSplit bp[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ks++) bp[ks] = load_pieces(xp + ks * 512, lane);
            float part = 0.0f;
            /* software pipeline over the m-tiles: the nine dependent products of tile k are issued in front of the clamp / exp / store / sum of
             * tile k - 1 (a wave issues in order: behind a dependent MFMA nothing of the wave issues, so the VALU work belongs between them) */
            auto tile_out = [&](int k, const f32x4 &acc) {
                f32x4 ex;
#pragma unroll
                for (int r = 0; r < 4; r++) ex[r] = e_of(acc[r]);
                *(f32x4 *)(myring + k * 256) = ex;
                part += (ex[0] + ex[1]) + (ex[2] + ex[3]);
                if (k % SH_SUM_GROUP == SH_SUM_GROUP - 1) { group_out(part, buf, 2 * pw + k / SH_SUM_GROUP); part = 0.0f; }
            };
            f32x4 acc_prev = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < TPP; k++) {
                f32x4 acc = *(const f32x4 *)(mybias + k * 16);
                w_load((k + SH_FVT_WBUF - 1) % TPP);          /* into the buffer tile k - 1's products have read */
                acc = split_dot<KS>(W[k % SH_FVT_WBUF], bp, acc);
                if (k > 0) tile_out(k - 1, acc_prev);
                acc_prev = acc;
                if (SH_FVT_BAR3 && INLOOP && k == SH_FVT_NB1) { VSTAMP(vA); lds_barrier(); VSTAMP(vB); }      /* the decoders' scans are done */
                __builtin_amdgcn_sched_barrier(0);
            }
            tile_out(TPP - 1, acc_prev);
            if (pw == NPW - 1) {
                /* the stay state's tile (row 1024 and 15 rows of padding, whose results are masked: only the lanes that hold row 0 of the A
                 * operand need real weights -- 384 bytes, kept in LDS) */
                ShSplit Ws[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
                    const u32x4 a1 = *(const u32x4 *)(sStay + ((ks * 2 + 0) * 4 + q) * 4), a2 = *(const u32x4 *)(sStay + ((ks * 2 + 1) * 4 + q) * 4);
                    const u32x4 z = {0u, 0u, 0u, 0u};
                    Ws[ks].p1 = __builtin_bit_cast(f16x8, b == 0 ? a1 : z);
                    Ws[ks].p2 = __builtin_bit_cast(f16x8, b == 0 ? a2 : z);
                }
                f32x4 acc = *(const f32x4 *)(sBias + 64 * 16 + 4 * q);
                acc = split_dot<KS>(Ws, bp, acc);
                f32x4 ex;
#pragma unroll
                for (int r = 0; r < 4; r++) ex[r] = (4 * q + r < 1) ? e_of(acc[r]) : 0.0f;        /* rows >= NS are padding */
                group_out((ex[0] + ex[1]) + (ex[2] + ex[3]), buf, NG - 1);
            }
        };

        if (s1 > s0) xraw_load(s0);
#pragma unroll
        for (int k = 0; k < SH_FVT_WBUF - 1; k++) w_load(k);
        __syncthreads();                                    /* P0: bias, stay row, initial scores */
        if (s1 > s0) xp_publish();
        lds_barrier();                                      /* X1 */
        if (s1 > s0) {
            xraw_load(s0 + 1);
            s1_block(s0 & 1, std::false_type{});
            if (SH_FVT_ABL & 1) s1_block((s0 & 1) ^ 1, std::false_type{});     /* (ablation: valid row sums in both slots) */
        }
        lds_barrier();                                      /* X2: block s0's emissions are in the ring */
        if (a.dbg) vt0 = __builtin_readcyclecounter();
        for (int t = s0; t < s1; t++) {
            const bool more = (t + 1 < s1) && !((SH_FVT_ABL & 1));
            if (more) xp_publish();                         /* block t+1 (the pieces of block t were read a phase ago) */
#if SH_FVT_BAR3
            lds_barrier();                                  /* the decoders have taken block t out of the ring */
            if (more) {
                xraw_load(t + 2);
                s1_block((t + 1) & 1, std::true_type{});    /* (the scan barrier is inside) */
            } else { VSTAMP(vA); lds_barrier(); VSTAMP(vB); }
#else
            VSTAMP(vA);
            lds_barrier();
            VSTAMP(vB);
            if (more) {
                xraw_load(t + 2);
                s1_block((t + 1) & 1, std::false_type{});
            }
#endif
            VSTAMP(vC);
            lds_barrier();
            VSTAMP(vD);
        }
        /* the decoder team's epilogue (hand-over / final state) has two workgroup barriers either way */
        __syncthreads();
        if (s1 >= Tt) __syncthreads();
        if (a.dbg && lane == 0) { unsigned long long *d = a.dbg + ((long long)blockIdx.x * 12 + wave) * 8; d[0] = vA; d[1] = vB; d[2] = vC; d[3] = vD; d[4] = (unsigned long long)(s1 - s0); }
        return;
    }
    {
        /* ================================================================== */
        /* decoder team                                                         */
        /* ================================================================== */
        __syncthreads();                                    /* P0 */
        lds_barrier();                                      /* X1 */
        lds_barrier();                                      /* X2 */
        if (a.dbg) vt0 = __builtin_readcyclecounter();
        const unsigned tofs = (unsigned)lane;
        for (int t = s0; t < s1; t++) {
            const long long cb = boff + t;
            const int par = t & 1;

            /* ---- phase B: this block's emissions out of the ring; step maxima of my quads; skip maxima of my two suffixes ---- */
            f32x4 e[PPT];
#pragma unroll
            for (int i = 0; i < PPT; i++) e[i] = *(const f32x4 *)(mye + SH_FVT_MT(i) * 256);
#if SH_FVT_BAR3
            lds_barrier();                                  /* the ring is the S1 team's again */
#endif
            float sv[PPT];
            int sr[PPT];
#pragma unroll
            for (int i = 0; i < PPT; i++) {
                /* step: max over the 4 prefixes of suffix Q, the first maximum (decode.c:186-210) */
                float v = mysrc[SH_FVT_MT(i) * 64];
                int ri = 0;
                if (!(SH_FVT_ABL & 2)) {
#pragma unroll
                    for (int r = 1; r < 4; r++) {
                        const float c = mysrc[(r * 64 + SH_FVT_MT(i)) * 64];
                        const bool up = v < c;
                        v = up ? c : v;
                        ri = up ? r : ri;
                    }
                }
                sv[i] = v; sr[i] = ri;
            }
#pragma unroll
            for (int c = 0; c < 2; c++) {
                /* skip into suffix j = 32 c + 4 w + q: the prefixes are r = 4 r4 + rl, lowest first (decode.c:228-251) */
                float v = sv[c];
                int ri = 4 * sr[c];
                if (!(SH_FVT_ABL & 2)) {
#pragma unroll
                    for (int rl = 1; rl < 4; rl++) argmax_merge(v, ri, sv[2 * rl + c], 4 * sr[2 * rl + c] + rl);
                }
                *(f32x2 *)(skk + ((32 * c + 4 * cw + q) * 16 + b) * 2) = (f32x2){v, __builtin_bit_cast(float, ri)};
            }
            /* the end state's predecessor (decode.c:343-348) reads a quad of another thread: taken here, before anything is overwritten */
            float ev = 0.f; int ei = 0; f32x4 q4 = {0.f, 0.f, 0.f, 0.f};
            if (cw == 0) {
                ev = redv[par * NCW * 16 + b];
                ei = redi[par * NCW * 16 + b];
                for (int w = 1; w < NCW; w++) argmax_merge(ev, ei, redv[(par * NCW + w) * 16 + b], redi[(par * NCW + w) * 16 + b]);
                q4 = *(const f32x4 *)(sc + ((ei & (NQ - 1)) * 16 + b) * 4);
            }
            VSTAMP(vA);
            lds_barrier();
            VSTAMP(vB);

            /* ---- phase C: S2 + update of my states, in place ---- */
            float tot = 0.0f;
#pragma unroll
            for (int w = 0; w < NG; w++) tot += gsum[(par * NG + w) * 16 + b];
            const float rmf = d_rcp(tot) * mpm1;                        /* fin_log's factor; v_rcp_f32 in every consumer of the row sum: the forms keep identical bits */
            const float stay_lp = fin_log(gsum[(par * NG + NG - 1) * 16 + b], rmf, mp);
            const bool active = t < myT;
            const unsigned long long actmask = __builtin_amdgcn_ballot_w64(active);
            if (a.hp_side && active && tid < 16) (a.hp_side + (hpo + t) * 5)[4] = stay_lp;
            /* a read past its end keeps its scores: see k_viterbi */
            const float rm = active ? rmf : 0.0f;
            const float mpx = active ? mp : 0.0f;
            const float stay_v = active ? stay_lp - a.stay_pen : 0.0f;  /* decode.c:175-176 */
            const float stay_act = stay_lp - a.stay_pen;
            const float hold = fmaxf(-a.local_pen, stay_act);
            const float nstart = pstart + hold;                 /* decode.c:326 */
            if (cw == 0) {
                float nend = pend + hold;                       /* decode.c:339 */
                const bool enter_end = ev > nend;               /* decode.c:343-348 */
                nend = enter_end ? ev : nend;
                if (active && tid < 16) {
                    int tbe = NH + 1;
                    if (enter_end) {
                        int e0 = 3;
                        e0 = (q4[2] - a.local_pen == ev) ? 2 : e0;
                        e0 = (q4[1] - a.local_pen == ev) ? 1 : e0;
                        e0 = (q4[0] - a.local_pen == ev) ? 0 : e0;
                        tbe = 4 * ei + e0;
                    }
                    a.tb_end[cb * 16 + b] = tbe;
                }
                if (active) pend = nend;
            }
            float bv = -INFINITY;
            int bi = 0x7fffffff;
            /* a quad's inputs from LDS are read one quad ahead */
            f32x4 pv_n; f32x2 kk_n;
            auto q_fetch = [&](int i) {
                pv_n = *(const f32x4 *)(myq + SH_FVT_MT(i) * 256);
                kk_n = *(const f32x2 *)(skk + ((SH_FVT_MT(i) + cw) * 16 + b) * 2);
            };
            q_fetch(0);
            float hpv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < PPT; i++) {
                const int QB = 4 * (SH_FVT_MT(i) + cw);            /* first quad of the wave's m-tile (uniform) */
                const int Q = QB + q;
                const f32x4 pv = pv_n;
                const float kv = kk_n[0];
                const float krf = kk_n[1];                     /* (bit_cast straight from the vector element reads element 0: hipcc 7.2) */
                const int kr = __builtin_bit_cast(int, krf);
                if (i + 1 < PPT) q_fetch(i + 1);
                f32x4 l4;
#pragma unroll
                for (int k = 0; k < 4; k++) l4[k] = fin_log(e[i][k], rm, mpx);
                /* the only five posterior rows homopolymer_path reads (homopolymer.c:200,209): repeatblock(k, klen) and
                 * stay; kept here, stored after the loop (no branches inside it) */
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int s = k * ((NH - 1) / 3), sq = s >> 2;
                    if (i == 2 * (sq >> 6) + ((sq >> 5) & 1)) hpv[k] = l4[s & 3];
                }
                const float svi = sv[i];
                const unsigned cstep = SH_TB_STEP + (unsigned)sr[i], cskip = SH_TB_SKIP + (unsigned)kr;
                const unsigned cstart = SH_TB_START;
                unsigned codes = 0;                             /* four SH_TB_STAY */
                f32x4 ns;
#define SH_FVT_STATE(E)                                                                                         \
                {                                                                                               \
                    float s_ = pv[E] + stay_v;                  /* stay  :180 */                                \
                    const float st = l4[E] + svi;               /* step  :214-218 */                            \
                    SH_CODE_LT(E, codes, s_, st, cstep);                                                        \
                    s_ = __builtin_fmaxf(s_, st);                                                               \
                    const float sk = SKIP0 ? l4[E] + kv : (l4[E] + kv) - a.skip_pen;   /* skip  :256-262 */     \
                    SH_CODE_LT(E, codes, s_, sk, cskip);                                                        \
                    s_ = __builtin_fmaxf(s_, sk);                                                               \
                    const float fs = pstart + l4[E];            /* leave start :331-335 */                      \
                    SH_CODE_LT(E, codes, s_, fs, cstart);                                                       \
                    s_ = __builtin_fmaxf(s_, fs);                                                               \
                    ns[E] = s_;                                                                                 \
                }
                /* one addition per state for the three moves into it where the runner-up is clear of the maximum: see k_ff_viterbi */
                bool fast = false;
                float m = 0.f;
                unsigned cm = cstart;
                if (SKIP0) {
                    m = __builtin_fmaxf(__builtin_fmaxf(svi, kv), pstart);
                    const float md = __builtin_amdgcn_fmed3f(svi, kv, pstart);
                    const unsigned long long unclear = __builtin_amdgcn_fcmpf(m - md, (lbound + __builtin_fabsf(m)) * 4.76837158203125e-07f, 13 /* ULE */);
                    fast = (unclear & actmask) == 0;
                    cm = (kv == m) ? cskip : cm;
                    cm = (svi == m) ? cstep : cm;
                }
                if (fast) {
#define SH_FVT_FAST(E)                                                                                          \
                    {                                                                                           \
                        const float s_ = pv[E] + stay_v;        /* stay  :180 */                                \
                        const float mv = l4[E] + m;             /* the best move into the state */              \
                        SH_CODE_LT(E, codes, s_, mv, cm);                                                       \
                        ns[E] = __builtin_fmaxf(s_, mv);                                                        \
                    }
                    SH_FVT_FAST(0) SH_FVT_FAST(1) SH_FVT_FAST(2) SH_FVT_FAST(3)
#undef SH_FVT_FAST
                } else { SH_FVT_STATE(0) SH_FVT_STATE(1) SH_FVT_STATE(2) SH_FVT_STATE(3) }
#undef SH_FVT_STATE
                *(f32x4 *)(myq + SH_FVT_MT(i) * 256) = ns;          /* in place: nobody else reads these four */
                if (!(SH_FVT_ABL & 4)) (a.tb + (cb * NQ + QB) * 16)[tofs] = codes;   /* also for reads past their end (never read back): no branch */
                else asm volatile("" :: "v"(codes));
                {   /* next block's end-state scan, per quad: this thread meets its quads in increasing index order,
                     * so a strict compare keeps the first maximum */
                    const float ve = __builtin_fmaxf(__builtin_fmaxf(ns[0], ns[1]), __builtin_fmaxf(ns[2], ns[3])) - a.local_pen;
                    bi = (ve > bv) ? Q : bi;
                    bv = __builtin_fmaxf(bv, ve);
                }
                __builtin_amdgcn_sched_barrier(0);          /* quads one after the other */
#if SH_FVT_FLIP
                if (cw >= 4) { if (i == PPT - 1) __builtin_amdgcn_s_setprio(1); else if (i == SH_FVT_FLIP - 1) __builtin_amdgcn_s_setprio(0); }
#endif
            }
            if (active) pstart = nstart;
            if (a.hp_side && active) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int sq = (k * ((NH - 1) / 3)) >> 2;
                    if (cw == ((sq >> 2) & 7) && q == (sq & 3)) (a.hp_side + (hpo + t) * 5)[k] = hpv[k];
                }
            }
            {
                float ov = __shfl_xor(bv, 16); int oi = __shfl_xor(bi, 16);
                argmax_merge(bv, bi, ov, oi);
                ov = __shfl_xor(bv, 32); oi = __shfl_xor(bi, 32);
                argmax_merge(bv, bi, ov, oi);
                if (lane < 16) { redv[((par ^ 1) * NCW + wave) * 16 + b] = bv; redi[((par ^ 1) * NCW + wave) * 16 + b] = bi; }
            }
            VSTAMP(vC);
            lds_barrier();
            VSTAMP(vD);
        }
    }

    if (s1 < Tt) {
        /* the tile's later blocks run on another workgroup: leave it the state */
        if (decoder) {
            float *vst = a.vstate + (long long)tile * (NH * 16 + 32);
#pragma unroll
            for (int i = 0; i < PPT; i++) {
                const int Q = 4 * (SH_FVT_MT(i) + cw) + q;
                *(f32x4 *)(vst + (Q * 16 + b) * 4) = *(const f32x4 *)(myq + SH_FVT_MT(i) * 256);
            }
            if (tid < 16) { vst[NH * 16 + b] = pstart; vst[NH * 16 + 16 + b] = pend; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(a.flag + tile, (unsigned)ord + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        /* argmaxf over nh+2 final scores, first maximum wins (decode.c:68, util.c:9) */
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        if (decoder) {
#pragma unroll
            for (int i = 0; i < PPT; i++) {
                const int Q = 4 * (SH_FVT_MT(i) + cw) + q;
                const f32x4 pv = *(const f32x4 *)(myq + SH_FVT_MT(i) * 256);
#pragma unroll
                for (int k = 0; k < 4; k++) argmax_merge(bv, bi, pv[k], 4 * Q + k);
            }
            float ov = __shfl_xor(bv, 16); int oi = __shfl_xor(bi, 16);
            argmax_merge(bv, bi, ov, oi);
            ov = __shfl_xor(bv, 32); oi = __shfl_xor(bi, 32);
            argmax_merge(bv, bi, ov, oi);
        }
        __syncthreads();
        if (decoder && lane < 16) { redv[wave * 16 + b] = bv; redi[wave * 16 + b] = bi; }
        __syncthreads();
        if (tid < 16) {
            float ev = redv[b]; int ei = redi[b];
            for (int w = 1; w < NCW; w++) argmax_merge(ev, ei, redv[w * 16 + b], redi[w * 16 + b]);
            if (pstart > ev) { ev = pstart; ei = NH; }
            if (pend > ev) { ev = pend; ei = NH + 1; }
            a.final_state[rd] = ei;
            a.final_score[rd] = ev;
        }
        if (a.dump_final && a.vstate && decoder) {
            float *vst = a.vstate + (long long)tile * (NH * 16 + 32);
#pragma unroll
            for (int i = 0; i < PPT; i++) {
                const int Q = 4 * (SH_FVT_MT(i) + cw) + q;
                *(f32x4 *)(vst + (Q * 16 + b) * 4) = *(const f32x4 *)(myq + SH_FVT_MT(i) * 256);
            }
            if (tid < 16) { vst[NH * 16 + b] = pstart; vst[NH * 16 + 16 + b] = pend; }
        }
    }
    if (a.dbg && lane == 0) { unsigned long long *d = a.dbg + ((long long)blockIdx.x * 12 + wave) * 8; d[0] = vA; d[1] = vB; d[2] = vC; d[3] = vD; d[4] = (unsigned long long)(s1 - s0); }
#undef VSTAMP
#undef SH_FVT_MT
}

#endif /* SH_DECODE_TEAMS_H */
