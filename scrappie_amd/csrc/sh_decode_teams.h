/* sh_decode_teams.h -- part of sh_kernels.h (included from there, behind sh_decode.h): S1 + S2 + D1 forward in one kernel on TWO TEAMS of waves.
 * Device code for gfx950 only; see sh_kernels.h for conventions (layouts, split products, citations).
 *
 * k_ff_viterbi (sh_decode.h) lets each of its eight waves do everything: stream its rows of the S1 weights, multiply, exponentiate AND decode
 * -- 256 VGPRs a wave, two waves per SIMD, 28 % of wave time in s_waitcnt vmcnt on a weight stream only the products need
 * (profiles/r4_decoder_ablations.txt).  Here the work is divided the way k_gru_proj divides a recurrent layer:
 *
 *  * S1 TEAM, waves 8-11 (one per SIMD).  Producer p owns m-tiles 16p .. 16p+15 of the 1040 x 96 output layer (producer 3 also the stay
 *    state's tile).  The weights are what bounds S1 here: 394 KB of fp16 pieces per pass from L2, and a CU gets ~57 B/clk of that stream
 *    (tools/l2_stream_probe.hip) -- ~7 k cycles, more than the decoder team needs for a block (profiles/r5_decoder_teams_v1.txt: one pass
 *    per block 11.4 ms, the decoder team alone 6.9).  So ONE WEIGHT PASS SERVES TWO BLOCKS: the B operand of every m-tile is the trunk
 *    column of blocks u and u+1 (two independent MFMA chains on the same A), the stream is halved and runs SH_FVT_WBUF - 1 m-tiles ahead of
 *    the products through a ring of register buffers.  The products, the clamp + v_exp_f32 and the row sums are those of k_ff_viterbi,
 *    operation for operation (same association of the row sums: groups of SH_SUM_GROUP m-tiles), so every S1 form gives identical bits.
 *  * DECODER TEAM, waves 0-7 (two per SIMD): no MFMA, no weight stream, no vector-memory wait in the block loop (its only global traffic is
 *    the traceback store).  Scores are updated IN PLACE: one 64 KB score buffer instead of two.
 *  * Between them a 64 KB LDS ring of 64 slots ([lane][4] = the MFMA result image).  A pass is handed over in two halves of 32 m-tiles x 2
 *    blocks: the first half is written during the first block of a pair and taken into the decoders' registers in phase B of the second, the
 *    second half the other way round -- behind the barrier of a phase B the ring is the S1 team's again.  A decoder thread so holds the
 *    emissions of two blocks (64 VGPRs).
 *
 * What makes the in-place update possible is the division of the states.  Thread (wave w, q, read b) owns the eight quads
 *      Q = 64 rl + 32 c + 4 w + q,   rl < 4, c < 2        (a quad = the four one-base extensions of one 4-mer)
 * i.e. m-tiles 16 rl + 8 c + w (lane = 16 q + b: the MFMA result layout again; c = the half of producer rl's pass).  Then
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
 *    phase B   decoders: ring -> registers, step / skip maxima      producers: (first block of a pair) the next two trunk columns -> pieces; prime the stream
 *    phase C   decoders: S2 + update in place + traceback           producers: half a pass (8 m-tiles x 2 blocks) -> ring, row sums
 * LDS: scores 64 KB + ring 64 KB + skip maxima 8 KB + bias 4 KB + trunk pieces 12 KB + sums / end-state scan 4.6 KB = 157 KB.
 * 12 waves x <= 168 VGPRs, three waves per SIMD.  Not built for the slip move (k_ff_viterbi keeps that). */
#ifndef SH_DECODE_TEAMS_H
#define SH_DECODE_TEAMS_H

#ifndef SH_FVT_WBUF
#define SH_FVT_WBUF 3        /* register buffers of the S1 weight stream (24 VGPRs each): the stream runs SH_FVT_WBUF - 1 m-tiles ahead of the products */
#endif
#ifndef SH_FVT_PROD_PRIO
#define SH_FVT_PROD_PRIO 1   /* s_setprio of the S1 team: its waves are the youngest of their SIMDs (issue arbitration: priority, then age) and a late producer holds up the barrier */
#endif
#ifndef SH_FVT_ABL
#define SH_FVT_ABL 0         /* timing ablations (results invalid unless 0): 1 producers idle (the decoders keep the first pair's emissions), 2 no phase B scans, 4 no traceback store, 8 S2 (fma, v_log_f32, scale) on the S1 team instead of the decoders, with a made-up row sum, 16 no S2 anywhere */
#endif
#ifndef SH_FVT_FLIP
#define SH_FVT_FLIP 3        /* (0 -> 3: decode 10.21 -> 10.0 ms, profiles/r5_decoder_teams_v2.txt) */ /* n > 0: the younger decoder wave of a SIMD (waves 4-7) has priority for its first n quads of a block, the older one (by age) after that */
#endif
#ifndef SH_FVT_STAMP
#ifdef SH_EXPERIMENTS
#define SH_FVT_STAMP 1       /* per-wave cycle stamps of the phases (SH_VIT_STAMP=1): the experiments build only -- their accumulators are ten scalar registers */
#else
#define SH_FVT_STAMP 0
#endif
#endif
#define SH_FVT_NTH 768
#define SH_FVT_LDS_FLOATS (1024 * 16 + 64 * 256 + 2 * 64 * 16 + 2 * 2 * 8 * 16 + 4 * 9 * 16 + 65 * 16 + 2 * 3 * 512 + 3 * 2 * 4 * 4)

template <bool SKIP0, bool DIV, bool NOCLAMP = false>
__global__ __launch_bounds__(SH_FVT_NTH) void k_ff_viterbi_teams(ShFfArgs f, ShVitArgs a, ShMeta md) {
    static_assert(!(DIV && NOCLAMP), "with temperatures the clamp stays");
    constexpr int NCW = 8, NPW = 4, PPT = 8, NQ = 256, NH = 1024, KS = 3, KQ = 6, TPP = 16, HT = 8, NG = 9, WB = SH_FVT_WBUF;
    static_assert(SH_SUM_GROUP == HT, "half a pass of a producer is one row-sum group");
    static_assert(WB >= 2 && WB <= 4, "weight ring");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sc = smem;                                   /* scores, ONE buffer: quad Q of read b at (Q * 16 + b) * 4 */
    float *ring = sc + NH * 16;                         /* exp values on their way to the decoders: slot 16 p + 2 k + j = m-tile (16 p + 8 h + k) of block u + j, h = the half in flight */
    float *skk = ring + 64 * 256;                       /* skip maxima [suffix j < 64][read][value, prefix] */
    float *redv = skk + 2 * 64 * 16;                    /* end-state scan [block of the pair][NCW][16] */
    int *redi = (int *)(redv + 2 * NCW * 16);
    float *gsum = (float *)(redi + 2 * NCW * 16);       /* row-sum groups [block & 3][NG][16] (group 8 = the stay state's exp value) */
    float *sBias = gsum + 4 * NG * 16;                  /* bias x 2^14 by state row [65 * 16] */
    unsigned *xp = (unsigned *)(sBias + 65 * 16);       /* the trunk columns of the pair in the making, as pieces [2][KS][2][64][4] */
    unsigned *sStay = xp + 2 * KS * 512;                /* row 1024 of the weights as pieces [KS][2][4 k groups][4] */

    const int tid = threadIdx.x, lane = tid & 63, b = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool decoder = wave < NCW;
    const float mp = a.min_prob, mpm1 = 1.0f - a.min_prob;
    const float lbound = (mp > 0.0f) ? 1.0e-3f - __logf(mp) : INFINITY;      /* |log-posterior| <= this */
    unsigned long long vA = 0, vB = 0, vC = 0, vD = 0, vt0 = 0, vt1 = 0;
    (void)vt0; (void)vt1;
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
    int myT = md.rT[rd];
    /* waited for HERE, once: left to the compiler the wait sits at the value's first use inside the block loop, as s_waitcnt vmcnt(0) --
     * which there also covers every traceback store of the previous pair of blocks, on every pair */
    asm volatile("" : "+v"(myT));
    const long long hpo = a.hp_side ? a.hp_off[rd] : 0;
    float pstart = 0.0f, pend = -SH_BIG;

    /* the decoder team's quads: i = 2 rl + c, in increasing order of Q */
    const int cw = decoder ? wave : 0;
    float *myq = sc + cw * 256 + lane * 4;                       /* quad i of this thread: + (16 rl + 8 c) * 256 */
    const float *mye = ring + cw * 512 + lane * 4;               /* its emissions in the ring: slot 16 rl + 2 w + j */
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
            if (lane < 16) { redv[wave * 16 + b] = bv; redi[wave * 16 + b] = bi; }
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
        /* this wave's rows of the S1 weights: 96 KB of fp16 pieces per pass, from L2, through WB register buffers.  Global addresses as (ONE
         * running wave-uniform base in scalar registers, advanced by a tile per call -- the calls come in cyclic tile order) + (32-bit lane
         * offset): see k_ff_viterbi.  Tile k of a pass lands in buffer (k mod 8) mod WB: the stream is primed afresh for each half. */
        const unsigned *wmine = f.wpiece + (long long)(TPP * pw) * KS * 512;
        ShSplit W[WB][KS];
        const unsigned lofs = (unsigned)lane * 4u;
        typedef const __attribute__((address_space(1))) unsigned *gu32;
        typedef const __attribute__((address_space(1))) u32x4 *gu32x4;
        gu32 wp = (gu32)wmine;
        auto w_load = [&](int k) {
            static_assert(KS == 3, "two bases per tile: immediate offsets reach 4095 bytes");
            gu32 b0 = wp, b1 = wp + 1024;
            asm volatile("" : "+s"(b0), "+s"(b1));
            W[(k % HT) % WB][0].p1 = __builtin_bit_cast(f16x8, *(gu32x4)(b0 + lofs));
            W[(k % HT) % WB][0].p2 = __builtin_bit_cast(f16x8, *(gu32x4)(b0 + 256 + lofs));
            W[(k % HT) % WB][1].p1 = __builtin_bit_cast(f16x8, *(gu32x4)(b0 + 512 + lofs));
            W[(k % HT) % WB][1].p2 = __builtin_bit_cast(f16x8, *(gu32x4)(b0 + 768 + lofs));
            W[(k % HT) % WB][2].p1 = __builtin_bit_cast(f16x8, *(gu32x4)(b1 + lofs));
            W[(k % HT) % WB][2].p2 = __builtin_bit_cast(f16x8, *(gu32x4)(b1 + 256 + lofs));
            wp = (k == TPP - 1) ? (gu32)wmine : wp + KS * 512;
            asm volatile("" : "+s"(wp));
        };
        auto w_prime = [&](int k0) {
#pragma unroll
            for (int k = 0; k < WB - 1; k++) w_load(k0 + k);
        };
        /* the trunk columns of blocks u, u + 1, k step min(pw, 2), as raw fp32 (the loads are unconditional: straight-line vmcnt accounting) ... */
        const int xks = pw < KS ? pw : KS - 1;
        f32x4 xr[2][2];
        auto xraw_load = [&](int u) {
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const float *p = f.in + ((boff + min(u + j, s1 - 1)) * KQ + 2 * xks) * 256;       /* uniform */
                xr[j][0] = *(const f32x4 *)(p + lofs);
                xr[j][1] = *(const f32x4 *)(p + 256 + lofs);
            }
        };
        /* ... cut into pieces for the team */
        auto xp_publish = [&]() {
            if (pw < KS) {
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    f32x4 v0 = xr[j][0], v1 = xr[j][1];
                    if (DIV) { v0 = v0 / f.in_div; v1 = v1 / f.in_div; }      /* shift_scale_matrix_inplace: division (Q5); x / 1 = x */
                    const ShSplit sp = split8(v0, v1);
                    unsigned *d = xp + (j * KS + pw) * 512 + lane * 4;
                    *(u32x4 *)d = __builtin_bit_cast(u32x4, sp.p1);
                    *(u32x4 *)(d + 256) = __builtin_bit_cast(u32x4, sp.p2);
                }
            }
        };
        auto e_of = [&](float acc) { return DIV ? d_exp((acc * SH_OINV) / f.out_div) : NOCLAMP ? d_exp_acc_inrange(acc) : d_exp_acc(acc); };   /* no max subtraction (Q2) */
        auto group_out = [&](float part, int u, int g) {
            const float v = rows_sum(part);
            if (lane < 16) gsum[((u & 3) * NG + g) * 16 + b] = v;
        };
        /* half H of the pass for blocks u, u + 1: m-tiles 8 H .. 8 H + 7 of this producer -> 16 ring slots, one row-sum group per block;
         * producer 3, second half: the stay state's tile too */
        float *myring = ring + (TPP * pw) * 256 + lane * 4;
        auto s1_half = [&](auto Hc, int u) {
            constexpr int H = decltype(Hc)::value;
#if defined(SH_FVT_DBG_ROLE) && SH_FVT_DBG_ROLE == 1
            return;
#endif
            const float *mybias = sBias + (TPP * pw + HT * H) * 16 + 4 * q;
            ShSplit bp[2][KS];
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int ks = 0; ks < KS; ks++) bp[j][ks] = load_pieces(xp + (j * KS + ks) * 512, lane);
            float part0 = 0.0f, part1 = 0.0f;
            /* per m-tile eighteen products (two independent chains of nine on the same A), then clamp / exp / store / sum of both blocks */
            auto tile_out = [&](int k, const f32x4 &acc0, const f32x4 &acc1) {
                f32x4 ex0, ex1;
#pragma unroll
                for (int r = 0; r < 4; r++) { ex0[r] = e_of(acc0[r]); ex1[r] = e_of(acc1[r]); }
                part0 += (ex0[0] + ex0[1]) + (ex0[2] + ex0[3]);
                part1 += (ex1[0] + ex1[1]) + (ex1[2] + ex1[3]);
                if (SH_FVT_ABL & 8) {       /* timing ablation: S2's three instructions per value HERE, with a made-up row sum (VERDICT r5 item 1's best case) */
#pragma unroll
                    for (int r = 0; r < 4; r++) { ex0[r] = fin_log(ex0[r], mpm1, mp); ex1[r] = fin_log(ex1[r], mpm1, mp); }
                }
                *(f32x4 *)(myring + (2 * k) * 256) = ex0;
                *(f32x4 *)(myring + (2 * k + 1) * 256) = ex1;
            };
#pragma unroll
            for (int k = 0; k < HT; k++) {
                f32x4 acc0 = *(const f32x4 *)(mybias + k * 16), acc1 = acc0;
                if (k + WB - 1 < HT) w_load(HT * H + k + WB - 1);      /* into the buffer tile k - 1's products have read */
                const ShSplit (&A)[KS] = W[k % WB];
#pragma unroll
                for (int ks = 0; ks < KS; ks++) { acc0 = mfma16(A[ks].p1, bp[0][ks].p2, acc0); acc1 = mfma16(A[ks].p1, bp[1][ks].p2, acc1); }
#pragma unroll
                for (int ks = 0; ks < KS; ks++) { acc0 = mfma16(A[ks].p2, bp[0][ks].p1, acc0); acc1 = mfma16(A[ks].p2, bp[1][ks].p1, acc1); }
#pragma unroll
                for (int ks = 0; ks < KS; ks++) { acc0 = mfma16(A[ks].p1, bp[0][ks].p1, acc0); acc1 = mfma16(A[ks].p1, bp[1][ks].p1, acc1); }
                tile_out(k, acc0, acc1);
                __builtin_amdgcn_sched_barrier(0);
            }
            group_out(part0, u, 2 * pw + H);
            group_out(part1, u + 1, 2 * pw + H);
            if (H == 1 && pw == NPW - 1) {
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
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    f32x4 acc = *(const f32x4 *)(sBias + 64 * 16 + 4 * q);
                    acc = split_dot<KS>(Ws, bp[j], acc);
                    f32x4 ex;
#pragma unroll
                    for (int r = 0; r < 4; r++) ex[r] = (4 * q + r < 1) ? e_of(acc[r]) : 0.0f;        /* rows >= NS are padding */
                    group_out((ex[0] + ex[1]) + (ex[2] + ex[3]), u + j, NG - 1);
                }
            }
        };
        const std::integral_constant<int, 0> H0{};
        const std::integral_constant<int, 1> H1{};

        /* the pass for the piece's first pair, handed over with barriers of its own */
        const bool any = s1 > s0;
        if (any) xraw_load(s0);
        __syncthreads();                                    /* P0: bias, stay row, initial scores */
        if (any) { xp_publish(); w_prime(0); }
        lds_barrier();                                      /* P1 */
        if (any) s1_half(H0, s0);
        lds_barrier();                                      /* P2: the decoders take the first half */
        if (any) w_prime(HT);
        lds_barrier();                                      /* P3 */
        if (any) { s1_half(H1, s0); __builtin_amdgcn_sched_barrier(0); xraw_load(s0 + 2); }
        lds_barrier();                                      /* P4: the second half is in the ring */
        if (a.dbg) vt0 = __builtin_readcyclecounter();
        for (int t = s0; t < s1; t += 2) {
            const bool more = (t + 2 < s1) && !(SH_FVT_ABL & 1);      /* there is a pair t + 2, t + 3 to prepare */
            if (more) { xp_publish(); w_prime(0); }         /* (the pieces of the pair before were last read two phases ago) */
            VSTAMP(vA);
            lds_barrier();
            VSTAMP(vB);
            if (more) s1_half(H0, t + 2);
            VSTAMP(vC);
            lds_barrier();
            VSTAMP(vD);
            if (more) w_prime(HT);
            VSTAMP(vA);
            lds_barrier();
            VSTAMP(vB);
            if (more) s1_half(H1, t + 2);
            __builtin_amdgcn_sched_barrier(0);
            xraw_load(t + 4);                               /* (unconditional: xr is redefined every iteration, i.e. not live through the pass; in flight only where the weight ring has drained) */
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
        /* the emissions of the pair being decoded: [block of the pair][quad].  At the top of a pair the even quads are in place; phase B of the
         * first block brings the odd ones; phase B of the second block brings the NEXT pair's even quads, into the first block's registers (dead
         * by then: block u + 2 -> its even slots, block u + 3 -> its odd slots, moved over to the second block's even slots behind its phase C) */
        f32x4 eN[2][PPT];
        /* half a pass out of the ring: slot 16 rl + 2 w + j -> (rl, block j) */
        auto drain = [&](auto a0, auto s0c, auto a1, auto s1c) {
            constexpr int A0 = decltype(a0)::value, S0 = decltype(s0c)::value, A1 = decltype(a1)::value, S1 = decltype(s1c)::value;
#pragma unroll
            for (int rl = 0; rl < PPT / 2; rl++) { eN[A0][2 * rl + S0] = *(const f32x4 *)(mye + rl * 4096); eN[A1][2 * rl + S1] = *(const f32x4 *)(mye + rl * 4096 + 256); }
        };
        const std::integral_constant<int, 0> I0{};
        const std::integral_constant<int, 1> I1{};
        __syncthreads();                                    /* P0 */
        lds_barrier();                                      /* P1 */
        lds_barrier();                                      /* P2 */
        drain(I0, I0, I1, I0);
        lds_barrier();                                      /* P3 */
        lds_barrier();                                      /* P4 */
        if (a.dbg) vt0 = __builtin_readcyclecounter();
        const unsigned tofs = (unsigned)lane;
        /* one block.  FIRST: the first block of its pair -- phase B takes the pair's second half (odd quads) out of the ring; else the first half of the next pair */
        auto block = [&](const int t, auto first_c) {
            constexpr bool FIRST = decltype(first_c)::value;
            const f32x4 (&e)[PPT] = eN[FIRST ? 0 : 1];
            const long long cb = boff + t;
            constexpr int par = FIRST ? 0 : 1;              /* end-state scan slots by position in the pair (compile-time addresses; the piece starts with a first block) */
#if defined(SH_FVT_DBG_ROLE) && SH_FVT_DBG_ROLE == 2
            lds_barrier(); lds_barrier(); return;
#endif
            if (!FIRST && t >= s1) {                        /* a piece of an odd number of blocks */
                drain(I0, I0, I0, I1);                      /* (redefined every pair whatever happens: not live through the first block) */
                lds_barrier(); lds_barrier();
                return;
            }

            /* ---- phase B: step maxima of my quads; skip maxima of my two suffixes; emissions out of the ring ---- */
            float sv[PPT];
            int sr[PPT];                                    /* (phase B only: across the barrier the prefixes travel two bits each in srp) */
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
                if (i == PPT / 2 - 1) __builtin_amdgcn_sched_barrier(0);      /* (register budget: the scans' loads four quads at a time) */
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
            unsigned srp = 0;
#pragma unroll
            for (int i = 0; i < PPT; i++) srp |= (unsigned)sr[i] << (2 * i);
            /* the end state's predecessor (decode.c:343-348) reads a quad of another thread: resolved here, before anything is overwritten.  ei is the
             * first QUAD that holds the maximum of (score - local_pen); the state is the first of its four that attains it */
            float ev = 0.f; int tbe_in = 0;
            if (cw == 0) {
                int ei;
                /* (one address register + immediate offsets; an LDS pointer by type: as a generic one these were sixteen flat loads behind
                 * s_waitcnt vmcnt(0) lgkmcnt(0), i.e. behind the wave's traceback stores) */
                typedef __attribute__((address_space(3))) const float *ldsf;
                ldsf rv = (ldsf)(redv + b);
                asm volatile("" : "+v"(rv));
                ev = rv[par * NCW * 16];
                ei = __builtin_bit_cast(int, rv[(2 + par) * NCW * 16]);
#pragma unroll
                for (int w = 1; w < NCW; w++) {
                    const float ov = rv[(par * NCW + w) * 16], oif = rv[((2 + par) * NCW + w) * 16];
                    argmax_merge(ev, ei, ov, __builtin_bit_cast(int, oif));
                }
                const f32x4 q4 = *(const f32x4 *)(sc + ((ei & (NQ - 1)) * 16 + b) * 4);
                int e0 = 3;
                e0 = (q4[2] - a.local_pen == ev) ? 2 : e0;
                e0 = (q4[1] - a.local_pen == ev) ? 1 : e0;
                e0 = (q4[0] - a.local_pen == ev) ? 0 : e0;
                tbe_in = 4 * ei + e0;
            }
            /* half a pass out of the ring, behind the scans (register budget): the second half of this pair / the first half of the next */
            __builtin_amdgcn_sched_barrier(0);
            if (FIRST) drain(I0, I1, I1, I1);
            else drain(I0, I0, I0, I1);
            VSTAMP(vA);
            lds_barrier();
            VSTAMP(vB);

            /* ---- phase C: S2 + update of my states, in place ---- */
            float tot = 0.0f;
#pragma unroll
            for (int w = 0; w < NG; w++) tot += gsum[((t & 3) * NG + w) * 16 + b];
            const float rmf = d_rcp(tot) * mpm1;                        /* fin_log's factor; v_rcp_f32 in every consumer of the row sum: the forms keep identical bits */
            const float stay_lp = fin_log(gsum[((t & 3) * NG + NG - 1) * 16 + b], rmf, mp);
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
                if (active && tid < 16) a.tb_end[cb * 16 + b] = enter_end ? tbe_in : NH + 1;
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
                for (int k = 0; k < 4; k++) l4[k] = (SH_FVT_ABL & 24) ? e[i][k] : fin_log(e[i][k], rm, mpx);
                /* the only five posterior rows homopolymer_path reads (homopolymer.c:200,209): repeatblock(k, klen) and
                 * stay; kept here, stored after the loop (no branches inside it) */
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int s = k * ((NH - 1) / 3), sq = s >> 2;
                    if (i == 2 * (sq >> 6) + ((sq >> 5) & 1)) hpv[k] = l4[s & 3];
                }
                const float svi = sv[i];
                const unsigned cstep = SH_TB_STEP + ((srp >> (2 * i)) & 3u), cskip = SH_TB_SKIP + (unsigned)kr;
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
                    /* SH_FVT_GAP (sh_decode.h): 2^-21 (|l|max + |m|).  (One threshold per read and block from |pstart| >= |m| instead -- two instructions fewer
                     * per quad -- was measured SLOWER, 10.0 against 9.7 ms: profiles/r6_decoder_issue.txt) */
                    const unsigned long long unclear = __builtin_amdgcn_fcmpf(m - md, (lbound + __builtin_fabsf(m)) * SH_FVT_GAP, 13 /* ULE */);
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
                    const float ve = d_max4(ns) - a.local_pen;
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
                rows_argmax(bv, bi);
                if (lane < 16) { redv[((par ^ 1) * NCW + wave) * 16 + b] = bv; redi[((par ^ 1) * NCW + wave) * 16 + b] = bi; }
            }
            VSTAMP(vC);
            lds_barrier();
            VSTAMP(vD);
        };
        for (int t = s0; t < s1; t += 2) {
            block(t, std::true_type{});
            block(t + 1, std::false_type{});
#pragma unroll
            for (int rl = 0; rl < PPT / 2; rl++) eN[1][2 * rl] = eN[0][2 * rl + 1];
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
