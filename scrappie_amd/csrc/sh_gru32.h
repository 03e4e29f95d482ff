/* sh_gru32.h -- part of sh_kernels.h: L1 + G1/G2 (+R1) of one recurrent layer on tiles of 32 reads (two adjacent 16-read
 * tiles of the launch group), v_mfma_f32_32x32x16_f16 split products, eight specialised waves per workgroup.
 * layers.c:373-527 (gru_forward / gru_backward / gru_step), scrappie_matrix.c:323 (feedforward_linear), layers.c:303
 * (residual).  Device code for gfx950 only; conventions in sh_kernels.h.
 *
 * WHY.  k_gru_proj (sh_gru.h) steps two 16-read tiles on 12 waves, three to a SIMD: every wave reads the whole B
 * operand (h, r*h or the input column as fp16 pieces) from LDS for 16 units' worth of MFMAs, all twelve are in the
 * same kind of phase between two barriers, and the SIMDs that host two recurrence waves set the pace (DESIGN.md
 * section 5: 5430 cycles per double step for 2592 of matrix pipe).  On a 32 x 32 tile a wave covers 32 units per B
 * operand read (half the LDS traffic per read), the serial chain of a step -- reset gate -> r*h -> candidate -> blend
 * -- belongs to ONE wave per SIMD, and everything that is not on that chain runs on other waves' instruction
 * streams:
 *
 *   wave 0-2  R_j   (SIMD j)   the chain for units 32j..32j+31: reset-gate recurrence product, logistic, r*h -> pieces;
 *                              candidate recurrence product, logistic(z), tanh, blend, h -> HBM, h -> pieces
 *   wave 4-6  G_j   (SIMD j)   gates z and r of the same units off the chain: projection rows of iW (block t+1) and the
 *                              update gate's recurrence product sW_z . h(t), which needs nothing but h
 *   wave 3    C     (SIMD 3)   the candidate's projection rows of iW for all 96 units
 *   wave 7    L     (SIMD 3)   fetches the layer's input column two blocks ahead and cuts it into pieces
 *
 * Matrix pipe per step and SIMD: R 36 + G 54 MFMAs of 32 cycles on SIMDs 0-2, C 54 on SIMD 3 (the 16-read form: 81 of 16
 * cycles per tile and SIMD, on average).  Two LDS-only barriers per step as before; the gate inputs still never
 * exist in HBM, and since every hand-over slot is written in one interval and read in the other they need no ring:
 * 84 KB of LDS per workgroup.
 *
 * DATA.  The HBM layout is unchanged (chunks of 16 units x 16 reads, sh_kernels.h).  The accumulator of
 * v_mfma_f32_32x32x16 holds, in lane (n = l & 31, hf = l >> 5) and register r, unit 32 j + 8 (r >> 2) + 4 hf + (r & 3) of
 * read n: a group of four registers is one 16-byte vector of chunk 2 j + (r >> 3) of the read's own 16-read tile
 * (n >> 4), so loads and stores stay whole vectors (two 512-byte runs per wave instruction), and a lane's 16 values
 * are the B operand of k steps 2 j, 2 j + 1 (16 units each) for the same lane: cutting into pieces is lane-local, one
 * ds_write_b128 per piece and k step.  Weights are cut on the host in the matching k order (make_piece_frags32).
 *
 * ARITHMETIC.  As everywhere (sh_kernels.h): fp32 operands as two fp16 pieces, three products, one fp32 accumulator
 * in 2^14 units.  Differences to the 16-read kernels, all inside the stated fp32 tolerance: the products of a k step
 * are issued together (a1 b2, a2 b1, a1 b1 per 16-wide k step instead of all cross terms first: the B pieces stream
 * through 8 registers instead of living in 48), and the matrix instruction sums 16 instead of 32 products at a time.
 * Both change the last bits, so a model runs ALL its tiles through one form (a read's call must not depend on its
 * batch).  Columns of an MFMA are independent: which tile a read's tile is paired with changes nothing.
 */
#ifndef SH_GRU32_H
#define SH_GRU32_H

typedef float f32x16 __attribute__((ext_vector_type(16)));


#ifndef SH_G32_KA
#define SH_G32_KA 4          /* k steps (of 6) of the z / r projection a G wave issues in interval A, behind the update gate's recurrence product */
#endif
#ifndef SH_G32_KC
#define SH_G32_KC 3          /* k steps of the candidate projection wave C issues in interval B (the rest in the next interval A) */
#endif
#ifndef SH_G32_D
#define SH_G32_D 3           /* B operands (k steps) a G or C wave reads ahead of its MFMAs */
#endif
#ifndef SH_G32_ZG
#define SH_G32_ZG 0          /* 1: the G waves apply the update gate's logistic themselves (under the chain waves' reset-gate products) and hand z over; 0: the chain waves do */
#endif
#ifndef SH_G32_ZC
#define SH_G32_ZC 0          /* 1: the update gate's logistic is applied in interval B by the waves of SIMD 3 (C: units of R_0 and R_1, L: of R_2), which are idle there, in place in
                                the ring slot, and handed to the chain waves through an LDS flag per chain wave (it is off the chain until the blend); 0: by the chain waves.
                                Same bits.  MEASURED SLOWER: five layers 16.0 against 15.25 ms (profiles/r4_gru32_offload.txt) although leaving the logistic out altogether
                                (SH_G32_ABL=16) gains 14 %: the chain waves' z / tanh / blend segment does shrink (1264 -> ~900 cycles), but their candidate products and output
                                stores stretch by as much while C and L work (2700 / 3050 instead of 1250 / 1500 cycles of interval B).  On a wave of the chain wave's own SIMD
                                (SH_G32_ZG) it costs interval A what it saves interval B (15.8-16.0 against 15.5, at any SH_G32_KA). */
#endif
#ifndef SH_G32_MIXCUT
#define SH_G32_MIXCUT 1      /* 1: cut into pieces with v_fma_mixlo/hi_f16 (4 instructions per pair, the same bits); 0: split_pair (8) */
#endif
#ifndef SH_G32_BLEND2
#define SH_G32_BLEND2 1      /* 1: h' = hbar + z (h - hbar) with hbar = fma(2, y, -1) (3 operations behind the reciprocal); 0: the reference's z h + (1 - z) hbar (6) */
#endif
#ifndef SH_G32_LATE_X
#define SH_G32_LATE_X 0      /* 1: a chain wave's products start from zero on its own pieces (registers) the moment the interval begins, and the gate
                                input the G / C waves left in LDS is added behind them; 0: the gate input is the accumulator's start (an LDS round trip in front of the chain) */
#endif
#ifndef SH_G32_ABL
#define SH_G32_ABL 0         /* timing ablations (results invalid unless 0): 1 G and C issue no MFMAs, 2 no transcendentals in the chain waves, 4 the chain waves issue no MFMAs, 8 L cuts nothing,
                                16 no logistic of the update gate in the chain waves, 32 the chain waves store no output (what handing either to an idle wave could gain at most) */
#endif
#ifndef SH_G32_RPRIO
#define SH_G32_RPRIO 2       /* s_setprio of the chain waves (the others stay at 0) */
#endif

__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
/* the three products of one 16-wide k step on one accumulator (cross terms first within the step) */
__device__ __forceinline__ f32x16 split_k32(const ShSplit &a, const ShSplit &b, f32x16 c) {
    c = mfma32(a.p1, b.p2, c);
    c = mfma32(a.p2, b.p1, c);
    return mfma32(a.p1, b.p1, c);
}
__device__ __forceinline__ f32x4 g32_logistic(f32x4 a) { return (SH_G32_ABL & 2) ? a * (0.25f * SH_OINV) + 0.5f : d_logistic4_acc(a); }
template <int G>
__device__ __forceinline__ f32x4 grp16(const f32x16 &a) { return __builtin_shufflevector(a, a, 4 * G, 4 * G + 1, 4 * G + 2, 4 * G + 3); }

/* split_pair (sh_kernels.h) in four instructions: p1 = f16(64 x) and p2 = f16(64 x - p1) each straight out of one fused
 * multiply-add rounded once to fp16 -- 64 x and 64 x - p1 are exact in fp32, so these are the same bits */
__device__ __forceinline__ void split_pair32(float x, float y, unsigned &w1, unsigned &w2) {
#if SH_G32_MIXCUT
    const float sc = SH_ASCALE;
    asm("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
        "v_fma_mixhi_f16 %0, %3, %4, 0\n\t"
        "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(w1), "=&v"(w2) : "v"(x), "v"(y), "s"(sc));
#else
    split_pair(x, y, w1, w2);
#endif
}
/* 16 values of a lane (four groups of four consecutive units) -> the pieces of k steps 2 j and 2 j + 1 */
__device__ __forceinline__ void cut16(const f32x4 (&v)[4], u32x4 (&p1)[2], u32x4 (&p2)[2]) {
#pragma unroll
    for (int s = 0; s < 2; s++) {
        unsigned a1, a2, b1, b2, c1, c2, d1, d2;
        split_pair32(v[2 * s][0], v[2 * s][1], a1, a2);
        split_pair32(v[2 * s][2], v[2 * s][3], b1, b2);
        split_pair32(v[2 * s + 1][0], v[2 * s + 1][1], c1, c2);
        split_pair32(v[2 * s + 1][2], v[2 * s + 1][3], d1, d2);
        p1[s] = (u32x4){a1, b1, c1, d1};
        p2[s] = (u32x4){a2, b2, c2, d2};
    }
}
__device__ __forceinline__ void put16(unsigned *buf, int j, int lane, const u32x4 (&p1)[2], const u32x4 (&p2)[2]) {
#pragma unroll
    for (int s = 0; s < 2; s++) {
        *(u32x4 *)(buf + ((2 * j + s) * 2 + 0) * 256 + lane * 4) = p1[s];
        *(u32x4 *)(buf + ((2 * j + s) * 2 + 1) * 256 + lane * 4) = p2[s];
    }
}
__device__ __forceinline__ void publish16(unsigned *buf, int j, int lane, const f32x4 (&v)[4]) {
    u32x4 p1[2], p2[2];
    cut16(v, p1, p2);
    put16(buf, j, lane, p1, p2);
}
/* an accumulator image in LDS: [group of four registers][64 lanes][4] */
__device__ __forceinline__ f32x16 acc_read(const float *p, int lane) {
    const f32x4 a = *(const f32x4 *)(p + lane * 4), b = *(const f32x4 *)(p + 256 + lane * 4);
    const f32x4 c = *(const f32x4 *)(p + 512 + lane * 4), d = *(const f32x4 *)(p + 768 + lane * 4);
    return (f32x16){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], c[0], c[1], c[2], c[3], d[0], d[1], d[2], d[3]};
}
__device__ __forceinline__ void acc_write(float *p, int lane, const f32x16 &a) {
    *(f32x4 *)(p + lane * 4) = grp16<0>(a);
    *(f32x4 *)(p + 256 + lane * 4) = grp16<1>(a);
    *(f32x4 *)(p + 512 + lane * 4) = grp16<2>(a);
    *(f32x4 *)(p + 768 + lane * 4) = grp16<3>(a);
}
/* a bias in accumulator units: [m-tile][hf][16] in LDS (the same for every read of the tile) */
__device__ __forceinline__ f32x16 bias_read(const float *tab, int mt, int lane) {
    const float *p = tab + (mt * 2 + (lane >> 5)) * 16;
    const f32x4 a = *(const f32x4 *)p, b = *(const f32x4 *)(p + 4), c = *(const f32x4 *)(p + 8), d = *(const f32x4 *)(p + 12);
    return (f32x16){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], c[0], c[1], c[2], c[3], d[0], d[1], d[2], d[3]};
}

/* global access as (uniform 64-bit base in scalar registers) + (32-bit lane offset in bytes) + constant */
typedef __attribute__((address_space(1))) char *sh_gchar;
typedef __attribute__((address_space(1))) f32x4 *sh_gf32x4;
__device__ __forceinline__ sh_gchar sh_uniform_ptr(const void *p) {       /* the caller's pointer is wave-uniform: say so */
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return (sh_gchar)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ f32x4 gload_so(const float *base, unsigned voff, int imm) {
    sh_gchar b = sh_uniform_ptr(base);
    asm volatile("" : "+s"(b));
    return *(sh_gf32x4)(b + (unsigned long long)voff + imm);
}
__device__ __forceinline__ void gstore_so(float *base, unsigned voff, int imm, f32x4 v) {
    sh_gchar b = sh_uniform_ptr(base);
    asm volatile("" : "+s"(b));
    *(sh_gf32x4)(b + (unsigned long long)voff + imm) = v;
}

/* a lane's walk over its segments (steps of pairs); wave-uniform part */
struct ShPairCursor {
    int sgi, sge;
    int pair, s, s1, Tt;
    long long boff0;
    bool ok;
};

#define SH_G32_COLB 6144          /* bytes per column block of 96 units x 16 reads */
#define SH_G32_LDS_WORDS (4 * 3072 + 9 * 1024 + 288 + 4)

template <bool RESID, bool STAMP>
__global__ __launch_bounds__(512) void k_gru_proj32(const float *__restrict__ in, float *__restrict__ out,
                                                    const unsigned *__restrict__ iWp /* [9][6][2][256] */, const float *__restrict__ ibias /* [9][2][16] x 2^14 */,
                                                    const unsigned *__restrict__ sWp /* [6][6][2][256] */, const unsigned *__restrict__ sW2p /* [3][6][2][256] */,
                                                    ShMeta md, int backward, ShGruPairs L, unsigned long long *dbg) {
    extern __shared__ __attribute__((aligned(16))) unsigned ldsw[];
    constexpr int PB = 3072;                       /* one operand as pieces: [6 k steps][2 pieces][64 lanes][4 words] */
    unsigned *const H = ldsw, *const RH = ldsw + PB;
    auto IN = [&](int par) { return ldsw + (2 + par) * PB; };
    float *const RING = (float *)(ldsw + 4 * PB);   /* [z | r | candidate][j][accumulator image of 1024 floats] */
    auto ring = [&](int gate, int j) { return RING + (gate * 3 + j) * 1024; };
    float *const BIAS = RING + 9 * 1024;
    unsigned *const ZF = (unsigned *)(BIAS + 288);  /* ZF[j] = steps whose update gate (after the logistic) is in ring(0, j) */
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    unsigned long long sa = 0, sb = 0, sc = 0, sd = 0, st0 = 0, st1;
    unsigned long long q[8] = {0, 0, 0, 0, 0, 0, 0, 0}, qt0 = 0, qt1;     /* finer marks inside the chain waves' intervals (STAMP builds) */
#define QMARK(i) do { if (STAMP) { __builtin_amdgcn_sched_barrier(0); qt1 = __builtin_readcyclecounter(); q[i] += qt1 - qt0; qt0 = qt1; __builtin_amdgcn_sched_barrier(0); } } while (0)
#define GSTAMP(acc) do { if (STAMP) { st1 = __builtin_readcyclecounter(); acc += st1 - st0; st0 = st1; } } while (0)

    const int sg0 = __builtin_amdgcn_readfirstlane(L.lane_off[blockIdx.x]);
    const int sg1 = __builtin_amdgcn_readfirstlane(L.lane_off[blockIdx.x + 1]);
    int nit = 0;
    for (int i = sg0; i < sg1; i++) nit += L.seg[i].s1 - L.seg[i].s0;
    nit = __builtin_amdgcn_readfirstlane(nit);
    if (nit == 0) return;
    if (threadIdx.x < 288) BIAS[threadIdx.x] = ibias[threadIdx.x];
    if (threadIdx.x < 4) ZF[threadIdx.x] = 0u;

    /* per-lane view of the current pair: blocks of the lane's own tile (hT), of its read (myT), byte offset of the
     * tile's column 0 from the pair's first tile (+ the lane's vector inside a chunk) */
    const int half = (lane >> 4) & 1;
    const unsigned lanepart = (unsigned)(((lane >> 5) * 16 + (lane & 15)) * 16);
    int hT = 0, myT = 0;
    unsigned voff = 0;
    ShPairCursor c;
    c.sgi = sg0; c.sge = sg1; c.ok = false; c.pair = 0; c.s = 0; c.s1 = 0; c.Tt = 0; c.boff0 = 0;
    auto enter = [&]() {
        c.ok = c.sgi < c.sge;
        if (c.ok) {
            const ShGruSegD sg = L.seg[c.sgi];
            c.pair = __builtin_amdgcn_readfirstlane(sg.tile);
            c.s = __builtin_amdgcn_readfirstlane(sg.s0);
            c.s1 = __builtin_amdgcn_readfirstlane(sg.s1);
            const int tA = __builtin_amdgcn_readfirstlane(L.pair_tile[2 * c.pair]);
            const int tB = __builtin_amdgcn_readfirstlane(L.pair_tile[2 * c.pair + 1]);
            const int T0 = __builtin_amdgcn_readfirstlane(md.tile_T[tA]);
            const int T1 = tB >= 0 ? __builtin_amdgcn_readfirstlane(md.tile_T[tB]) : 0;
            c.Tt = max(T0, T1);
            {   /* (wave-uniform, and told so: left in vector registers every address below becomes 64-bit VALU arithmetic) */
                const unsigned long long b0 = (unsigned long long)md.tile_boff[tA];
                c.boff0 = (long long)(((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(b0 >> 32)) << 32) | __builtin_amdgcn_readfirstlane((unsigned)b0));
            }
            long long boff1 = c.boff0;
            if (tB >= 0) {
                const unsigned long long b1 = (unsigned long long)md.tile_boff[tB];
                boff1 = (long long)(((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(b1 >> 32)) << 32) | __builtin_amdgcn_readfirstlane((unsigned)b1));
            }
            const bool second = half && T1 > 0;
            hT = half ? T1 : T0;
            myT = (half ? tB >= 0 : true) ? md.rT[(half ? tB : tA) * 16 + (lane & 15)] : 0;
            voff = lanepart + (second ? (unsigned)(boff1 - c.boff0) * (unsigned)SH_G32_COLB : 0u);
        }
    };
    /* byte offset of the lane's vector of block t of its tile from the pair's first column; a block the tile does not
     * have is replaced by one it has (or by the first tile's) -- such lanes are inactive, whatever they read */
    auto block_off = [&](int t) {
        const int lim = hT > 0 ? hT - 1 : 0;
        return (unsigned)min(t, lim) * (unsigned)SH_G32_COLB + voff;
    };

    /* SH_G32_ZC: logistic of the update gate G_j left in ring(0, j), in place, then the flag (a wave's LDS operations execute in order) */
    auto z_in_place = [&](const int j, const int it) {
        float *zp = ring(0, j);
        f32x4 z0 = *(const f32x4 *)(zp + lane * 4), z1 = *(const f32x4 *)(zp + 256 + lane * 4);
        f32x4 z2 = *(const f32x4 *)(zp + 512 + lane * 4), z3 = *(const f32x4 *)(zp + 768 + lane * 4);
        z0 = g32_logistic(z0); z1 = g32_logistic(z1); z2 = g32_logistic(z2); z3 = g32_logistic(z3);
        *(f32x4 *)(zp + lane * 4) = z0; *(f32x4 *)(zp + 256 + lane * 4) = z1;
        *(f32x4 *)(zp + 512 + lane * 4) = z2; *(f32x4 *)(zp + 768 + lane * 4) = z3;
        asm volatile("" ::: "memory");
        if (lane == 0) *(volatile unsigned *)(ZF + j) = (unsigned)(it + 1);
        asm volatile("" ::: "memory");
    };
    if (wave < 3) {
        /* ------------------------------ R_j: the chain ------------------------------ */
        const int j = wave;
        if (SH_G32_RPRIO) __builtin_amdgcn_s_setprio(SH_G32_RPRIO);
        /* k steps in the order own first: local step i is k step (2 j + i) % 6, so that the two k steps this wave cuts itself come
         * straight from its registers and its products start before the other waves' pieces have arrived from LDS */
        ShSplit wr[6], wc[6];
        int kofs[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const int ks = (2 * j + i) % 6;
            kofs[i] = ks * 512;
            wr[i] = load_pieces(sWp + ((3 + j) * 6 + ks) * 512, lane);
            wc[i] = load_pieces(sW2p + (j * 6 + ks) * 512, lane);
        }
#pragma unroll
        for (int ks = 0; ks < 6; ks++) asm volatile("" : "+v"(wr[ks].p1), "+v"(wr[ks].p2), "+v"(wc[ks].p1), "+v"(wc[ks].p2));
        f32x4 h[4];
        ShSplit own[2];                                    /* the pieces this wave published last (h, then r*h, then h ...) */
        auto publish_own = [&](unsigned *buf, const f32x4 (&v)[4]) {
            u32x4 p1[2], p2[2];
            cut16(v, p1, p2);
            put16(buf, j, lane, p1, p2);
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) { own[s2].p1 = __builtin_bit_cast(f16x8, p1[s2]); own[s2].p2 = __builtin_bit_cast(f16x8, p2[s2]); }
        };
        auto take_over = [&]() {
#pragma unroll
            for (int g = 0; g < 4; g++) h[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (c.ok && c.s > 0) {                         /* the pair's first steps ran on another lane */
                if (!sh_wait_flag(L.flag + c.pair, 3u, L.err) && lane == 0)
                    __hip_atomic_store(L.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                const float *hs = L.hstate + ((long long)c.pair * 3 + j) * 1024 + lane * 4;
#pragma unroll
                for (int g = 0; g < 4; g++)
#pragma unroll
                    for (int k = 0; k < 4; k++) h[g][k] = __hip_atomic_load(hs + g * 256 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            /* consumed here, so that the step loop never waits for these (rare) loads where the paths join */
            asm volatile("" : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(hT), "+v"(myT), "+v"(voff));
        };
        enter();
        take_over();
        publish_own(H, h);
        lds_barrier();
        lds_barrier();
        if (STAMP) qt0 = st0 = __builtin_readcyclecounter();
        for (int it = 0; it < nit; it++) {
            const int t = backward ? c.Tt - 1 - c.s : c.s;
            f32x4 rs[4];
            if (RESID) {                                   /* networks.c:583: the layer's input column is added to its output */
                const float *rb = in + c.boff0 * 1536 + j * 512;        /* (this wave's two chunks: the constant part of the address stays a constant) */
                const unsigned o = block_off(t);
#pragma unroll
                for (int g = 0; g < 4; g++) rs[g] = gload_so(rb, o, (g >> 1) * 1024 + (g & 1) * 512);
            }
            /* interval A: reset gate (layers.c:511-515) */
            /* (all LDS reads of an interval are issued before its first MFMA: left alone the compiler gives every
             * piece the same four registers and the wave pays an LDS round trip per product) */
            f32x16 acc = acc_read(ring(1, j), lane);
            f32x16 xin = acc;
            if (SH_G32_LATE_X) acc = (f32x16){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            {
                ShSplit hp[6];
                hp[0] = own[0]; hp[1] = own[1];
#pragma unroll
                for (int i = 2; i < 6; i++) hp[i] = load_pieces(H + kofs[i], lane);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < 6; ks++) { if (SH_G32_ABL & 4) acc[ks] += (float)hp[ks].p1[0]; else acc = split_k32(wr[ks], hp[ks], acc); }
            }
            if (SH_G32_LATE_X) acc += xin;
            if (STAMP) asm volatile("" : "+v"(acc));
            QMARK(0);                                      /* LDS reads + reset-gate products complete */
            {
                f32x4 rh[4];
                rh[0] = g32_logistic(grp16<0>(acc)) * h[0];
                rh[1] = g32_logistic(grp16<1>(acc)) * h[1];
                rh[2] = g32_logistic(grp16<2>(acc)) * h[2];
                rh[3] = g32_logistic(grp16<3>(acc)) * h[3];
                if (STAMP) asm volatile("" : "+v"(rh[0]), "+v"(rh[1]), "+v"(rh[2]), "+v"(rh[3]));
                QMARK(1);                                  /* logistic(r) * h */
                publish_own(RH, rh);
                QMARK(2);                                  /* cut + LDS writes issued */
            }
            GSTAMP(sa);
            lds_barrier();
            GSTAMP(sb);
            if (STAMP) qt0 = __builtin_readcyclecounter();
            /* interval B: candidate on r*h (layers.c:517-521), update gate as G_j left it, blend (layers.c:525) */
            acc = acc_read(ring(2, j), lane);
            xin = acc;
            if (SH_G32_LATE_X) acc = (f32x16){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            f32x16 za;
            {
                ShSplit rp[6];
                rp[0] = own[0]; rp[1] = own[1];
#pragma unroll
                for (int i = 2; i < 6; i++) rp[i] = load_pieces(RH + kofs[i], lane);
                if (!SH_G32_ZC) za = acc_read(ring(0, j), lane);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < 6; ks++) { if (SH_G32_ABL & 4) acc[ks] += (float)rp[ks].p1[0]; else acc = split_k32(wc[ks], rp[ks], acc); }
            }
            unsigned zf = 0;
            if (SH_G32_ZC) {       /* flag first, then the gate: if the flag read saw this step's value, so did the reads behind it; both travel while the products run */
                zf = *(volatile unsigned *)(ZF + j);
                asm volatile("" ::: "memory");
                za = acc_read(ring(0, j), lane);
            }
            if (SH_G32_LATE_X) acc += xin;
            if (STAMP) asm volatile("" : "+v"(acc));
            QMARK(3);                                      /* LDS reads + candidate products complete */
            const bool active = t < myT;
#define SH_G32_HBAR(G, HB)                                                                  \
            {                                                                              \
                f32x4 y = grp16<G>(acc) * (-2.0f * 1.44269504088896341f * SH_OINV);        \
                _Pragma("unroll") for (int k = 0; k < 4; k++) y[k] = (SH_G32_ABL & 2) ? y[k] * 0.25f + 0.5f : d_rcp(1.0f + __builtin_amdgcn_exp2f(y[k])); \
                _Pragma("unroll") for (int k = 0; k < 4; k++) HB[k] = __builtin_fmaf(2.0f, y[k], -1.0f); \
            }
#define SH_G32_BLEND(G)                                                                    \
            {                                                                              \
                const f32x4 z = (SH_G32_ZG || SH_G32_ZC) ? grp16<G>(za) : (SH_G32_ABL & 16) ? grp16<G>(za) * (0.25f * SH_OINV) + 0.5f : g32_logistic(grp16<G>(za));  \
                f32x4 hn;                                                                  \
                if (SH_G32_BLEND2) {                                                       \
                    f32x4 hbar;                                                            \
                    if (SH_G32_ZC) hbar = hb[G]; else SH_G32_HBAR(G, hbar)                 \
                    _Pragma("unroll") for (int k = 0; k < 4; k++) hn[k] = __builtin_fmaf(z[k], h[G][k] - hbar[k], hbar[k]); \
                } else {                                                                   \
                    const f32x4 hbar = d_tanh4_acc(grp16<G>(acc));                         \
                    hn = z * h[G] + (1.0f - z) * hbar;                                     \
                }                                                                          \
                _Pragma("unroll") for (int k = 0; k < 4; k++) h[G][k] = active ? hn[k] : 0.0f; \
            }
            f32x4 hb[4];
            if (SH_G32_ZC) {
                static_assert(!SH_G32_ZC || SH_G32_BLEND2, "SH_G32_ZC is written for the three-operation blend");
                SH_G32_HBAR(0, hb[0]) SH_G32_HBAR(1, hb[1]) SH_G32_HBAR(2, hb[2]) SH_G32_HBAR(3, hb[3])
                /* the update gate of this step, unless the flag read above came too early (rare: SIMD 3's waves have nothing else to do first) */
                if (__builtin_amdgcn_readfirstlane(zf) < (unsigned)(it + 1)) {
                    do { asm volatile("" ::: "memory"); zf = *(volatile unsigned *)(ZF + j); } while (__builtin_amdgcn_readfirstlane(zf) < (unsigned)(it + 1));
                    asm volatile("" ::: "memory");
                    za = acc_read(ring(0, j), lane);
                }
            }
            SH_G32_BLEND(0) SH_G32_BLEND(1) SH_G32_BLEND(2) SH_G32_BLEND(3)
#undef SH_G32_BLEND
#undef SH_G32_HBAR
            if (STAMP) asm volatile("" : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]));
            QMARK(4);                                      /* logistic(z), tanh, blend */
            if (t < hT && !((SH_G32_ABL & 32) && it > 0)) {
                float *ob = out + (c.boff0 + t) * 1536 + j * 512;
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    f32x4 o = h[g];
                    if (RESID) o += rs[g];
                    gstore_so(ob, voff, (g >> 1) * 1024 + (g & 1) * 512, o);
                }
            }
            c.s++;
            if (c.s == c.s1) {                             /* segment done */
                if (c.s1 < c.Tt) {                         /* the pair continues on another lane */
                    float *hs = L.hstate + ((long long)c.pair * 3 + j) * 1024 + lane * 4;
#pragma unroll
                    for (int g = 0; g < 4; g++)
#pragma unroll
                        for (int k = 0; k < 4; k++) __hip_atomic_store(hs + g * 256 + k, h[g][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    if (lane == 0) __hip_atomic_fetch_add(L.flag + c.pair, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
                c.sgi++;
                enter();
                take_over();
            }
            QMARK(5);                                      /* output store, segment bookkeeping */
            publish_own(H, h);
            QMARK(6);                                      /* cut + LDS writes issued */
            GSTAMP(sc);
            lds_barrier();
            GSTAMP(sd);
            if (STAMP) qt0 = __builtin_readcyclecounter();
        }
    } else if (wave >= 4 && wave < 7) {
        /* ------------------------------ G_j: gates z and r off the chain ------------------------------ */
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
        lds_barrier();
        f32x16 xz = bias_read(BIAS, j, lane), xr = bias_read(BIAS, 3 + j, lane);
#pragma unroll
        for (int ks = 0; ks < 6; ks++) {
            const ShSplit ip = load_pieces(IN(0) + ks * 512, lane);
            xz = split_k32(wz[ks], ip, xz);
            xr = split_k32(wrr[ks], ip, xr);
        }
        acc_write(ring(1, j), lane, xr);
        lds_barrier();
        if (STAMP) st0 = __builtin_readcyclecounter();
        for (int it = 0; it < nit; it++) {
            const unsigned *nin = IN((it + 1) & 1);
            /* interval A: the update gate of block `it` = its projection (kept from the last step) + sW_z . h, then the
             * projection of block it + 1; twelve B operands (6 k steps of h, 6 of the input column) streamed SH_G32_D ahead */
            ShSplit q[12];
            f32x16 zpre;
            auto item = [&](const int i) { return load_pieces((i < 6 ? H + i * 512 : nin + (i - 6) * 512), lane); };
#pragma unroll
            for (int i = 0; i < SH_G32_D; i++) q[i] = item(i);
#pragma unroll
            for (int i = 0; i < 12; i++) {
                if (i + SH_G32_D < 12) q[i + SH_G32_D] = item(i + SH_G32_D);
                if (SH_G32_ABL & 1) xz[i] += (float)q[i].p1[0];
                else if (i < 6) xz = split_k32(uz[i], q[i], xz);
                else {
                    xz = split_k32(wz[i - 6], q[i], xz);
                    xr = split_k32(wrr[i - 6], q[i], xr);
                }
                if (i == 5) {
                    zpre = xz;
                    if (!SH_G32_ZG) acc_write(ring(0, j), lane, zpre);
                    xz = bias_read(BIAS, j, lane);
                    xr = bias_read(BIAS, 3 + j, lane);
                }
                if (i == 5 + SH_G32_KA) {
                    if (SH_G32_ZG) {      /* (behind the MFMAs just issued: the logistic runs while they do) */
                        const f32x4 z0 = d_logistic4_acc(grp16<0>(zpre)), z1 = d_logistic4_acc(grp16<1>(zpre));
                        const f32x4 z2 = d_logistic4_acc(grp16<2>(zpre)), z3 = d_logistic4_acc(grp16<3>(zpre));
                        float *zp = ring(0, j);
                        *(f32x4 *)(zp + lane * 4) = z0; *(f32x4 *)(zp + 256 + lane * 4) = z1;
                        *(f32x4 *)(zp + 512 + lane * 4) = z2; *(f32x4 *)(zp + 768 + lane * 4) = z3;
                    }
                    GSTAMP(sa);
                    lds_barrier();
                    GSTAMP(sb);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            acc_write(ring(1, j), lane, xr);            /* (R_j read block it's reset-gate input in interval A) */
            GSTAMP(sc);
            lds_barrier();
            GSTAMP(sd);
        }
    } else if (wave == 3) {
        /* ------------------------------ C: the candidate's projection ------------------------------ */
        ShSplit w[3][6];
#pragma unroll
        for (int m = 0; m < 3; m++)
#pragma unroll
            for (int ks = 0; ks < 6; ks++) w[m][ks] = load_pieces(iWp + ((6 + m) * 6 + ks) * 512, lane);
#pragma unroll
        for (int m = 0; m < 3; m++)
#pragma unroll
            for (int ks = 0; ks < 6; ks++) asm volatile("" : "+v"(w[m][ks].p1), "+v"(w[m][ks].p2));
        lds_barrier();
        f32x16 a0 = bias_read(BIAS, 6, lane), a1 = bias_read(BIAS, 7, lane), a2 = bias_read(BIAS, 8, lane);
        auto ksteps = [&](const unsigned *ibuf, const int k0, const int k1) {
            ShSplit q[6];
#pragma unroll
            for (int ks = k0; ks < k1 && ks < k0 + SH_G32_D; ks++) q[ks] = load_pieces(ibuf + ks * 512, lane);
#pragma unroll
            for (int ks = k0; ks < k1; ks++) {
                if (ks + SH_G32_D < k1) q[ks + SH_G32_D] = load_pieces(ibuf + (ks + SH_G32_D) * 512, lane);
                const ShSplit &ip = q[ks];
                if (SH_G32_ABL & 1) { a0[ks] += (float)ip.p1[0]; continue; }
                a0 = mfma32(w[0][ks].p1, ip.p2, a0); a1 = mfma32(w[1][ks].p1, ip.p2, a1); a2 = mfma32(w[2][ks].p1, ip.p2, a2);
                a0 = mfma32(w[0][ks].p2, ip.p1, a0); a1 = mfma32(w[1][ks].p2, ip.p1, a1); a2 = mfma32(w[2][ks].p2, ip.p1, a2);
                a0 = mfma32(w[0][ks].p1, ip.p1, a0); a1 = mfma32(w[1][ks].p1, ip.p1, a1); a2 = mfma32(w[2][ks].p1, ip.p1, a2);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        ksteps(IN(0), 0, SH_G32_KC);
        lds_barrier();
        if (STAMP) st0 = __builtin_readcyclecounter();
        for (int it = 0; it < nit; it++) {
            const int par = it & 1;
            ksteps(IN(par), SH_G32_KC, 6);                /* interval A: the rest of block `it` */
            acc_write(ring(2, 0), lane, a0);
            acc_write(ring(2, 1), lane, a1);
            acc_write(ring(2, 2), lane, a2);
            GSTAMP(sa);
            lds_barrier();
            GSTAMP(sb);
            if (SH_G32_ZC) {      /* first: the chain waves want it behind their candidate products */
                z_in_place(0, it);
                __builtin_amdgcn_sched_barrier(0);
                z_in_place(1, it);
                __builtin_amdgcn_sched_barrier(0);
            }
            a0 = bias_read(BIAS, 6, lane); a1 = bias_read(BIAS, 7, lane); a2 = bias_read(BIAS, 8, lane);
            ksteps(IN(par ^ 1), 0, SH_G32_KC);            /* interval B: the first k steps of block it + 1 */
            GSTAMP(sc);
            lds_barrier();
            GSTAMP(sd);
        }
    } else {
        /* ------------------------------ L: the input column, two blocks ahead, as pieces ------------------------------ */
        struct Q { f32x4 v[3][4]; };
        int lastT = 0;
        auto fetch = [&]() {
            Q q;
            if (c.ok) lastT = backward ? c.Tt - 1 - c.s : c.s;       /* past the lane's end: the last block again */
            const float *cb = in + c.boff0 * 1536;
            const unsigned o = block_off(lastT);
#pragma unroll
            for (int j = 0; j < 3; j++)
#pragma unroll
                for (int g = 0; g < 4; g++) q.v[j][g] = gload_so(cb, o, (2 * j + (g >> 1)) * 1024 + (g & 1) * 512);
            if (c.ok) {
                c.s++;
                if (c.s == c.s1) {
                    const long long keep = c.boff0; const int keepT = c.Tt;
                    c.sgi++; enter();
                    if (!c.ok) { c.boff0 = keep; c.Tt = keepT; }
                }
            }
            return q;
        };
        auto cut = [&](const Q &q, u32x4 (&p1)[3][2], u32x4 (&p2)[3][2]) {
#pragma unroll
            for (int j = 0; j < 3; j++) {
                if (SH_G32_ABL & 8) { p1[j][0] = p1[j][1] = p2[j][0] = p2[j][1] = __builtin_bit_cast(u32x4, q.v[j][0]); }
                else cut16(q.v[j], p1[j], p2[j]);
            }
        };
        auto put = [&](unsigned *buf, const u32x4 (&p1)[3][2], const u32x4 (&p2)[3][2]) {
#pragma unroll
            for (int j = 0; j < 3; j++) put16(buf, j, lane, p1[j], p2[j]);
        };
        enter();
        Q e0 = fetch(), e1 = fetch();
        {
            u32x4 p1[3][2], p2[3][2];
            cut(e0, p1, p2); put(IN(0), p1, p2);
            e0 = fetch();
            cut(e1, p1, p2); put(IN(1), p1, p2);
            e1 = fetch();
        }
        lds_barrier();
        lds_barrier();
        if (STAMP) st0 = __builtin_readcyclecounter();
        /* the queue does not shift (two steps per trip, the entries' roles fixed at compile time) */
        auto step = [&](Q &e, const int par, const int it) {
            u32x4 p1[3][2], p2[3][2];
            cut(e, p1, p2);                                /* block it + 2, in registers until the slot is free */
            if (SH_G32_ZC) {                               /* (cut HERE, in interval A: behind the barrier this wave has the update gate to do first) */
#pragma unroll
                for (int jj = 0; jj < 3; jj++) asm volatile("" : "+v"(p1[jj][0]), "+v"(p1[jj][1]), "+v"(p2[jj][0]), "+v"(p2[jj][1]));
            }
            GSTAMP(sa);
            lds_barrier();
            GSTAMP(sb);
            if (SH_G32_ZC) { z_in_place(2, it); __builtin_amdgcn_sched_barrier(0); }
            put(IN(par), p1, p2);
            e = fetch();                                   /* block it + 4 */
            GSTAMP(sc);
            lds_barrier();
            GSTAMP(sd);
        };
        for (int it = 0; it < nit; it += 2) {
            step(e0, 0, it);
            if (it + 1 < nit) step(e1, 1, it + 1);
        }
    }
    if (STAMP && dbg && lane == 0) {
        unsigned long long *d = dbg + ((long long)blockIdx.x * 8 + wave) * 16;
        d[0] = sa; d[1] = sb; d[2] = sc; d[3] = sd; d[4] = (unsigned long long)nit;
#pragma unroll
        for (int i = 0; i < 8; i++) d[5 + i] = q[i];
    }
#undef GSTAMP
#undef QMARK
}

#endif /* SH_GRU32_H */
