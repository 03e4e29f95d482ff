/* scrappie_hip.hip -- engine + C ABI of libscrappie_hip.so (gfx950 only).
 *
 * One engine = one GPU, three HIP streams (kernels; results -> host; host signals -> device),
 * grow-only device arena sized for HBM (a launch group is bounded by device memory: about 100 KB per
 * tile and block for the transducer models).  Reads handed to the engine are coalesced into LAUNCH
 * GROUPS: sorted by block count, cut into tiles of 16, and pushed through
 *
 *   rgrgr / rnnrf:   k_conv_act -> 5 x k_gru_proj (projection team + recurrence team, gate inputs in LDS)
 *                    -> k_ff_viterbi (S1 inside the decoder; 4^5 + 1 states over 96 units)
 *                       | k_ff_lds / k_ff_exp -> k_viterbi                       (other shapes, posterior wanted)
 *                    -> k_backtrace                                            (rnnrf: k_affine -> k_crf)
 *   input != state width, S not in {32, 64, 96}, SH_GRU_SEPARATE:
 *                    ... 5 x (k_affine[_lds] -> k_gru_split | k_gru) ...
 *   raw_r94:         k_conv_act -> 2 x {k_gru_proj fwd, bwd -> k_affine2_tanh} -> S1 -> decode
 *   events:          k_feat_in -> 2 x {k_lstm_proj fwd, bwd (projection + peephole LSTM in one kernel) -> k_affine2_tanh} -> S1 -> decode
 *                    (SH_GRU_SEPARATE: k_affine + k_lstm_lanes)
 *   a GRU layer with a weight outside the split products' operand range (|w| >= 255): k_affine<.., F32> -> k_gru_lanes (exact fp32)
 *
 * The contractions of the projection, the recurrence and S1 run as split products on the f16 matrix pipe
 * (sh_kernels.h, split_pair / split_dot): fp32 in, fp32 out, fp32 accuracy.  The recurrent kernels walk a lane schedule and
 * the decoder works on pieces of tiles (sh_sched.h).  Two launch groups can be in flight: the host stitches
 * group k (homopolymer correction, k-mer overlap: sh_host.c, C) while group k+1 runs.  Only decoded paths
 * (and the 5-row homopolymer side buffer) cross PCIe.
 */
#include <hip/hip_runtime.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>

#include "scrappie_hip.h"
#include "sh_internal.h"
#include "sh_kernels.h"
#include "sh_sched.h"
#include "sh_dev.h"
#include "sh_coalesce.h"

/* function attributes (dynamic LDS limit) are per device: remember for which devices a kernel has had its attribute set
 * (engines on several GPUs may share one process).  A real once per device: the thread that finds the attribute unset holds
 * the lock until hipFuncSetAttribute has returned (`if (auto turn = once.first()) HIPCHK(hipFuncSetAttribute(...));` -- the
 * turn lives to the end of the if statement), so a second engine on the same device -- the helper engine runs launch groups
 * on a host thread of its own -- cannot launch the kernel with more than 64 KB of dynamic LDS before the limit is raised. */
struct DevOnce {
    std::atomic<unsigned long long> done{0};
    std::mutex mu;
    struct Turn {
        DevOnce *o; unsigned long long bit;
        Turn(DevOnce *o_, unsigned long long b_) : o(o_), bit(b_) {}
        Turn(Turn &&t) : o(t.o), bit(t.bit) { t.o = nullptr; }
        Turn(const Turn &) = delete;
        explicit operator bool() const { return o != nullptr; }
        ~Turn() { if (o) { o->done.fetch_or(bit, std::memory_order_release); o->mu.unlock(); } }
    };
    Turn first(bool wanted = true) {
        if (!wanted) return Turn(nullptr, 0);
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned long long bit = 1ull << (dev & 63);
        if (done.load(std::memory_order_acquire) & bit) return Turn(nullptr, 0);
        mu.lock();
        if (done.load(std::memory_order_acquire) & bit) { mu.unlock(); return Turn(nullptr, 0); }
        return Turn(this, bit);
    }
};

/* Development switches (kernel-family selection, cycle stamps).  Read from the environment ONCE, when the
 * first engine is created -- never on the launch path. */
struct Tunables {
    bool affine_reg, gru_single, gru_stamp, gru_separate, gru_f32, gru_lanes_stamp, proj_stamp, ff_reg, ff_stamp, vit_stamp, ff_separate, fv_single, host_stamp, host_stitch, helper_fence, gru_free, gru_barrier, input_order, conv_valu, conv_in_layer, gru32, gru16, gru32_stamp;
    int gru_debug;       /* -1: off */
    int conv_tchunk;     /* blocks per workgroup pass of k_conv_act (SH_CONV_TCHUNK, default 16) */
    double gru_two_ratio; /* step time of a two-tile workgroup of k_gru_proj over a one-tile one (SH_GRU_TWO_RATIO, default 1.8) */
    bool fake_timeout;   /* SH_FAKE_HANDOVER_TIMEOUT: collect() treats the first launch group as timed out (test hook) */
    Tunables() {
        auto on = [](const char *k) { return getenv(k) != nullptr; };
        /* experiment switches (kernel forms measured and not adopted, cycle stamps, scheduling experiments) exist only in the experiments
         * build (-DSH_EXPERIMENTS: libscrappie_hip_exp.so, what the tests of those forms load); the product library does not carry the kernels */
#ifdef SH_EXPERIMENTS
        auto xon = on;
#else
        auto xon = [](const char *) { return false; };
#endif
        affine_reg = xon("SH_AFFINE_REG"); gru_single = xon("SH_GRU_SINGLE"); gru_stamp = xon("SH_GRU_STAMP");
        gru_separate = on("SH_GRU_SEPARATE"); gru_f32 = on("SH_GRU_F32"); gru_lanes_stamp = xon("SH_GRU_LANES_STAMP");
        proj_stamp = xon("SH_PROJ_STAMP"); ff_reg = xon("SH_FF_REG"); ff_stamp = xon("SH_FF_STAMP"); vit_stamp = xon("SH_VIT_STAMP");
        host_stamp = xon("SH_HOST_STAMP");         /* host-side wall times of a launch group on stderr */
        gru_free = xon("SH_GRU_FREE");             /* recurrent layers on k_gru_free (no s_barrier in the step loop: LDS counters) */
        gru_barrier = xon("SH_GRU_BARRIER");       /* ... on k_gru_proj (two s_barriers per step shared by both teams) */
        helper_fence = xon("SH_HELPER_FENCE");     /* experiment: the first recurrent layer waits for the previous group's traceback walk + k_stitch */
        host_stitch = on("SH_HOST_STITCH");       /* homopolymer correction + k-mer stitching on host threads (paths + 5 rows over PCIe) instead of k_stitch */
        gru32 = xon("SH_GRU32");                   /* recurrent layers of S = 96 on tiles of 32 reads (k_gru_proj32) */
        gru16 = xon("SH_GRU16");                   /* ... on tiles of 16 reads (k_gru_proj) */
        gru32_stamp = xon("SH_GRU32_STAMP");       /* ... with cycle stamps of one launch on stderr */
        conv_in_layer = xon("SH_CONV_IN_LAYER");   /* experiment (measured 1 ms per step SLOWER): the first recurrent layer of the rgrgr models computes the convolution itself (k_gru_conv) */
        conv_valu = on("SH_CONV_VALU");           /* the convolution as VALU multiplies and additions (k_conv_act) where k_conv_mfma applies */
        input_order = xon("SH_INPUT_ORDER");       /* experiment: a call's launch groups cut in input order instead of sorted by length */
        ff_separate = on("SH_FF_SEPARATE");      /* S1 and the decoder as two kernels even where k_ff_viterbi applies */
        fv_single = on("SH_FV_SINGLE");          /* S1 inside the decoder on eight do-everything waves (k_ff_viterbi) instead of two teams (k_ff_viterbi_teams) */
        fake_timeout = on("SH_FAKE_HANDOVER_TIMEOUT");
#ifdef SH_EXPERIMENTS
        const char *dm = getenv("SH_GRU_DEBUG");
#else
        const char *dm = nullptr;
#endif
        gru_debug = dm ? atoi(dm) : -1;
        const char *tc = getenv("SH_CONV_TCHUNK");
        conv_tchunk = tc ? std::max(1, std::min(atoi(tc), 256)) : 16;
        const char *tr = getenv("SH_GRU_TWO_RATIO");
        gru_two_ratio = tr ? atof(tr) : 1.8;
    }
};
static const Tunables &tun() { static const Tunables t; return t; }

#ifndef SH_GRU_FREE_DEFAULT
#define SH_GRU_FREE_DEFAULT 0     /* 1: recurrent layers run k_gru_free unless SH_GRU_BARRIER is set; 0: k_gru_proj unless SH_GRU_FREE is set */
#endif
#ifndef SH_AFF_NB
#define SH_AFF_NB 3      /* column blocks per wave in k_affine_lds */
#endif
#ifndef SH_AFF_NTH
#define SH_AFF_NTH 512   /* threads per workgroup in k_affine_lds */
#endif
#ifndef SH_FF_NB
#define SH_FF_NB 4     /* column blocks per wave in k_ff_exp */
#endif

/* errors (thread-local text behind scrappie_hip_last_error), HIPCHK, grow-only device / pinned buffers: sh_dev.h */
static thread_local char g_err[512] = "";
int sh_set_err_v(const char *fmt, va_list ap) {
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    return -1;
}
extern "C" const char *scrappie_hip_last_error(void) { return g_err; }

/* ------------------------------------------------------------------ */
/* model                                                                */
/* ------------------------------------------------------------------ */
static ShConvGeom conv_geom(int WL, int st, int F) {      /* layers.c:169-207 */
    ShConvGeom g;
    g.WL = WL; g.st = st; g.F = F;
    g.padL = (g.WL - 1) / 2; g.padR = g.WL / 2;
    g.c0 = (g.padL + g.st - 1) / g.st;
    g.shiftX = g.c0 * g.st - g.padL;
    g.nstepC = (g.WL + g.st - 1) / g.st;
    g.nstepX = g.st * g.nstepC;
    return g;
}

struct HostMat { int nr = 0, nc = 0; std::vector<float> v; };   /* v[c*nr + r]: column c = output unit */

struct Model {
    std::string name;
    int arch = 0, conv_act = 0, stride = 5;
    int F = 0, WL = 0, S = 0, NS = 0;
    ShConvGeom geom{};
    size_t min_samples = 0;
    /* device weights */
    DBuf conv_W, conv_b;                 /* [WL][F], [F] */
    DBuf iW[5], ib[5], sW[5], sW2[5];    /* fragments (fp32: exact-fp32 MFMA kernels) */
    DBuf iWp[5], sWp[5], sW2p[5], ffWp;  /* the same rows as fp16 pieces (split products: sh_kernels.h) */
    DBuf ibs[5], ffbs;                   /* biases in the split products' accumulator units (x 2^14) */
    DBuf iWp32[5], sWp32[5], sW2p32[5], ib32[5];   /* GRU layers of S = 96 for k_gru_proj32: pieces in 32-row m-tiles / 16-wide k steps, bias table */
    bool has32 = false;
    DBuf ffW, ffb;
    int ff_mtiles = 0;
    DBuf ff2W[2][2], ff2b[2];            /* raw_r94 / events: FF1/FF2 {Wf, Wb}, b (feedforward2_tanh) */
    DBuf lp[4];                          /* events: LSTM peepholes [update | forget | output] in accumulator layout */
    int nfeat = 0;                       /* events: input features per event (12), padded to F = 16 */
    bool layer_f32[5] = {false, false, false, false, false};   /* GRU layer has a weight outside the split products' range: exact-fp32 kernels */
    void release() {
        conv_W.release(); conv_b.release(); ffW.release(); ffb.release();
        for (int l = 0; l < 5; l++) { iW[l].release(); ib[l].release(); sW[l].release(); sW2[l].release(); iWp[l].release(); sWp[l].release(); sW2p[l].release(); }
        ffWp.release(); ffbs.release();
        for (int l = 0; l < 5; l++) { ibs[l].release(); iWp32[l].release(); sWp32[l].release(); sW2p32[l].release(); ib32[l].release(); }
        for (int k = 0; k < 2; k++) { ff2W[k][0].release(); ff2W[k][1].release(); ff2b[k].release(); }
        for (int l = 0; l < 4; l++) lp[l].release();
    }
};

/* MFMA A fragments of an (M x K) weight matrix given as rows m (output unit)
 * of K inputs: frag[(mt*(K/4) + r)*64 + l] = W[16mt + (l&15)][16*(r>>2) + 4*(l>>4) + (r&3)] */
static std::vector<float> make_frags(const HostMat &w, int &mtiles) {
    const int M = w.nc, K = w.nr;
    mtiles = (M + 15) / 16;
    const int KR = K / 4;
    std::vector<float> f((size_t)mtiles * KR * 64, 0.0f);
    for (int mt = 0; mt < mtiles; mt++)
        for (int r = 0; r < KR; r++)
            for (int l = 0; l < 64; l++) {
                const int m = 16 * mt + (l & 15);
                const int k = 16 * (r >> 2) + 4 * (l >> 4) + (r & 3);
                if (m < M) f[((size_t)mt * KR + r) * 64 + l] = w.v[(size_t)m * K + k];
            }
    return f;
}

/* fp32 -> fp16, round to nearest even (the device's v_cvt_f16_f32), and back */
static uint16_t f32_to_f16_rne(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0));       /* inf / nan */
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                                          /* rounds to >= 65520: inf */
    if (x < 0x33000001u) return (uint16_t)sign;                                                       /* <= 2^-25: zero */
    if (x < 0x38800000u) {                                                                            /* subnormal half */
        const int e = (int)(x >> 23);                      /* 102 .. 112 */
        uint32_t m = (x & 0x7fffffu) | 0x800000u;
        const int shift = 126 - e;                         /* 14 .. 24: m * 2^(e - 150) in units of 2^-24 */
        const uint32_t r = m >> shift, rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
        return (uint16_t)(sign | (r + ((rem > half || (rem == half && (r & 1))) ? 1 : 0)));
    }
    const uint32_t r = x - 0x38000000u;                    /* rebias: 127 -> 15 */
    const uint32_t h = r >> 13, rem = r & 0x1fffu;
    return (uint16_t)(sign | (h + ((rem > 0x1000u || (rem == 0x1000u && (h & 1))) ? 1 : 0)));
}
static float f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31, m = h & 0x3ffu;
    float f;
    if (e == 0) f = ldexpf((float)m, -24);
    else if (e == 31) f = m ? NAN : INFINITY;
    else f = ldexpf((float)(m | 0x400u), (int)e - 25);
    uint32_t x; memcpy(&x, &f, 4); x |= sign; memcpy(&f, &x, 4);
    return f;
}

/* The same rows as fp16 pieces for the split products (sh_kernels.h): 256 x = p1 + p2, cut here once.
 * piece[((mt*KS + ks)*2 + pc)*256 + l*4 + w] holds, as two halves (low = j even), the values
 * j = 2w, 2w+1 of W[16mt + (l&15)][32ks + 16(j>>2) + 4(l>>4) + (j&3)] -- the 8 values of k lane l feeds to one
 * v_mfma_f32_16x16x32_f16, in the order the activations' chunks deliver them. */
static std::vector<uint32_t> make_piece_frags(const HostMat &w) {
    const int M = w.nc, K = w.nr;
    const int mtiles = (M + 15) / 16, KS = K / 32;
    std::vector<uint32_t> f((size_t)mtiles * KS * 2 * 256, 0u);
    for (int mt = 0; mt < mtiles; mt++)
        for (int ks = 0; ks < KS; ks++)
            for (int l = 0; l < 64; l++)
                for (int j = 0; j < 8; j++) {
                    const int m = 16 * mt + (l & 15);
                    const int k = 32 * ks + 16 * (j >> 2) + 4 * (l >> 4) + (j & 3);
                    if (m >= M) continue;
                    const float x = w.v[(size_t)m * K + k] * SH_WSCALE;
                    const uint16_t p1 = f32_to_f16_rne(x);
                    const uint16_t p2 = f32_to_f16_rne(x - f16_to_f32(p1));
                    const size_t base = ((size_t)(mt * KS + ks) * 2) * 256 + (size_t)l * 4 + (j >> 1);
                    f[base] |= (uint32_t)p1 << (16 * (j & 1));
                    f[base + 256] |= (uint32_t)p2 << (16 * (j & 1));
                }
    return f;
}
/* ... and for v_mfma_f32_32x32x16_f16 (k_gru_proj32, sh_gru32.h): m-tiles of 32 rows, k steps of 16.
 * piece[((mt*KS + ks)*2 + pc)*256 + l*4 + w]: values j = 2w, 2w+1 of W[32mt + (l&31)][16ks + 8(j>>2) + 4(l>>5) + (j&3)] -- the k order in
 * which a lane's 16 accumulator values (units 8g + 4(l>>5) + 0..3, g = 0..3) become the B operand of two k steps. */
static std::vector<uint32_t> make_piece_frags32(const HostMat &w) {
    const int M = w.nc, K = w.nr;
    const int mtiles = M / 32, KS = K / 16;
    std::vector<uint32_t> f((size_t)mtiles * KS * 2 * 256, 0u);
    for (int mt = 0; mt < mtiles; mt++)
        for (int ks = 0; ks < KS; ks++)
            for (int l = 0; l < 64; l++)
                for (int j = 0; j < 8; j++) {
                    const int m = 32 * mt + (l & 31);
                    const int k = 16 * ks + 8 * (j >> 2) + 4 * (l >> 5) + (j & 3);
                    const float x = w.v[(size_t)m * K + k] * SH_WSCALE;
                    const uint16_t p1 = f32_to_f16_rne(x);
                    const uint16_t p2 = f32_to_f16_rne(x - f16_to_f32(p1));
                    const size_t base = ((size_t)(mt * KS + ks) * 2) * 256 + (size_t)l * 4 + (j >> 1);
                    f[base] |= (uint32_t)p1 << (16 * (j & 1));
                    f[base + 256] |= (uint32_t)p2 << (16 * (j & 1));
                }
    return f;
}
/* bias table of k_gru_proj32, accumulator units: t[(mt*2 + hf)*16 + r] = 2^14 b[32mt + 8(r>>2) + 4hf + (r&3)] */
static std::vector<float> make_bias32(const HostMat &b) {
    const int M = b.nr * b.nc, mtiles = M / 32;
    std::vector<float> t((size_t)mtiles * 32, 0.0f);
    for (int mt = 0; mt < mtiles; mt++)
        for (int hf = 0; hf < 2; hf++)
            for (int r = 0; r < 16; r++) t[((size_t)mt * 2 + hf) * 16 + r] = b.v[32 * mt + 8 * (r >> 2) + 4 * hf + (r & 3)] * SH_OSCALE;
    return t;
}
/* operand range of the split products (sh_kernels.h): |w| * SH_WSCALE must stay a finite fp16 */
static float max_abs(const HostMat &w) {
    float m = 0.0f;
    for (float x : w.v) { const float a = std::fabs(x); if (std::isnan(a)) return a; if (a > m) m = a; }      /* a NaN anywhere is the answer */
    return m;
}
static bool in_split_range(const HostMat &w) { return max_abs(w) < SH_W_LIMIT; }

static int upload_u32(DBuf &d, const std::vector<uint32_t> &h) {
    if (d.ensure(std::max<size_t>(h.size(), 1) * 4)) return -1;
    if (!h.empty() && hipMemcpy(d.p, h.data(), h.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return set_err("hipMemcpy of weights failed");
    return 0;
}

/* bias in accumulator (D) layout: bf[(mt*64 + l)*4 + r] = b[16mt + 4*(l>>4) + r] */
static std::vector<float> make_bias_frags(const HostMat &b, int mtiles) {
    const int M = b.nr * b.nc;
    std::vector<float> f((size_t)mtiles * 256, 0.0f);
    for (int mt = 0; mt < mtiles; mt++)
        for (int l = 0; l < 64; l++)
            for (int r = 0; r < 4; r++) {
                const int m = 16 * mt + 4 * (l >> 4) + r;
                if (m < M) f[((size_t)mt * 64 + l) * 4 + r] = b.v[m];
            }
    return f;
}

/* ... in the accumulator units of the split products (2^14) */
static std::vector<float> scaled(std::vector<float> v, float f) { for (float &x : v) x *= f; return v; }

static int upload(DBuf &d, const std::vector<float> &h) {
    if (d.ensure(h.size() * sizeof(float))) return -1;
    HIPCHK(hipMemcpy(d.p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

/* ------------------------------------------------------------------ */
/* engine                                                               */
/* ------------------------------------------------------------------ */
struct LaunchGroup {
    size_t n = 0, npad = 0, ntile = 0;
    long long ncb = 0;            /* total column blocks */
    long long nseq = 0;           /* total path ints */
    long long nhp = 0;            /* total blocks of real reads (hp side rows) */
    std::vector<int> order;       /* tiled index -> original read index (or -1) */
    std::vector<int> rT, rN;
    std::vector<long long> seq_off, hp_off, bases_off;
    long long nbases_cap = 0;     /* bytes of the per-slot bases buffer (k_stitch) */
    bool dev_stitch = false;      /* bases were made on the device (k_stitch); else paths (+ side rows) come to the host */
    bool dev_pos = false;         /* ... and pos[] too */
    int model = -1;
    bool hp_on = false;
    bool valid = false;
    int gru_nwg = 0;              /* lane schedule of the recurrent kernel (sh_sched.h) */
    int gru1_nwg = 0;             /* ... with one lane per workgroup (k_gru_proj with fewer tiles than CUs) */
    bool gru_two = false;         /* more live tiles than CUs: k_gru_proj steps two tiles per workgroup */
    int gru32_nwg = 0;            /* ... over pairs of tiles, one pair at a time per workgroup (k_gru_proj32) */
    int gru32x2_nwg = 0;          /* ... two pairs at a time per workgroup (k_gru_proj32x2) */
    int vit_nwg = 0;              /* ... and of the Viterbi decoder */
    /* what the group was launched with, kept so that scrappie_hip_collect can run it again on whole tiles
     * should a state hand-over between workgroups time out */
    const float *d_signal = nullptr;
    std::vector<uint64_t> in_off;
    std::vector<uint32_t> in_len;
    scrappie_hip_params params{};
};

struct scrappie_hip_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t cstream = nullptr;   /* results -> host, so that the copy overlaps the next group's kernels */
    hipStream_t ustream = nullptr;   /* host signals -> device (scrappie_hip_basecall_batch), same reason */
    hipStream_t pstream = nullptr;   /* prologue of a launch group: metadata upload, flag clears, convolution -- runs under the PREVIOUS group's
                                        recurrent layers (k_conv_act fits beside k_gru_proj's waves), the main stream waits for pdone[slot] */
    hipEvent_t pdone[2];
    DBuf d_conv[2];                  /* convolution output per slot (the layers' ping-pong buffers belong to the group that is running) */
    hipEvent_t up[2];                /* upload into d_signal[k] finished */
    size_t total_mem = (size_t)64 << 30;
    size_t max_launch_blocks = 0;    /* column blocks (16 reads x 1 block) per launch group; 0 = from device memory */
    std::vector<Model *> models;
    size_t max_launch_reads = 16384;
    bool profiling = false;
    scrappie_hip_timing timing{};
    /* profiling: events are only RECORDED while a launch group runs (no host
     * synchronisation inside the timed region); elapsed times are read back in
     * scrappie_hip_get_timing after the stream has drained. */
    /* Two launch-group slots: group k+1 can be enqueued while the host is still
     * stitching group k (its metadata, pinned result buffers, completion event and
     * profiling events are per slot, and so are the device buffers the host reads back, which a
     * second stream copies out while the next group computes; all other device buffers are shared,
     * ordered by the stream). */
    hipEvent_t ev[2][48];
    hipEvent_t done[2];
    hipEvent_t kdone[2];         /* kernels of the slot finished (stream) -> copies may start (cstream) */
    hipEvent_t hdone[2];         /* traceback walk + k_stitch of the slot finished (cstream) */
    bool ev_ok = false;
    int evn = 0;
    struct Span { int field, i, j; };
    std::vector<Span> spans[2];
    int cur = 0;                 /* slot of the most recent run_pipeline */
    bool pending[2] = {false, false};
    int oldest = 0;              /* next slot collect() will take */
    /* arena */
    DBuf d_hstate, d_gflag[2], d_vstate, d_vflag;
    DBuf d_bad[2];                /* [npad] per slot: read whose input left the split products' operand range (k_conv_act, k_feat_in) */
    HBuf h_err[2], h_bad[2];
    DBuf d_pos[2], d_bases[2], d_blen[2], d_redo[2];     /* k_stitch: pos / bases / lengths / host-decides flags per slot */
    DBuf d_edge[2]; HBuf h_edge[2];                      /* k_gru_conv: where each read's convolution windows end (ShConvFuse::edge) */
    HBuf h_pos[2], h_bases[2], h_blen[2], h_redo[2];
    int ncu = 256;
    bool handover = true;         /* cut tiles between lanes / into pieces (SCRAPPIE_HIP_HANDOVER=0: whole tiles only) */
    DBuf d_meta[2], d_signal[2], d_act[3], d_xaff, d_E, d_sums, d_tb, d_tbend, d_fstate, d_fscore[2], d_seq[2], d_hp[2];
    HBuf h_meta[2], h_seq[2], h_score[2], h_hp[2], h_sig[2];
    LaunchGroup lgs[2];
    scrappie_hip_timing slot_timing[2];
    /* scrappie_hip_set_decoder_input: caller-supplied probabilities in place of the S1 output */
    const float *alt_prob = nullptr;
    std::vector<uint64_t> alt_off;
    DBuf d_Ealt, d_sums_alt, d_altoff;
    uint64_t alt_key = 0;
    bool alt_valid = false;
    /* scrappie_hip_set_trunk_input: caller-supplied trunk activations in front of S1 */
    const float *alt_trunk = nullptr;
    std::vector<uint64_t> trk_off;
    DBuf d_act_alt, d_trkoff;
    uint64_t trk_key = 0;
    bool trk_valid = false;
    /* scrappie_hip_debug_option */
    bool dbg_ff_separate = false;    /* S1 and the decoder as two kernels (as SH_FF_SEPARATE, per engine) */
    bool dbg_fv_single = false;      /* S1 inside the decoder on k_ff_viterbi's eight do-everything waves (as SH_FV_SINGLE, per engine) */
    bool dbg_dump_final = false;     /* decoders leave every tile's final scores in d_vstate */
    int dbg_fail_run = 0;            /* k > 0: the k-th next launch group is refused (failure-path tests) */
    bool dbg_redo_all = false;       /* treat every read as one k_stitch left to the host (tests the fallback) */
    int dbg_gru_tiles = 0;           /* 1 / 2: tiles per workgroup of k_gru_proj whatever the schedules say (0: choose) */
    bool dbg_force_f32 = false;      /* models loaded from now on run their GRU layers on the exact-fp32 kernels (as if out of the split products' range) */
    int dbg_gru32 = -1;              /* 0 / 1: recurrent layers on 16- / 32-read tiles whatever the build's default (-1) */
    /* chain-bound reads beside the rest of a call (scrappie_hip_basecall_batch): a helper engine on the same device, created on first use */
    scrappie_hip_engine *tail = nullptr;
    scrappie_hip_engine *tail2 = nullptr;   /* a second helper, created when a ticket arrives while the first is busy: a chain-bound launch group lasts
                                               as long as its longest read whatever it holds, so a stream of calls with a heavy tail keeps two going */
    bool is_tail = false;
    int tail_mode = -1;              /* 0 / 1: never / whenever the plan says so; -1: SCRAPPIE_HIP_TAIL (default 1) */
    int dbg_fail_tail = 0;           /* k > 0: the helper engine's k-th next launch group is refused (failure-path tests) */
    double mem_frac = 0.7;           /* share of the device's memory a launch group's arena may take; creating the helper engine (the first call with
                                        chain-bound reads) lowers it to 0.45 for good -- the helper takes 0.15, a second helper another 0.15 -- so later calls
                                        cut slightly smaller launch groups whether or not they have a long tail (include/scrappie_hip.h) */
    struct Blob { std::string name; std::vector<unsigned char> bytes; bool force_f32; };
    std::vector<Blob> blobs;         /* the models as they were loaded (replayed into the helper engine) */
    unsigned long long n_tail_calls = 0, n_tail_reads = 0;       /* calls split so far, reads that went to the helper (debug_fetch) */
    /* the helper's host thread: takes ALL waiting tickets of one kind as one launch group -- chain-bound groups last as long as their
     * longest read however many long reads they hold, so the long reads of several calls cost what those of one call cost */
    struct TailTicket {
        long id = 0; int model = 0; scrappie_hip_params p{};
        std::vector<raw_table> reads; std::vector<scrappie_hip_call> calls;
        std::vector<float> own;          /* device-resident callers: the deferred reads' signals, copied back so that they outlive the caller's buffer */
        int rc = 0; std::string err; bool done = false;
    };
    std::thread tail_th, tail_th2;
    bool tail_th_live = false, tail_th2_live = false, tail_stop = false;
    int tail_busy = 0;                       /* helpers with a launch group in hand */
    std::mutex tail_mu;
    std::condition_variable tail_cv;
    std::deque<std::shared_ptr<TailTicket>> tail_q;
    std::map<long, std::shared_ptr<TailTicket>> tail_open;
    long tail_next = 1;
    unsigned long long n_redo_tail = 0;      /* reads the helper's k_stitch left to the host */
    unsigned long long n_tail_groups = 0;    /* launch-group calls the helper has made (fewer than tickets when tickets were merged) */
    unsigned host_thread_budget = 0; /* stitching threads of this engine while several engines share a call (0: host_threads()) */
    unsigned long long n_redo = 0;   /* reads k_stitch left to the host so far (scrappie_hip_debug_fetch "n_redo") */
    std::mutex mu;
};

static int pick_mt(int mtiles) {
    for (int mt : {6, 4, 3, 2}) if (mtiles % mt == 0) return mt;
    return 1;
}

/* HIP gives a process 4 hardware queues by default and maps its streams onto them round robin.  An engine owns 4
 * streams (main, prologue, helpers, upload); with anybody else's streams in the process (the null stream, torch,
 * RCCL, a second engine) two of them share a queue and work that is meant to overlap serialises (bench.py under a
 * process group: 29.7 against 28.2 ms per step; two engines on one device: no overlap at all, profiles/r3_long_tail.txt).
 * GPU_MAX_HW_QUEUES is read when the HIP runtime initialises, so this default only takes effect if the library makes the
 * process's first HIP call; a host that has HIP running already sets the variable itself.  Never
 * overrides a value the user has set. */
static void hw_queue_default() {
    static std::once_flag once;
    std::call_once(once, [] { (void)setenv("GPU_MAX_HW_QUEUES", "8", 0); });
}

static int device_count(hipError_t *why) {
    hw_queue_default();
    int n = 0;
    const hipError_t rc = hipGetDeviceCount(&n);
    if (why) *why = rc;
    return rc == hipSuccess ? n : 0;
}
extern "C" int scrappie_hip_device_count(void) { return device_count(nullptr); }

extern "C" scrappie_hip_engine *scrappie_hip_engine_create(int device) {
    hipError_t why = hipSuccess;
    int n = device_count(&why);
    if (n <= 0) {
        /* the usual reason on a box that has a GPU: two copies of the HIP runtime in one process (a torch wheel's own next to
         * /opt/rocm's) -- the one that initialises second finds no device */
        std::string first, second;
        if (FILE *maps = fopen("/proc/self/maps", "r")) {
            char line[1024];
            while (fgets(line, sizeof line, maps)) {
                const char *path = strchr(line, '/');
                if (!path || !strstr(path, "libamdhip64")) continue;
                std::string s(path, strcspn(path, "\n"));
                if (first.empty()) first = s; else if (s != first && second.empty()) second = s;
            }
            fclose(maps);
        }
        if (!second.empty())
            set_err("no HIP device visible (hipGetDeviceCount: %s): two HIP runtimes are loaded in this process (%s and %s); load the "
                    "other user of HIP (e.g. import torch) before this library, see INTEGRATION.md", hipGetErrorString(why), first.c_str(), second.c_str());
        else set_err("no HIP device visible (hipGetDeviceCount: %s)", hipGetErrorString(why));
        return nullptr;
    }
    if (device < 0 || device >= n) { set_err("device %d out of range (have %d)", device, n); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { set_err("hipSetDevice(%d) failed", device); return nullptr; }
    hipDeviceProp_t prop;
    int ncu = 256;
    size_t total_mem = (size_t)64 << 30;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        if (prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
        if (prop.totalGlobalMem > 0) total_mem = prop.totalGlobalMem;
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            set_err("device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
            return nullptr;
        }
    }
    (void)tun();                                   /* development switches: environment read here, once */
    scrappie_hip_engine *e = new scrappie_hip_engine();
    e->device = device;
    e->ncu = ncu;
    e->total_mem = total_mem;
    { const char *h = getenv("SCRAPPIE_HIP_HANDOVER"); if (h && atoi(h) == 0) e->handover = false; }
    /* the main stream at the highest priority: where a helper kernel and a recurrent layer compete for a CU, the layer goes first */
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
#ifdef SH_EXPERIMENTS
    if (!getenv("SH_STREAM_PRIO"))
#endif
    prio_lo = prio_hi = 0;      /* experiment switch: stream priorities off unless asked for */
    if (hipStreamCreateWithPriority(&e->stream, hipStreamNonBlocking, prio_hi) != hipSuccess ||
        hipStreamCreateWithPriority(&e->cstream, hipStreamNonBlocking, prio_lo) != hipSuccess ||
        hipStreamCreateWithPriority(&e->pstream, hipStreamNonBlocking, prio_lo) != hipSuccess ||
        hipStreamCreateWithFlags(&e->ustream, hipStreamNonBlocking) != hipSuccess) {
        set_err("hipStreamCreate failed");
        delete e;
        return nullptr;
    }
    e->ev_ok = true;
    for (auto &row : e->ev) for (auto &x : row) if (hipEventCreate(&x) != hipSuccess) e->ev_ok = false;
    for (auto &x : e->done) if (hipEventCreateWithFlags(&x, hipEventDisableTiming) != hipSuccess) e->ev_ok = false;
    for (auto &x : e->kdone) if (hipEventCreateWithFlags(&x, hipEventDisableTiming) != hipSuccess) e->ev_ok = false;
    for (auto &x : e->hdone) if (hipEventCreateWithFlags(&x, hipEventDisableTiming) != hipSuccess) e->ev_ok = false;
    for (auto &x : e->pdone) if (hipEventCreateWithFlags(&x, hipEventDisableTiming) != hipSuccess) e->ev_ok = false;
    for (auto &x : e->up) if (hipEventCreateWithFlags(&x, hipEventDisableTiming) != hipSuccess) e->ev_ok = false;
    return e;
}

extern "C" void scrappie_hip_engine_destroy(scrappie_hip_engine *e) {
    if (!e) return;
    if (e->tail_th_live) {
        { std::lock_guard<std::mutex> lk(e->tail_mu); e->tail_stop = true; }
        e->tail_cv.notify_all();
        e->tail_th.join();
        e->tail_th_live = false;
        if (e->tail_th2_live) { e->tail_th2.join(); e->tail_th2_live = false; }
    }
    for (auto &kv : e->tail_open) if (kv.second->done && !kv.second->rc) scrappie_hip_free_calls(kv.second->calls.data(), kv.second->calls.size());     /* never collected */
    e->tail_open.clear();
    if (e->tail) { scrappie_hip_engine_destroy(e->tail); e->tail = nullptr; }
    if (e->tail2) { scrappie_hip_engine_destroy(e->tail2); e->tail2 = nullptr; }
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    if (e->cstream) (void)hipStreamSynchronize(e->cstream);
    if (e->ustream) (void)hipStreamSynchronize(e->ustream);
    if (e->pstream) (void)hipStreamSynchronize(e->pstream);
    for (Model *m : e->models) { m->release(); delete m; }
    for (DBuf *b : {&e->d_meta[0], &e->d_meta[1], &e->d_signal[0], &e->d_signal[1], &e->d_act[0], &e->d_act[1], &e->d_act[2], &e->d_xaff, &e->d_E, &e->d_sums,
                    &e->d_tb, &e->d_tbend, &e->d_fstate, &e->d_fscore[0], &e->d_seq[0], &e->d_hp[0], &e->d_fscore[1], &e->d_seq[1], &e->d_hp[1],
                    &e->d_hstate, &e->d_gflag[0], &e->d_gflag[1], &e->d_vstate, &e->d_vflag, &e->d_Ealt, &e->d_sums_alt, &e->d_altoff, &e->d_act_alt, &e->d_trkoff, &e->d_bad[0], &e->d_bad[1], &e->d_conv[0], &e->d_conv[1],
                    &e->d_edge[0], &e->d_edge[1], &e->d_pos[0], &e->d_pos[1], &e->d_bases[0], &e->d_bases[1], &e->d_blen[0], &e->d_blen[1], &e->d_redo[0], &e->d_redo[1]}) b->release();
    for (int k = 0; k < 2; k++) for (HBuf *b : {&e->h_meta[k], &e->h_seq[k], &e->h_score[k], &e->h_hp[k], &e->h_pos[k], &e->h_bases[k], &e->h_blen[k], &e->h_redo[k]}) b->release();
    e->h_edge[0].release(); e->h_edge[1].release();
    e->h_sig[0].release(); e->h_sig[1].release(); e->h_err[0].release(); e->h_err[1].release(); e->h_bad[0].release(); e->h_bad[1].release();
    if (e->ev_ok) { for (auto &row : e->ev) for (auto &x : row) (void)hipEventDestroy(x); for (auto &x : e->done) (void)hipEventDestroy(x); for (auto &x : e->kdone) (void)hipEventDestroy(x); for (auto &x : e->hdone) (void)hipEventDestroy(x); for (auto &x : e->pdone) (void)hipEventDestroy(x); for (auto &x : e->up) (void)hipEventDestroy(x); }
    (void)hipStreamDestroy(e->stream);
    if (e->cstream) (void)hipStreamDestroy(e->cstream);
    if (e->ustream) (void)hipStreamDestroy(e->ustream);
    if (e->pstream) (void)hipStreamDestroy(e->pstream);
    delete e;
}

extern "C" scrappie_hip_params scrappie_hip_default_params(void) {
    scrappie_hip_params p;
    p.min_prob = 1e-5f; p.tempW = 1.0f; p.tempb = 1.0f;
    p.stay_pen = 0.0f; p.skip_pen = 0.0f; p.local_pen = 2.0f;
    p.use_slip = 0; p.homopolymer = HOMOPOLYMER_MEAN; p.want_pos = 0;
    return p;
}

/* -------------------- .scrm container (scrappie_amd/model.py) -------- */
static const HostMat *find_mat(const std::vector<std::pair<std::string, HostMat>> &ms, const char *nm) {
    for (auto &kv : ms) if (kv.first == nm) return &kv.second;
    return nullptr;
}

static int load_model_mem_one(scrappie_hip_engine *e, const char *name, const void *blob, size_t nbytes);
extern "C" int scrappie_hip_load_model_mem(scrappie_hip_engine *e, const char *name, const void *blob, size_t nbytes) {
    const int idx = load_model_mem_one(e, name, blob, nbytes);
    if (idx < 0 || e->is_tail) return idx;
    /* kept, so that the helper engine for chain-bound reads (scrappie_hip_basecall_batch) can be given the same models at the same indices */
    scrappie_hip_engine::Blob b{name, std::vector<unsigned char>((const unsigned char *)blob, (const unsigned char *)blob + nbytes), e->dbg_force_f32};
    bool found = false;
    for (auto &x : e->blobs) if (x.name == b.name) { x = b; found = true; }
    if (!found) e->blobs.push_back(b);
    if (e->tail || e->tail2) {      /* a helper may still be running a deferred ticket (its worker reads helper->models): wait until both are idle */
        std::unique_lock<std::mutex> lk(e->tail_mu);
        e->tail_cv.wait(lk, [&] { return e->tail_busy == 0 && e->tail_q.empty(); });
    }
    for (scrappie_hip_engine **tp : {&e->tail, &e->tail2}) if (*tp) {
        (*tp)->dbg_force_f32 = e->dbg_force_f32;
        if (load_model_mem_one(*tp, name, blob, nbytes) != idx) return set_err("model '%s' did not load at the same index on the helper engine", name);
    }
    return idx;
}
static int load_model_mem_one(scrappie_hip_engine *e, const char *name, const void *blob, size_t nbytes) {
    if (!e || !name || !blob) return set_err("load_model: null argument");
    const unsigned char *p = (const unsigned char *)blob, *end = p + nbytes;
    if (nbytes < 24 || memcmp(p, "SCRMDL01", 8) != 0) return set_err("model '%s': not a .scrm container", name);
    uint32_t hdr[4];
    memcpy(hdr, p + 8, 16);
    p += 24;
    std::vector<std::pair<std::string, HostMat>> mats;
    for (uint32_t i = 0; i < hdr[3]; i++) {
        if (p + 40 > end) return set_err("model '%s': truncated", name);
        char nm[33]; memcpy(nm, p, 32); nm[32] = 0;
        uint32_t nr, nc; memcpy(&nr, p + 32, 4); memcpy(&nc, p + 36, 4);
        p += 40;
        const size_t cnt = (size_t)nr * nc;
        if (p + cnt * 4 > end) return set_err("model '%s': truncated matrix %s", name, nm);
        HostMat hm; hm.nr = (int)nr; hm.nc = (int)nc; hm.v.resize(cnt);
        memcpy(hm.v.data(), p, cnt * 4);
        p += cnt * 4;
        mats.emplace_back(nm, std::move(hm));
    }
    (void)hipSetDevice(e->device);
    Model *m = new Model();
    m->name = name;
    m->arch = (int)hdr[0]; m->conv_act = (int)hdr[1]; m->stride = (int)hdr[2];
    const HostMat *cw = find_mat(mats, "conv_W"), *cb = find_mat(mats, "conv_b");
    const HostMat *fw = find_mat(mats, "ff_W"), *fb = find_mat(mats, "ff_b");
    if (m->arch == 3) {
        /* events bi-LSTM (networks.c:146-193): no convolution; lstm0..3 = F1, B1, F2, B2 */
        if (!fw || !fb) { delete m; return set_err("model '%s': missing ff matrices", name); }
        m->NS = fw->nc; m->S = fw->nr; m->stride = 1; m->WL = 0;
        const HostMat *i0 = find_mat(mats, "lstm0_iW");
        m->nfeat = i0 ? i0->nr : 0;
        m->F = 16;
        if (!i0 || m->nfeat < 1 || m->nfeat > 16 || m->S % 32 != 0 || m->S > 96 || (m->NS - 1) % 64 != 0) {
            delete m; return set_err("model '%s': unsupported events dims features=%d S=%d NS=%d", name, m->nfeat, m->S, m->NS);
        }
        for (int l = 0; l < 4; l++) {
            char nm[32];
            const HostMat *mi, *ms, *mb, *mpp;
            snprintf(nm, sizeof nm, "lstm%d_iW", l); mi = find_mat(mats, nm);
            snprintf(nm, sizeof nm, "lstm%d_sW", l); ms = find_mat(mats, nm);
            snprintf(nm, sizeof nm, "lstm%d_b", l); mb = find_mat(mats, nm);
            snprintf(nm, sizeof nm, "lstm%d_p", l); mpp = find_mat(mats, nm);
            const int I = (l < 2) ? m->nfeat : m->S;
            if (!mi || !ms || !mb || !mpp || mi->nr != I || mi->nc != 4 * m->S || ms->nr != m->S || ms->nc != 4 * m->S ||
                mb->nr * mb->nc != 4 * m->S || mpp->nr * mpp->nc != 3 * m->S) {
                m->release(); delete m;
                return set_err("model '%s': LSTM layer %d has wrong shapes", name, l);
            }
            for (const HostMat *x : {mi, ms}) if (!in_split_range(*x)) {
                const float mx = max_abs(*x);
                m->release(); delete m;
                return set_err("model '%s': LSTM layer %d has a weight of magnitude %g, outside the split products' range (< %g); "
                               "there is no exact-fp32 LSTM kernel", name, l, mx, (double)SH_W_LIMIT);
            }
            HostMat padded;                                  /* first level: K padded from 12 to 16 with zeros */
            const HostMat *src = mi;
            if (I % 16 != 0) {
                padded.nr = 16; padded.nc = mi->nc; padded.v.assign((size_t)16 * mi->nc, 0.0f);
                for (int c = 0; c < mi->nc; c++) for (int r = 0; r < I; r++) padded.v[(size_t)c * 16 + r] = mi->v[(size_t)c * I + r];
                src = &padded;
            }
            int mt, mtp;
            int mt_s;
            if (upload(m->iW[l], make_frags(*src, mt)) || upload(m->ib[l], make_bias_frags(*mb, mt)) ||
                upload(m->sW[l], make_frags(*ms, mt_s))) { m->release(); delete m; return -1; }
            HostMat padded32;                                /* ... and to one 32-wide k step for the split products of k_lstm_proj */
            const HostMat *psrc = src;
            if (src->nr % 32 != 0) {
                padded32.nr = 32; padded32.nc = mi->nc; padded32.v.assign((size_t)32 * mi->nc, 0.0f);
                for (int c = 0; c < mi->nc; c++) for (int r = 0; r < I; r++) padded32.v[(size_t)c * 32 + r] = mi->v[(size_t)c * I + r];
                psrc = &padded32;
            }
            if (upload_u32(m->iWp[l], make_piece_frags(*psrc)) || upload(m->ibs[l], scaled(make_bias_frags(*mb, mt), SH_OSCALE))) { m->release(); delete m; return -1; }
            if (upload_u32(m->sWp[l], make_piece_frags(*ms))) { m->release(); delete m; return -1; }
            mtp = 3 * m->S / 16;
            if (upload(m->lp[l], make_bias_frags(*mpp, mtp))) { m->release(); delete m; return -1; }
        }
    } else {
    if (!cw || !cb || !fw || !fb) { delete m; return set_err("model '%s': missing conv/ff matrices", name); }
    m->WL = cw->nr; m->F = cw->nc; m->NS = fw->nc; m->S = fw->nr;
    bool ok = (m->F % 16 == 0) && (m->S % 16 == 0) && m->stride > 0 && m->WL > 0;
    if (m->arch == 1) ok = ok && (m->F == m->S) && m->NS == 25;          /* residuals: layers.c:286-288 */
    if (m->arch == 0 || m->arch == 2) ok = ok && ((m->NS - 1) % 64 == 0); /* decode.c:132-138 */
    if (m->arch > 3) ok = false;
    if (!ok) { delete m; return set_err("model '%s': unsupported dims F=%d S=%d NS=%d WL=%d", name, m->F, m->S, m->NS, m->WL); }
    {   /* conv taps as [WL][F] so 4 consecutive filters load as one vector */
        std::vector<float> w((size_t)m->WL * m->F);
        for (int f = 0; f < m->F; f++) for (int t = 0; t < m->WL; t++) w[(size_t)t * m->F + f] = cw->v[(size_t)f * m->WL + t];
        if (upload(m->conv_W, w) || upload(m->conv_b, cb->v)) { m->release(); delete m; return -1; }
    }
    const int ngru = (m->arch == 2) ? 4 : 5;
    for (int l = 0; l < ngru; l++) {
        char nm[32];
        const HostMat *mi, *ms, *ms2, *mb;
        snprintf(nm, sizeof nm, "gru%d_iW", l); mi = find_mat(mats, nm);
        snprintf(nm, sizeof nm, "gru%d_sW", l); ms = find_mat(mats, nm);
        snprintf(nm, sizeof nm, "gru%d_sW2", l); ms2 = find_mat(mats, nm);
        snprintf(nm, sizeof nm, "gru%d_b", l); mb = find_mat(mats, nm);
        const int I = (m->arch == 2) ? (l < 2 ? m->F : m->S) : ((l == 0) ? m->F : m->S);
        if (!mi || !ms || !ms2 || !mb || mi->nr != I || mi->nc != 3 * m->S || ms->nr != m->S || ms->nc != 2 * m->S ||
            ms2->nr != m->S || ms2->nc != m->S || mb->nr * mb->nc != 3 * m->S) {
            m->release(); delete m;
            return set_err("model '%s': GRU layer %d has wrong shapes", name, l);
        }
        /* a weight the fp16 pieces cannot hold (|w| >= 255): this layer runs on the exact-fp32 kernels
         * (k_affine<.., F32> + k_gru_lanes / k_gru) instead -- slower, same results as the reference's fp32 */
        m->layer_f32[l] = !(in_split_range(*mi) && in_split_range(*ms) && in_split_range(*ms2)) || e->dbg_force_f32;
        if (m->layer_f32[l] && !(std::isfinite(max_abs(*mi)) && std::isfinite(max_abs(*ms)) && std::isfinite(max_abs(*ms2)))) {
            m->release(); delete m;
            return set_err("model '%s': GRU layer %d holds a non-finite weight", name, l);
        }
        if (m->layer_f32[l] && !e->dbg_force_f32)
            fprintf(stderr, "scrappie_hip: model '%s' GRU layer %d has |w| >= %g: outside the split products' operand range, using the exact-fp32 kernels for it\n",
                    name, l, (double)SH_W_LIMIT);
        int mt, mt_s;
        std::vector<float> ifr = make_frags(*mi, mt);
        if (upload(m->iW[l], ifr) || upload(m->ib[l], make_bias_frags(*mb, mt)) ||
            upload(m->sW[l], make_frags(*ms, mt_s)) || upload(m->sW2[l], make_frags(*ms2, mt_s))) { m->release(); delete m; return -1; }
        if ((mi->nr % 32 == 0 && (upload_u32(m->iWp[l], make_piece_frags(*mi)) || upload(m->ibs[l], scaled(make_bias_frags(*mb, mt), SH_OSCALE)))) ||
            (m->S % 32 == 0 && (upload_u32(m->sWp[l], make_piece_frags(*ms)) || upload_u32(m->sW2p[l], make_piece_frags(*ms2))))) { m->release(); delete m; return -1; }
#ifdef SH_EXPERIMENTS
        if (m->S == 96 && I == 96 && !m->layer_f32[l]) {      /* k_gru_proj32 */
            if (upload_u32(m->iWp32[l], make_piece_frags32(*mi)) || upload_u32(m->sWp32[l], make_piece_frags32(*ms)) ||
                upload_u32(m->sW2p32[l], make_piece_frags32(*ms2)) || upload(m->ib32[l], make_bias32(*mb))) { m->release(); delete m; return -1; }
            m->has32 = true;
        }
#endif
    }
    }
    if (m->arch == 2 || m->arch == 3) {   /* the two joining layers: misc/parse_raw.py:93-99,121-126; networks.c:167,180 */
        for (int k = 0; k < 2; k++) {
            char nm[32];
            const HostMat *wf, *wb, *bb;
            snprintf(nm, sizeof nm, "ff%d_Wf", k + 1); wf = find_mat(mats, nm);
            snprintf(nm, sizeof nm, "ff%d_Wb", k + 1); wb = find_mat(mats, nm);
            snprintf(nm, sizeof nm, "ff%d_b", k + 1); bb = find_mat(mats, nm);
            if (!wf || !wb || !bb || wf->nr != m->S || wb->nr != m->S || wf->nc != m->S || wb->nc != m->S || bb->nr * bb->nc != m->S) {
                m->release(); delete m;
                return set_err("model '%s': FF%d has wrong shapes (need S x S)", name, k + 1);
            }
            for (const HostMat *x : {wf, wb}) if (!in_split_range(*x)) {
                const float mx = max_abs(*x);
                m->release(); delete m;
                return set_err("model '%s': FF%d has a weight of magnitude %g, outside the split products' range (< %g); "
                               "there is no exact-fp32 kernel for this layer", name, k + 1, mx, (double)SH_W_LIMIT);
            }
            const int mt = (m->S + 15) / 16;
            /* as fp16 pieces (split products), the bias in accumulator units */
            if (upload_u32(m->ff2W[k][0], make_piece_frags(*wf)) || upload_u32(m->ff2W[k][1], make_piece_frags(*wb)) ||
                upload(m->ff2b[k], scaled(make_bias_frags(*bb, mt), SH_OSCALE))) { m->release(); delete m; return -1; }
        }
    }
    if (m->S % 32 == 0 && !in_split_range(*fw)) {
        const float mx = max_abs(*fw);
        m->release(); delete m;
        return set_err("model '%s': the output layer has a weight of magnitude %g, outside the split products' range (< %g); "
                       "there is no exact-fp32 kernel for this layer", name, mx, (double)SH_W_LIMIT);
    }
    if (upload(m->ffW, make_frags(*fw, m->ff_mtiles)) || upload(m->ffb, make_bias_frags(*fb, m->ff_mtiles))) { m->release(); delete m; return -1; }
    if (m->S % 32 == 0 && (upload_u32(m->ffWp, make_piece_frags(*fw)) || upload(m->ffbs, scaled(make_bias_frags(*fb, m->ff_mtiles), SH_OSCALE)))) { m->release(); delete m; return -1; }
    if (m->arch == 3) {
        m->min_samples = 2;                       /* lstm_forward needs two columns (layers.c:697) */
    } else {
    /* conv geometry: layers.c:169-207 */
    ShConvGeom &g = m->geom;
    g = conv_geom(m->WL, m->stride, m->F);
    /* below this the reference's edge arithmetic under/overflows (layers.c:227-231)
     * and gru_forward needs two columns (layers.c:400) */
    m->min_samples = (size_t)(g.shiftX + 2 * g.nstepX + g.WL);
    if (m->min_samples < (size_t)(g.st + 1)) m->min_samples = (size_t)(g.st + 1);
    }
    std::lock_guard<std::mutex> lk(e->mu);
    for (size_t i = 0; i < e->models.size(); i++)
        if (e->models[i]->name == name) { e->models[i]->release(); delete e->models[i]; e->models[i] = m; return (int)i; }
    e->models.push_back(m);
    return (int)e->models.size() - 1;
}

extern "C" int scrappie_hip_load_model(scrappie_hip_engine *e, const char *name, const char *path) {
    if (!path) return set_err("load_model: null path");
    FILE *fh = fopen(path, "rb");
    if (!fh) return set_err("cannot open model file %s", path);
    fseek(fh, 0, SEEK_END);
    long sz = ftell(fh);
    fseek(fh, 0, SEEK_SET);
    std::vector<unsigned char> buf((size_t)std::max(0L, sz));
    const size_t got = fread(buf.data(), 1, buf.size(), fh);
    fclose(fh);
    if (got != buf.size()) return set_err("short read on %s", path);
    return scrappie_hip_load_model_mem(e, name, buf.data(), buf.size());
}

extern "C" int scrappie_hip_find_model(scrappie_hip_engine *e, const char *name) {
    if (!e || !name) return -1;
    std::lock_guard<std::mutex> lk(e->mu);
    for (size_t i = 0; i < e->models.size(); i++) if (e->models[i]->name == name) return (int)i;
    return -1;
}

static Model *get_model(scrappie_hip_engine *e, int model) {
    if (!e || model < 0 || (size_t)model >= e->models.size()) { set_err("invalid model handle %d", model); return nullptr; }
    return e->models[model];
}

extern "C" size_t scrappie_hip_min_samples(scrappie_hip_engine *e, int model) {
    Model *m = get_model(e, model);
    return m ? m->min_samples : 0;
}
extern "C" int scrappie_hip_model_stride(scrappie_hip_engine *e, int model) {
    Model *m = get_model(e, model);
    return m ? m->stride : -1;
}
extern "C" void scrappie_hip_set_profiling(scrappie_hip_engine *e, int on) { if (e) e->profiling = on != 0; }
static int resolve_spans(scrappie_hip_engine *e, int slot) {
    /* all events of `slot` have completed (caller waited on its done event or drained the stream) */
    scrappie_hip_timing &tm = e->slot_timing[slot];
    float dbg_wait = 0.f, dbg_lead = 0.f;
    float *fields[] = {&tm.conv_ms, &tm.affine_ms, &tm.gru_ms, &tm.ff_ms, &tm.decode_ms, &tm.backtrace_ms, &tm.total_ms, &tm.fused_ms, &tm.stitch_ms, &dbg_wait, &dbg_lead};
    for (auto &sp : e->spans[slot]) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, e->ev[slot][sp.i], e->ev[slot][sp.j]));
        *fields[sp.field] += ms;
    }
    e->spans[slot].clear();
    if (tun().host_stamp) fprintf(stderr, "host stamp: main stream waited %.2f ms for the prologue; the convolution had ended %.2f ms before the main stream got there (negative: after)\n", dbg_wait, dbg_lead);
    e->timing = tm;
    return 0;
}

/* timing of the launch group most recently collected (or, if none was collected
 * since, of the most recent run once the stream has drained) */
extern "C" int scrappie_hip_get_timing(scrappie_hip_engine *e, scrappie_hip_timing *t) {
    if (!e || !t) return -1;
    (void)hipSetDevice(e->device);
    if (!e->spans[e->cur].empty() && !e->pending[e->cur]) {
        HIPCHK(hipStreamSynchronize(e->stream));
        if (resolve_spans(e, e->cur)) return -1;
    }
    *t = e->timing;
    return 0;
}
extern "C" void scrappie_hip_set_max_launch_reads(scrappie_hip_engine *e, size_t n) { if (e && n >= 16) e->max_launch_reads = n; }      /* (the helpers take the engine's settings with every call) */
extern "C" void scrappie_hip_set_max_launch_blocks(scrappie_hip_engine *e, size_t n) { if (e) e->max_launch_blocks = n; }
extern "C" void *scrappie_hip_device_alloc(scrappie_hip_engine *e, size_t nbytes) {
    if (!e) return nullptr;
    (void)hipSetDevice(e->device);
    void *p = nullptr;
    if (hipMalloc(&p, nbytes) != hipSuccess) { set_err("hipMalloc(%zu) failed", nbytes); return nullptr; }
    return p;
}
extern "C" void scrappie_hip_device_free(scrappie_hip_engine *e, void *dptr) {
    if (e && dptr) { (void)hipSetDevice(e->device); (void)hipFree(dptr); }
}
extern "C" int scrappie_hip_memcpy_h2d(scrappie_hip_engine *e, void *dst, const void *src, size_t nbytes) {
    if (!e) return -1;
    (void)hipSetDevice(e->device);
    HIPCHK(hipMemcpy(dst, src, nbytes, hipMemcpyHostToDevice));
    return 0;
}
extern "C" int scrappie_hip_synchronize(scrappie_hip_engine *e) {
    if (!e) return -1;
    (void)hipSetDevice(e->device);
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}

/* The recurrent kernels' lane schedule, exposed for tests and introspection (host only, no device
 * needed): lanes_per_wg = 1 (k_gru_proj) or 2 (k_gru_split, k_lstm_lanes; = scrappie_hip_gru_schedule).
 * lane_off needs lanes_per_wg * ncu + 1 ints, seg takes cap rows of
 * {tile, first step, end step, 0}.  Returns the number of segments (even if > cap). */
extern "C" long scrappie_hip_lane_schedule(const int *tile_T, size_t ntile, int ncu, int lanes_per_wg, int *nwg, int *capacity,
                                           int *lane_off, int *seg, size_t cap) {
    if (!tile_T || ncu < 1 || lanes_per_wg < 1 || lanes_per_wg > 2) return -1;
    ShGruSchedule sc;
    sh_lane_schedule(tile_T, ntile, ncu, lanes_per_wg, sc);
    if (nwg) *nwg = sc.nwg;
    if (capacity) *capacity = sc.capacity;
    if (lane_off) memcpy(lane_off, sc.lane_off.data(), sc.lane_off.size() * sizeof(int));
    if (seg) for (size_t i = 0; i < sc.seg.size() && i < cap; i++) memcpy(seg + 4 * i, &sc.seg[i], 16);
    return (long)sc.seg.size();
}

extern "C" long scrappie_hip_gru_schedule(const int *tile_T, size_t ntile, int ncu, int *nwg, int *capacity,
                                          int *lane_off, int *seg, size_t cap) {
    return scrappie_hip_lane_schedule(tile_T, ntile, ncu, 2, nwg, capacity, lane_off, seg, cap);
}

/* The decoder's pieces (sh_sched.h), host only: seg takes cap rows of {tile, first block,
 * end block, 0}, in workgroup order.  Returns the number of pieces (even if > cap). */
extern "C" long scrappie_hip_decoder_pieces(const int *tile_T, size_t ntile, int ncu, int *seg, size_t cap) {
    if (!tile_T || ncu < 1) return -1;
    std::vector<ShGruSeg> v;
    sh_piece_schedule(tile_T, ntile, ncu, v);
    if (seg) for (size_t i = 0; i < v.size() && i < cap; i++) memcpy(seg + 4 * i, &v[i], 16);
    return (long)v.size();
}

/* Cut a list of reads (input order kept) into launch groups of at most max_reads reads and at most
 * max_blocks column blocks (16 reads x 1 block; what the device arena is proportional to).  Tiles are
 * formed from reads sorted by length inside a group, so a group's column blocks are bounded by
 * sum(T)/16 + max(T).  Host only.  starts takes cap group start indices; returns the number of groups
 * (even if > cap), or -1 when a single read alone exceeds max_blocks. */
extern "C" long scrappie_hip_plan_groups(const uint32_t *lengths, size_t n, int stride, size_t max_reads, size_t max_blocks,
                                         size_t *starts, size_t cap) {
    if ((!lengths && n) || stride < 1 || max_reads < 1) return -1;
    long ng = 0;
    size_t cnt = 0;
    unsigned long long sumT = 0, maxT = 0;
    for (size_t i = 0; i < n; i++) {
        const unsigned long long T = ((unsigned long long)lengths[i] + stride - 1) / stride;
        if (max_blocks && T / 16 + T + 1 > max_blocks) return -1;
        const unsigned long long ns = sumT + T, nm = std::max(maxT, T);
        if (cnt == 0 || cnt >= max_reads || (max_blocks && ns / 16 + nm + 1 > max_blocks)) {
            if (starts && (size_t)ng < cap) starts[ng] = i;
            ng++;
            cnt = 0; sumT = 0; maxT = 0;
        }
        cnt++; sumT += T; maxT = std::max(maxT, T);
    }
    return ng;
}

/* S1 inside the decoder (k_ff_viterbi): the posterior of a basecall is never written.  Whenever somebody wants to see
 * it (scrappie_hip_posterior, the decoder-input hook) or the shape is not the 4^5 + 1 states over 96 units the kernel
 * is built for, the two-kernel form runs instead -- with identical bits. */
static bool decoder_fused(const scrappie_hip_engine *e, const Model *m) {
    return m->NS > 25 && !e->alt_prob && m->NS == 1025 && m->S == 96 && !tun().ff_separate && !e->dbg_ff_separate;
}

/* device bytes one column block costs across the arena (activations x3, gate inputs where they exist, the posterior
 * where it is written, traceback, per-slot result buffers x2, signals x2): what bounds a launch group on a 288 GB part */
static size_t bytes_per_block(const Model *m, bool posterior) {
    const size_t S = (size_t)m->S, F = (size_t)m->F, w = std::max(S, F);
    /* activation buffers: the convolution's output per slot (2) + the layers' ping-pong partner; the bi-directional stacks
     * (raw_r94, events) keep a third layer buffer */
    size_t b = ((m->arch == 2 || m->arch == 3) ? 4 : 3) * w * 64 + 128;
    if (posterior) b += (size_t)m->ff_mtiles * 1024;
    if (m->arch == 3 || F != S || S % 32 || S / 16 > 6) b += (size_t)(m->arch == 3 ? 4 : 3) * S * 64;   /* gate inputs in HBM */
    if (m->NS > 25) b += (size_t)((m->NS - 1) / 4) * 64;     /* transducer traceback: one byte per state */
    else b += 16 * 4 * 4;
    /* per slot: path + position (4 + 4 bytes per read and block), side rows of the homopolymer correction (20), bases (5) */
    b += 16 * 2 * (4 + 4 + 20 + 5) + 2 * 16 * 4 * (size_t)std::max(m->stride, 1) * (m->arch == 3 ? (size_t)m->nfeat : 1);
    return b;
}

/* A launch group as large as the arena allows: a group lasts at least as long as its longest tile's serial chain
 * (blocks x layers), so mixed-length reads fill the device only when the group's blocks per lane reach the longest
 * tile (profiles/r3_mixed_rate_*.txt: 3000 reads of U{1000..40000} samples 6.4e8 samples/s, 16000 reads 1.44e9). */
static size_t launch_block_cap(scrappie_hip_engine *e, const Model *m) {
    if (e->max_launch_blocks) return e->max_launch_blocks;
    return (size_t)(e->mem_frac * (double)e->total_mem) / bytes_per_block(m, !decoder_fused(e, m));
}

/* ------------------------------------------------------------------ */
/* launch-group construction                                            */
/* ------------------------------------------------------------------ */
struct MetaPtrs { ShMeta md; const long long *seq_off, *hp_off, *bases_off; ShGruLanes lanes, lanes1; ShGruPairs pairs, pairs2; const ShGruSegD *vseg; };

static int build_group(scrappie_hip_engine *e, Model *m, const uint64_t *offsets, const uint32_t *lengths,
                       size_t n, bool hp_on, MetaPtrs &mp) {
    LaunchGroup &lg = e->lgs[e->cur];
    lg.valid = false;
    lg.n = n; lg.hp_on = hp_on;
    lg.ntile = (n + 15) / 16; lg.npad = lg.ntile * 16;
    const int st = m->stride;
    std::vector<int> T(n);
    for (size_t i = 0; i < n; i++)
        T[i] = (lengths[i] >= m->min_samples) ? (int)((lengths[i] + st - 1) / st) : 0;
    lg.order.assign(lg.npad, -1);
    std::vector<int> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return T[a] > T[b]; });
    for (size_t i = 0; i < n; i++) lg.order[i] = idx[i];
    lg.rT.assign(lg.npad, 0); lg.rN.assign(lg.npad, 0);
    lg.seq_off.assign(lg.npad, 0); lg.hp_off.assign(lg.npad, 0); lg.bases_off.assign(lg.npad, 0);
    long long nbases = 0;
    /* bases a read can give: k of the first k-mer + at most k per later path entry (k <= 5; CRF: one per block), + 1 */
    const long long per_entry = (m->arch == 1) ? 1 : 5;
    std::vector<unsigned long long> sig_off(lg.npad, 0);
    std::vector<int> tile_T(lg.ntile, 0);
    std::vector<long long> tile_boff(lg.ntile, 0);
    long long ncb = 0, nseq = 0, nhp = 0;
    for (size_t i = 0; i < lg.npad; i++) {
        const int o = lg.order[i];
        if (o >= 0 && T[o] > 0) {
            lg.rT[i] = T[o]; lg.rN[i] = (int)lengths[o]; sig_off[i] = offsets[o];
        }
        lg.hp_off[i] = nhp; nhp += lg.rT[i];
        lg.bases_off[i] = nbases; nbases += lg.rT[i] ? ((per_entry * ((long long)lg.rT[i] + 1) + 16 + 15) & ~15ll) : 0;
        tile_T[i >> 4] = std::max(tile_T[i >> 4], lg.rT[i]);
    }
    for (size_t t = 0; t < lg.ntile; t++) { tile_boff[t] = ncb; ncb += tile_T[t]; }
    /* decoded paths: tile-interleaved, entry t of read b of a tile at tile base + t * SH_SEQ_STRIDE + b (sh_kernels.h) */
    for (size_t t = 0; t < lg.ntile; t++) {
        for (int b = 0; b < 16; b++) lg.seq_off[t * 16 + b] = nseq + b;
        nseq += tile_T[t] ? ((long long)tile_T[t] + 1) * SH_SEQ_STRIDE : 0;
    }
    lg.ncb = ncb; lg.nseq = nseq; lg.nhp = nhp; lg.nbases_cap = nbases;
    ShGruSchedule sched;
    sh_lane_schedule(tile_T.data(), lg.ntile, e->ncu, 2, sched, e->handover);   /* GRU and LSTM kernels: two lanes per workgroup */
    lg.gru_nwg = sched.nwg;
    ShGruSchedule sched1;
    sh_lane_schedule(tile_T.data(), lg.ntile, e->ncu, 1, sched1, e->handover);  /* projection + recurrence kernel: one lane per workgroup */
    lg.gru1_nwg = sched1.nwg;
    { long long nlive = 0; int tmax = 0; for (int t : tile_T) { nlive += t > 0; tmax = std::max(tmax, t); }
      /* One or two tiles per workgroup: a launch lasts its longest lane's steps, and a workgroup with one tile steps
       * SH_GRU_TWO_RATIO times faster than one with two (1.9 against 3.2 us per step with every CU busy, 1.35 against
       * 2.65 with a quarter of them idle).  Equal reads: half the steps per lane beat that.  Mixed lengths whose longest
       * tile sets both capacities do not (6000 reads of U{1000..40000} samples: 54 against 109 ms for five layers).
       * (k_gru_proj<.., 2> keeps two block counts in one register: tmax < 65536.) */
      const int w1 = sched1.wg_iter.empty() ? 0 : *std::max_element(sched1.wg_iter.begin(), sched1.wg_iter.end());
      const int w2 = sched.wg_iter.empty() ? 0 : *std::max_element(sched.wg_iter.begin(), sched.wg_iter.end());
      lg.gru_two = nlive > e->ncu && tmax < 65536 && (double)w2 * tun().gru_two_ratio < (double)w1;
      if (e->dbg_gru_tiles == 1) lg.gru_two = false;
      if (e->dbg_gru_tiles == 2 && tmax < 65536) lg.gru_two = true; }
    /* k_gru_proj32: tiles stepped two at a time (neighbours in the length order; a tile of 2^18 blocks or more alone: the
     * kernel addresses both tiles of a pair from one base with 32-bit byte offsets), the pairs laid over the lanes like tiles */
    std::vector<int> pair_tile, pair_T;
    for (size_t t = 0; t < lg.ntile; ) {
        const bool two = t + 1 < lg.ntile && tile_T[t] < (1 << 18) && tile_T[t + 1] > 0;
        pair_tile.push_back((int)t); pair_tile.push_back(two ? (int)t + 1 : -1);
        pair_T.push_back(std::max(tile_T[t], two ? tile_T[t + 1] : 0));
        t += two ? 2 : 1;
    }
    ShGruSchedule sched32;
    sh_lane_schedule(pair_T.data(), pair_T.size(), e->ncu, 1, sched32, e->handover);
    lg.gru32_nwg = sched32.nwg;
    ShGruSchedule sched32b;
    sh_lane_schedule(pair_T.data(), pair_T.size(), e->ncu, 2, sched32b, e->handover);
    lg.gru32x2_nwg = sched32b.nwg;
    std::vector<ShGruSeg> vseg;                  /* decoder: one piece of a tile per workgroup */
    sh_piece_schedule(tile_T.data(), lg.ntile, e->ncu, vseg, e->handover);
    lg.vit_nwg = (int)vseg.size();
    /* pack metadata: [sig_off u64 npad][seq_off i64 npad][hp_off i64 npad][tile_boff i64 ntile][rN i32 npad][rT i32 npad][tile_T i32 ntile] */
    const size_t b_u64 = lg.npad * 8, b_i32 = lg.npad * 4;
    const size_t b_loff = sched.lane_off.size() * 4, b_seg = sched.seg.size() * sizeof(ShGruSeg), b_wit = sched.wg_iter.size() * 4;
    const size_t b_vloff = 0, b_vseg = vseg.size() * sizeof(ShGruSeg);
    const size_t b_loff1 = sched1.lane_off.size() * 4, b_seg1 = sched1.seg.size() * sizeof(ShGruSeg);
    const size_t b_loff32 = sched32.lane_off.size() * 4, b_seg32 = sched32.seg.size() * sizeof(ShGruSeg), b_pt = pair_tile.size() * 4;
    const size_t b_loff32b = sched32b.lane_off.size() * 4, b_seg32b = sched32b.seg.size() * sizeof(ShGruSeg);
    const size_t total = 4 * b_u64 + lg.ntile * 8 + 2 * b_i32 + lg.ntile * 4 + 16 + b_seg + b_loff + b_wit + 16 + b_vseg + b_vloff + 16 + b_seg1 + b_loff1 +
                         16 + b_seg32 + b_loff32 + b_pt + 16 + b_seg32b + b_loff32b;
    if (e->h_meta[e->cur].ensure(total + 16) || e->d_meta[e->cur].ensure(total + 16)) return -1;
    char *h = e->h_meta[e->cur].as<char>();
    size_t o = 0;
    memcpy(h + o, sig_off.data(), b_u64); const size_t o_sig = o; o += b_u64;
    memcpy(h + o, lg.seq_off.data(), b_u64); const size_t o_seq = o; o += b_u64;
    memcpy(h + o, lg.hp_off.data(), b_u64); const size_t o_hp = o; o += b_u64;
    memcpy(h + o, lg.bases_off.data(), b_u64); const size_t o_bs = o; o += b_u64;
    memcpy(h + o, tile_boff.data(), lg.ntile * 8); const size_t o_tb = o; o += lg.ntile * 8;
    memcpy(h + o, lg.rN.data(), b_i32); const size_t o_n = o; o += b_i32;
    memcpy(h + o, lg.rT.data(), b_i32); const size_t o_t = o; o += b_i32;
    memcpy(h + o, tile_T.data(), lg.ntile * 4); const size_t o_tt = o; o += lg.ntile * 4;
    o = (o + 15) & ~(size_t)15;
    memcpy(h + o, sched.seg.data(), b_seg); const size_t o_seg = o; o += b_seg;
    memcpy(h + o, sched.lane_off.data(), b_loff); const size_t o_loff = o; o += b_loff;
    memcpy(h + o, sched.wg_iter.data(), b_wit); const size_t o_wit = o; o += b_wit;
    o = (o + 15) & ~(size_t)15;
    memcpy(h + o, vseg.data(), b_vseg); const size_t o_vseg = o; o += b_vseg;
    o = (o + 15) & ~(size_t)15;
    memcpy(h + o, sched1.seg.data(), b_seg1); const size_t o_seg1 = o; o += b_seg1;
    memcpy(h + o, sched1.lane_off.data(), b_loff1); const size_t o_loff1 = o; o += b_loff1;
    o = (o + 15) & ~(size_t)15;
    memcpy(h + o, sched32.seg.data(), b_seg32); const size_t o_seg32 = o; o += b_seg32;
    memcpy(h + o, sched32.lane_off.data(), b_loff32); const size_t o_loff32 = o; o += b_loff32;
    memcpy(h + o, pair_tile.data(), b_pt); const size_t o_pt = o; o += b_pt;
    o = (o + 15) & ~(size_t)15;
    memcpy(h + o, sched32b.seg.data(), b_seg32b); const size_t o_seg32b = o; o += b_seg32b;
    memcpy(h + o, sched32b.lane_off.data(), b_loff32b); const size_t o_loff32b = o; o += b_loff32b;
    hipStream_t ps = e->ev_ok ? e->pstream : e->stream;      /* prologue stream: see run_pipeline */
    if (e->ev_ok) {
        const long long n16 = (long long)((total + 15) / 16);
        hipLaunchKernelGGL(k_upload_words, dim3((unsigned)std::min<long long>((n16 + 255) / 256, 64)), dim3(256), 0, ps, (const u32x4 *)h, e->d_meta[e->cur].as<u32x4>(), n16);
    } else HIPCHK(hipMemcpyAsync(e->d_meta[e->cur].p, h, total, hipMemcpyHostToDevice, ps));
    char *d = e->d_meta[e->cur].as<char>();
    mp.md.sig_off = (const unsigned long long *)(d + o_sig);
    mp.seq_off = (const long long *)(d + o_seq);
    mp.hp_off = (const long long *)(d + o_hp);
    mp.bases_off = (const long long *)(d + o_bs);
    mp.md.tile_boff = (const long long *)(d + o_tb);
    mp.md.rN = (const int *)(d + o_n);
    mp.md.rT = (const int *)(d + o_t);
    mp.md.tile_T = (const int *)(d + o_tt);
    static_assert(sizeof(ShGruSeg) == sizeof(ShGruSegD), "segment layout");
    mp.lanes.seg = (const ShGruSegD *)(d + o_seg);
    mp.lanes.lane_off = (const int *)(d + o_loff);
    mp.lanes.wg_iter = (const int *)(d + o_wit);
    mp.lanes.ntile = (int)lg.ntile;
    mp.vseg = (const ShGruSegD *)(d + o_vseg);
    if (e->d_hstate.ensure(std::max<size_t>(lg.ntile, 1) * 12 * 256 * 4) || e->d_gflag[e->cur].ensure((lg.ntile + 1) * 4)) return -1;
    mp.lanes.hstate = e->d_hstate.as<float>();
    mp.lanes.flag = e->d_gflag[e->cur].as<unsigned>();
    HIPCHK(hipMemsetAsync(e->d_gflag[e->cur].p, 0, (lg.ntile + 1) * 4, ps));
    if (e->d_bad[e->cur].ensure(lg.npad * 4)) return -1;
    HIPCHK(hipMemsetAsync(e->d_bad[e->cur].p, 0, lg.npad * 4, ps));
    mp.lanes1 = mp.lanes;
    mp.lanes1.seg = (const ShGruSegD *)(d + o_seg1);
    mp.lanes1.lane_off = (const int *)(d + o_loff1);
    mp.lanes1.wg_iter = nullptr;
    mp.pairs.seg = (const ShGruSegD *)(d + o_seg32);
    mp.pairs.lane_off = (const int *)(d + o_loff32);
    mp.pairs.pair_tile = (const int *)(d + o_pt);
    mp.pairs.hstate = mp.lanes.hstate;           /* 3072 floats per pair <= 2 x 6 x 256 per tile */
    mp.pairs.flag = mp.lanes.flag;
    mp.pairs.err = mp.lanes.flag + lg.ntile;
    mp.pairs2 = mp.pairs;
    mp.pairs2.seg = (const ShGruSegD *)(d + o_seg32b);
    mp.pairs2.lane_off = (const int *)(d + o_loff32b);
    return 0;
}

/* ------------------------------------------------------------------ */
/* kernel dispatch helpers                                              */
/* ------------------------------------------------------------------ */
template <int KQ>
static int launch_affine_k(hipStream_t s, const float *in, float *out, const float *wf, const unsigned *wp, const float *bf,
                           long long ncb, int mtiles) {
    const int mt = pick_mt(mtiles);
    long long gx = std::min<long long>((ncb + 3) / 4, 2048);
    if (gx < 1) gx = 1;
    dim3 grid((unsigned)gx, (unsigned)(mtiles / mt));
    switch (mt) {
    case 6: hipLaunchKernelGGL((k_affine<KQ, 6>), grid, dim3(256), 0, s, in, out, wf, wp, bf, ncb, mtiles); break;
    case 4: hipLaunchKernelGGL((k_affine<KQ, 4>), grid, dim3(256), 0, s, in, out, wf, wp, bf, ncb, mtiles); break;
    case 3: hipLaunchKernelGGL((k_affine<KQ, 3>), grid, dim3(256), 0, s, in, out, wf, wp, bf, ncb, mtiles); break;
    case 2: hipLaunchKernelGGL((k_affine<KQ, 2>), grid, dim3(256), 0, s, in, out, wf, wp, bf, ncb, mtiles); break;
    default: hipLaunchKernelGGL((k_affine<KQ, 1>), grid, dim3(256), 0, s, in, out, wf, wp, bf, ncb, mtiles); break;
    }
    return 0;
}

template <int KQ>
static int launch_affine_lds_k(hipStream_t s, const float *in, float *out, const float *wf, const unsigned *wp, const float *bf,
                               long long ncb, int mtiles) {
    constexpr int NB = SH_AFF_NB, NTH = SH_AFF_NTH;
    const size_t lds = ((size_t)mtiles * KQ * 256 + (size_t)mtiles * 256) * 4 + 16;
    static DevOnce attr_once;
    if (auto turn_ = attr_once.first())
        HIPCHK(hipFuncSetAttribute((const void *)k_affine_lds<KQ, NB, NTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    long long gx = std::min<long long>((ncb + (NTH / 64) * NB - 1) / ((NTH / 64) * NB), 256);
    if (gx < 1) gx = 1;
    /* column groups by fixed striding: measured 4 % faster here than the dynamic hand-out k_ff_lds uses */
    hipLaunchKernelGGL((k_affine_lds<KQ, NB, NTH>), dim3((unsigned)gx), dim3(NTH), lds, s, in, out, wf, wp, bf, ncb, mtiles);
    return 0;
}

/* exact-fp32 MFMAs on the fp32 fragments whatever K: the projection of a layer whose weights are outside the split
 * products' operand range (Model::layer_f32) */
template <int KQ>
static int launch_affine_f32_k(hipStream_t s, const float *in, float *out, const float *wf, const float *bf, long long ncb, int mtiles) {
    const int mt = (mtiles % 6 == 0) ? 6 : (mtiles % 2 == 0) ? 2 : 1;
    long long gx = std::min<long long>((ncb + 3) / 4, 2048);
    if (gx < 1) gx = 1;
    dim3 grid((unsigned)gx, (unsigned)(mtiles / mt));
    switch (mt) {
    case 6: hipLaunchKernelGGL((k_affine<KQ, 6, true>), grid, dim3(256), 0, s, in, out, wf, (const unsigned *)nullptr, bf, ncb, mtiles); break;
    case 2: hipLaunchKernelGGL((k_affine<KQ, 2, true>), grid, dim3(256), 0, s, in, out, wf, (const unsigned *)nullptr, bf, ncb, mtiles); break;
    default: hipLaunchKernelGGL((k_affine<KQ, 1, true>), grid, dim3(256), 0, s, in, out, wf, (const unsigned *)nullptr, bf, ncb, mtiles); break;
    }
    return 0;
}

static int launch_affine(hipStream_t s, int K, const float *in, float *out, const float *wf, const unsigned *wp, const float *bf_nat,
                         const float *bf_acc, long long ncb, int mtiles, bool force_f32 = false) {
    if (force_f32) {
        switch (K / 16) {
        case 2: return launch_affine_f32_k<2>(s, in, out, wf, bf_nat, ncb, mtiles);
        case 4: return launch_affine_f32_k<4>(s, in, out, wf, bf_nat, ncb, mtiles);
        case 6: return launch_affine_f32_k<6>(s, in, out, wf, bf_nat, ncb, mtiles);
        case 8: return launch_affine_f32_k<8>(s, in, out, wf, bf_nat, ncb, mtiles);
        default: break;            /* odd K / 16: the ordinary kernel is exact-fp32 already */
        }
    }
    if (K % 32 == 0 && (!wp || !bf_acc)) return set_err("layer weights were not cut into pieces (input size %d)", K);
    const float *bf = (K % 32 == 0) ? bf_acc : bf_nat;      /* split products start from the bias in accumulator units */
    /* big layers: LDS-resident weights, input read once */
    const size_t lds_need = ((size_t)mtiles * (K / 16) * 256 + (size_t)mtiles * 256) * 4;
    if (mtiles >= 12 && lds_need <= 150 * 1024 && ncb >= 4096 && !tun().affine_reg) {
        switch (K / 16) {
        case 1: return launch_affine_lds_k<1>(s, in, out, wf, wp, bf, ncb, mtiles);
        case 2: return launch_affine_lds_k<2>(s, in, out, wf, wp, bf, ncb, mtiles);
        case 4: return launch_affine_lds_k<4>(s, in, out, wf, wp, bf, ncb, mtiles);
        case 6: return launch_affine_lds_k<6>(s, in, out, wf, wp, bf, ncb, mtiles);
        default: break;
        }
    }
    switch (K / 16) {
    case 1: return launch_affine_k<1>(s, in, out, wf, wp, bf, ncb, mtiles);
    case 2: return launch_affine_k<2>(s, in, out, wf, wp, bf, ncb, mtiles);
    case 4: return launch_affine_k<4>(s, in, out, wf, wp, bf, ncb, mtiles);
    case 6: return launch_affine_k<6>(s, in, out, wf, wp, bf, ncb, mtiles);
    case 8: return launch_affine_k<8>(s, in, out, wf, wp, bf, ncb, mtiles);
    default: return set_err("unsupported layer input size %d (need 16, 32, 64, 96 or 128)", K);
    }
}

template <int KQ>
static int launch_affine2_k(hipStream_t s, const float *inF, const float *inB, float *out, const unsigned *wF, const unsigned *wB,
                            const float *bf, long long ncb, int mtiles) {
    long long gx = std::min<long long>((ncb + 3) / 4, 2048);
    if (gx < 1) gx = 1;
    /* S / 16 = 2, 4 or 6 m-tiles: two wave quartets per workgroup, each with half of them */
    const int mt = mtiles / 2;
    if (mt * 2 != mtiles || mt > 3) return set_err("unsupported joining layer of %d m-tiles", mtiles);
    dim3 grid((unsigned)gx);
    switch (mt) {
    case 3: hipLaunchKernelGGL((k_affine2_tanh<KQ, 3>), grid, dim3(512), 0, s, inF, inB, out, wF, wB, bf, ncb, mtiles); break;
    case 2: hipLaunchKernelGGL((k_affine2_tanh<KQ, 2>), grid, dim3(512), 0, s, inF, inB, out, wF, wB, bf, ncb, mtiles); break;
    default: hipLaunchKernelGGL((k_affine2_tanh<KQ, 1>), grid, dim3(512), 0, s, inF, inB, out, wF, wB, bf, ncb, mtiles); break;
    }
    return 0;
}

static int launch_affine2(hipStream_t s, int K, const float *inF, const float *inB, float *out, const unsigned *wF, const unsigned *wB,
                          const float *bf, long long ncb, int mtiles) {
    switch (K / 16) {
    case 2: return launch_affine2_k<2>(s, inF, inB, out, wF, wB, bf, ncb, mtiles);
    case 4: return launch_affine2_k<4>(s, inF, inB, out, wF, wB, bf, ncb, mtiles);
    case 6: return launch_affine2_k<6>(s, inF, inB, out, wF, wB, bf, ncb, mtiles);
    default: return set_err("unsupported bi-GRU size %d (need 32, 64 or 96)", K);
    }
}

static int launch_gru(hipStream_t s, int S, const float *xaff, float *out, const float *resid, const float *sW,
                      const float *sW2, const unsigned *sWp, const unsigned *sW2p, const ShMeta &md, int backward, size_t ntile, const ShGruLanes &lanes, int nwg,
                      bool force_f32 = false) {
    /* production path: two lanes per workgroup walking the lane schedule (sh_sched.h) */
    if (!tun().gru_single && !tun().gru_stamp && tun().gru_debug < 0 && S / 16 <= 6 && S % 32 == 0) {
        if (nwg <= 0) return 0;
        /* arrival counters of tiles cut between lanes: cleared before every launch */
        HIPCHK(hipMemsetAsync(lanes.flag, 0, (size_t)lanes.ntile * 4, s));
        dim3 lgrid((unsigned)nwg);
        const int NU = S / 16;
        const size_t lds = (size_t)2 * 2 * NU * 256 * 4;
        const bool stamp = tun().gru_lanes_stamp;
        static unsigned long long *ldbg = nullptr;
        static int lcalls = 0;
        if (stamp && !ldbg) (void)hipMalloc(&ldbg, 4096 * 16 * 8 * 8);
        const bool f32_env = tun().gru_f32 || force_f32;
        if (!stamp && !f32_env) {                  /* production: split products */
            const size_t plds = (size_t)2 * 2 * (NU / 2) * 2 * 64 * 4 * 4;
            switch (NU) {
            case 2: hipLaunchKernelGGL((k_gru_split<2>), lgrid, dim3(256), plds, s, xaff, out, resid, sWp, sW2p, md, backward, lanes); break;
            case 4: hipLaunchKernelGGL((k_gru_split<4>), lgrid, dim3(512), plds, s, xaff, out, resid, sWp, sW2p, md, backward, lanes); break;
            default: hipLaunchKernelGGL((k_gru_split<6>), lgrid, dim3(768), plds, s, xaff, out, resid, sWp, sW2p, md, backward, lanes); break;
            }
            return 0;
        }
        /* SH_GRU_F32 / SH_GRU_LANES_STAMP: the exact-fp32 MFMA kernel (same schedule), kept as the reference the
         * split products were measured against */
        switch (NU) {
        case 2: hipLaunchKernelGGL((k_gru_lanes<2, false>), lgrid, dim3(256), lds, s, xaff, out, resid, sW, sW2, md, backward, lanes, (unsigned long long *)nullptr); break;
        case 4: hipLaunchKernelGGL((k_gru_lanes<4, false>), lgrid, dim3(512), lds, s, xaff, out, resid, sW, sW2, md, backward, lanes, (unsigned long long *)nullptr); break;
        case 6:
#ifdef SH_EXPERIMENTS
            if (stamp) hipLaunchKernelGGL((k_gru_lanes<6, true>), lgrid, dim3(768), lds, s, xaff, out, resid, sW, sW2, md, backward, lanes, ldbg);
            else
#endif
            hipLaunchKernelGGL((k_gru_lanes<6, false>), lgrid, dim3(768), lds, s, xaff, out, resid, sW, sW2, md, backward, lanes, (unsigned long long *)nullptr);
            break;
        default: break;
        }
        if (stamp && NU == 6 && ++lcalls == 7) {
            (void)hipStreamSynchronize(s);
            std::vector<unsigned long long> h((size_t)nwg * 12 * 8);
            (void)hipMemcpy(h.data(), ldbg, h.size() * 8, hipMemcpyDeviceToHost);
            for (size_t g : {(size_t)nwg / 2}) for (int w = 0; w < 12; w++) {
                unsigned long long *d = &h[(g * 12 + w) * 8];
                fprintf(stderr, "gru lanes stamp wg %zu wave %2d: phase1 %.0f bar %.0f phase2 %.0f bar %.0f cycles/step (%llu steps)\n",
                        g, w, d[0] / (double)d[4], d[1] / (double)d[4], d[2] / (double)d[4], d[3] / (double)d[4], d[4]);
            }
        }
        return 0;
    }
    /* other sizes, and the instrumented single-tile kernel (SH_GRU_SINGLE / SH_GRU_STAMP / SH_GRU_DEBUG) */
    dim3 grid((unsigned)ntile);
    const int NUx = S / 16;
    if (tun().gru_debug >= 0) backward |= tun().gru_debug << 8;
    static unsigned long long *dbgbuf = nullptr;
    if (tun().gru_stamp && !dbgbuf) { (void)hipMalloc(&dbgbuf, 4096 * 8 * 8 * 8); }
    if (dbgbuf) {
        static int calls = 0;
        if (calls == 7) {   /* dump the stamps of an earlier launch */
            (void)hipStreamSynchronize(s);
            std::vector<unsigned long long> h(ntile * NUx * 8);
            (void)hipMemcpy(h.data(), dbgbuf, h.size() * 8, hipMemcpyDeviceToHost);
            { unsigned long long first_end = ~0ull, mn = ~0ull, mx = 0; size_t late = 0;
              for (size_t tl = 0; tl < ntile; tl++) { first_end = std::min(first_end, h[tl * NUx * 8 + 7]); mn = std::min(mn, h[tl * NUx * 8 + 6]); mx = std::max(mx, h[tl * NUx * 8 + 7]); }
              for (size_t tl = 0; tl < ntile; tl++) if (h[tl * NUx * 8 + 6] >= first_end) late++;
              fprintf(stderr, "residency: %zu tiles, %zu started after the first one finished; span %.1f us\n", ntile, late, (mx - mn) / 100.0); }
            for (size_t tl : {size_t(0)}) for (int w = 0; w < 1; w++) {
                unsigned long long *d = &h[(tl * NUx + w) * 8];
                fprintf(stderr, "stamp tile %zu wave %d: rgemm %.0f zgemm+valu %.0f bar1 %.0f gemm2+valu %.0f bar2 %.0f (cycles/step)\n", tl, w, d[0] / (double)d[5], d[1] / (double)d[5], d[2] / (double)d[5], d[3] / (double)d[5], d[4] / (double)d[5]);
            }
        }
        calls++;
    }
    const int NU = S / 16;
    const size_t lds = 0;
    switch (NU) {
    case 2: hipLaunchKernelGGL((k_gru<2>), grid, dim3(128), lds, s, xaff, out, resid, sW, sW2, md, backward, dbgbuf); break;
    case 4: hipLaunchKernelGGL((k_gru<4>), grid, dim3(256), lds, s, xaff, out, resid, sW, sW2, md, backward, dbgbuf); break;
    case 6: hipLaunchKernelGGL((k_gru<6>), grid, dim3(384), lds, s, xaff, out, resid, sW, sW2, md, backward, dbgbuf); break;
    case 8: {
        static DevOnce attr_once;
        if (auto turn_ = attr_once.first()) HIPCHK(hipFuncSetAttribute((const void *)k_gru<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_gru<8>), grid, dim3(512), lds, s, xaff, out, resid, sW, sW2, md, backward, dbgbuf);
        break;
    }
    default: return set_err("unsupported GRU size %d (need 32, 64, 96 or 128)", S);
    }
    return 0;
}

#ifndef SH_FFL_NB
#define SH_FFL_NB 3
#endif
#ifndef SH_FFL_NTH
#define SH_FFL_NTH 512
#endif
/* m-tiles per LDS-resident group of the S1 weight fragments (also fixes the order in which row sums are added) */
static int ff_mtp(int KQ, int mtiles) {
    const size_t per_mt = ((size_t)KQ * 256 + 256) * 4;
    int mt_fit = (int)((156 * 1024) / per_mt);
    mt_fit -= mt_fit % SH_SUM_GROUP;            /* whole row-sum groups per part */
    return std::min(mt_fit, (mtiles + SH_SUM_GROUP - 1) / SH_SUM_GROUP * SH_SUM_GROUP);
}

template <int KQ>
static int launch_ff_lds_k(hipStream_t s, const float *in, float *E, float *sums, const unsigned *wf, const float *bf,
                           long long ncb, int mtiles, int NS, float in_div, float out_div, int ncu) {
    constexpr int NB = SH_FFL_NB, NTH = SH_FFL_NTH;
    const int mtp = ff_mtp(KQ, mtiles);
    const size_t lds = (size_t)mtp * ((size_t)KQ * 256 + 256) * 4 + 16;
    static DevOnce attr_once;
    if (auto turn_ = attr_once.first()) {
        HIPCHK(hipFuncSetAttribute((const void *)k_ff_lds<KQ, NB, NTH, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(hipFuncSetAttribute((const void *)k_ff_lds<KQ, NB, NTH, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    long long gx = std::min<long long>((ncb + (NTH / 64) * NB - 1) / ((NTH / 64) * NB), ncu);
    if (gx < 1) gx = 1;
    if (out_div != 1.0f) hipLaunchKernelGGL((k_ff_lds<KQ, NB, NTH, true>), dim3((unsigned)gx), dim3(NTH), lds, s, in, E, sums, wf, bf, ncb, mtiles, mtp, NS, in_div, out_div, (unsigned long long *)nullptr);
    else {
        const bool stamp = tun().ff_stamp;
        static unsigned long long *fdbg = nullptr;
        if (stamp && !fdbg) (void)hipMalloc(&fdbg, 16 * 8 * 8);
        hipLaunchKernelGGL((k_ff_lds<KQ, NB, NTH, false>), dim3((unsigned)gx), dim3(NTH), lds, s, in, E, sums, wf, bf, ncb, mtiles, mtp, NS, in_div, out_div, fdbg);
        if (stamp) {
            (void)hipStreamSynchronize(s);
            unsigned long long h[NTH / 64 * 8];
            (void)hipMemcpy(h, fdbg, sizeof h, hipMemcpyDeviceToHost);
            for (int w = 0; w < NTH / 64; w++)
                fprintf(stderr, "ff stamp wave %d: fill %llu  B-load %llu  tiles %llu  sums %llu cycles; %llu m-tiles -> %.0f cycles per m-tile\n", w, h[w * 8], h[w * 8 + 1], h[w * 8 + 2], h[w * 8 + 3], h[w * 8 + 4], (double)h[w * 8 + 2] / (double)h[w * 8 + 4]);
        }
    }
    return 0;
}

static int launch_ff(hipStream_t s, int S, const float *in, float *E, float *sums, const unsigned *wf, const float *bf,
                     long long ncb, int mtiles, int NS, float in_div, float out_div, int ncu) {
    /* large batches: weight fragments in LDS (k_ff_lds); small ones: one wave per column group streaming them from L2 */
    if (ncb >= 8192 && !tun().ff_reg) {
        switch (S / 16) {
        case 2: return launch_ff_lds_k<2>(s, in, E, sums, wf, bf, ncb, mtiles, NS, in_div, out_div, ncu);
        case 4: return launch_ff_lds_k<4>(s, in, E, sums, wf, bf, ncb, mtiles, NS, in_div, out_div, ncu);
        case 6: return launch_ff_lds_k<6>(s, in, E, sums, wf, bf, ncb, mtiles, NS, in_div, out_div, ncu);
        default: break;
        }
    }
    constexpr int NB = SH_FF_NB;
    const int mtp = ff_mtp(S / 16, mtiles);
    const bool dv = out_div != 1.0f;
    const long long gx = (ncb + 4 * NB - 1) / (4 * NB);
    dim3 grid((unsigned)std::max<long long>(gx, 1));
    switch (S / 16) {
    case 2: if (dv) hipLaunchKernelGGL((k_ff_exp<2, NB, true>), grid, dim3(256), 0, s, in, E, sums, wf, bf, ncb, mtiles, mtp, NS, in_div, out_div);
            else hipLaunchKernelGGL((k_ff_exp<2, NB, false>), grid, dim3(256), 0, s, in, E, sums, wf, bf, ncb, mtiles, mtp, NS, in_div, out_div);
            break;
    case 4: if (dv) hipLaunchKernelGGL((k_ff_exp<4, NB, true>), grid, dim3(256), 0, s, in, E, sums, wf, bf, ncb, mtiles, mtp, NS, in_div, out_div);
            else hipLaunchKernelGGL((k_ff_exp<4, NB, false>), grid, dim3(256), 0, s, in, E, sums, wf, bf, ncb, mtiles, mtp, NS, in_div, out_div);
            break;
    case 6: if (dv) hipLaunchKernelGGL((k_ff_exp<6, NB, true>), grid, dim3(256), 0, s, in, E, sums, wf, bf, ncb, mtiles, mtp, NS, in_div, out_div);
            else hipLaunchKernelGGL((k_ff_exp<6, NB, false>), grid, dim3(256), 0, s, in, E, sums, wf, bf, ncb, mtiles, mtp, NS, in_div, out_div);
            break;
    case 8: if (dv) hipLaunchKernelGGL((k_ff_exp<8, NB, true>), grid, dim3(256), 0, s, in, E, sums, wf, bf, ncb, mtiles, mtp, NS, in_div, out_div);
            else hipLaunchKernelGGL((k_ff_exp<8, NB, false>), grid, dim3(256), 0, s, in, E, sums, wf, bf, ncb, mtiles, mtp, NS, in_div, out_div);
            break;
    default: return set_err("unsupported size %d", S);
    }
    return 0;
}

/* projection + recurrence in one kernel (k_gru_proj): layer input [ncb][S/16][256] -> layer output, the gate
 * inputs never in HBM.  Needs the layer input as wide as the state (K == S) and S in {32, 64, 96}. */
static bool gru_proj_ok(int K, int S) { return K == S && S % 32 == 0 && S / 16 <= 6; }

/* Where a read's convolution windows end, for the layer that computes its own input (k_gru_conv, ShConvFuse::edge): the
 * index arithmetic of layers.c:209-241 as k_conv_mfma / k_conv_act evaluate it per column, evaluated once per read.
 * out = {mask of the last 32 columns without a regular window (bit j: column T - 1 - j), first column with a right-edge
 * partial window, that window's w, N}.  For a given (t - c0) % nstepC the columns with a regular window are a prefix, so
 * 64 columns are looked at; false if one without lies more than 32 columns from the end (the caller then runs the
 * convolution as its own kernel). */
static bool conv_edge_words(const ShConvGeom &g, int N, int T, int out[4]) {
    out[0] = out[1] = out[2] = out[3] = 0;
    if (T <= 0) return true;
    unsigned irr = 0;
    bool ok = true;
    for (int t = std::max(g.c0, T - 64); t < T; t++) {
        const int kk = (t - g.c0) / g.nstepC, ii = (t - g.c0) - kk * g.nstepC;
        if (!((kk + 1) * g.nstepX <= N - g.shiftX - ii * g.st)) {
            const int j = T - 1 - t;
            if (j >= 32) ok = false; else irr |= 1u << j;
        }
    }
    const int maxCol = (N - g.shiftX) / g.nstepX;
    const int rem = (N - g.shiftX) % g.nstepX;
    const int colR = g.c0 + g.nstepC * (maxCol - 1) + rem / g.st + 1;
    const int startR = g.st - (g.padL + N - g.WL) % g.st - 1;
    out[0] = (int)irr; out[1] = colR + startR / g.st; out[2] = startR; out[3] = N;
    return ok;
}

extern "C" int scrappie_hip_conv_edge_words(int WL, int st, int F, int N, int out[4]) {
    if (WL < 1 || st < 1 || N < 1 || !out) return -1;
    return conv_edge_words(conv_geom(WL, st, F), N, (N + st - 1) / st, out) ? 1 : 0;
}

#ifdef SH_EXPERIMENTS
static int launch_gru_conv(hipStream_t s, int kst, int act, float *out, const unsigned *iW, const float *ib, const unsigned *sW,
                           const unsigned *sW2, const ShMeta &md, int backward, const ShGruLanes &lanes1, int nwg1,
                           const ShGruLanes &lanes2, int nwg2, bool two, const ShConvFuse &cf) {
    const ShGruLanes &lanes = two ? lanes2 : lanes1;
    const int nwg = two ? nwg2 : nwg1;
    if (nwg <= 0) return 0;
    HIPCHK(hipMemsetAsync(lanes.flag, 0, (size_t)lanes.ntile * 4, s));
    const size_t lds = (two ? 2 : 1) * ((size_t)4 * 3 * 2 * 64 * 4 + (size_t)2 * 3 * 6 * 256) * 4;
    dim3 grid((unsigned)nwg);
#define CONVG1(NTv, KSTv, ACTv)                                                                                               \
    {                                                                                                                        \
        static DevOnce attr_once;                                                                                            \
        if (auto turn_ = attr_once.first(lds > 48 * 1024))                                                                            \
            HIPCHK(hipFuncSetAttribute((const void *)k_gru_conv<NTv, KSTv, ACTv>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        hipLaunchKernelGGL((k_gru_conv<NTv, KSTv, ACTv>), grid, dim3(768), lds, s, out, iW, ib, sW, sW2, md, backward, lanes, cf); \
    }
#define CONVG(NTv) { if (kst == 3) { if (act) CONVG1(NTv, 3, 1) else CONVG1(NTv, 3, 0) } else { if (act) CONVG1(NTv, 5, 1) else CONVG1(NTv, 5, 0) } }
    if (tun().proj_stamp && two && kst == 3 && !act) {       /* cycle stamps of one launch on stderr (tuning aid) */
        static unsigned long long *pdbg = nullptr;
        static int calls = 0;
        if (!pdbg) (void)hipMalloc(&pdbg, 1024 * 12 * 16 * 8);
        static DevOnce once;
        if (auto turn_ = once.first()) HIPCHK(hipFuncSetAttribute((const void *)k_gru_conv_stamp<2, 3, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL((k_gru_conv_stamp<2, 3, 0>), grid, dim3(768), lds, s, out, iW, ib, sW, sW2, md, backward, lanes, cf, pdbg);
        if (++calls == 3) {
            (void)hipStreamSynchronize(s);
            std::vector<unsigned long long> h((size_t)nwg * 12 * 16);
            (void)hipMemcpy(h.data(), pdbg, h.size() * 8, hipMemcpyDeviceToHost);
            for (int w = 0; w < 12; w++) {
                unsigned long long *d = &h[((size_t)(nwg / 2) * 12 + w) * 16];
                fprintf(stderr, "conv-layer stamp wave %2d (%s): A %.0f bar %.0f B %.0f bar %.0f cycles per double step (%llu steps); chunk+fetch part of B (projection) / reads+MFMA issue (recurrence) %.0f\n", w,
                        w < 6 ? "recurrence" : "projection", d[0] / (double)d[4], d[1] / (double)d[4], d[2] / (double)d[4], d[3] / (double)d[4], d[4], d[5] / (double)d[4]);
            }
        }
        return 0;
    }
    if (two) CONVG(2) else CONVG(1)
#undef CONVG
#undef CONVG1
    return 0;
}
#endif
static int launch_gru_proj(hipStream_t s, int S, const float *in, float *out, const float *resid, const unsigned *iW, const float *ib,
                           const unsigned *sW, const unsigned *sW2, const ShMeta &md, int backward, const ShGruLanes &lanes1, int nwg1,
                           const ShGruLanes &lanes2, int nwg2, bool two) {
    /* two tiles per workgroup (every wave steps both inside each barrier interval) as soon as there are more tiles
     * than workgroups; else one tile per workgroup, one workgroup per CU */
    const ShGruLanes &lanes = two ? lanes2 : lanes1;
    const int nwg = two ? nwg2 : nwg1;
    if (nwg <= 0) return 0;
    HIPCHK(hipMemsetAsync(lanes.flag, 0, (size_t)lanes.ntile * 4, s));
    const int NU = S / 16;
    const size_t lds = (two ? 2 : 1) * ((size_t)4 * (NU / 2) * 2 * 64 * 4 + (size_t)2 * 3 * NU * 256) * 4;
    dim3 grid((unsigned)nwg);
#define PROJ_LAUNCH1(NUv, NTv, RSv)                                                                                          \
    {                                                                                                                        \
        static DevOnce attr_once;                                                                                            \
        if (auto turn_ = attr_once.first(lds > 48 * 1024))                                                                            \
            HIPCHK(hipFuncSetAttribute((const void *)k_gru_proj<NUv, NTv, RSv>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        hipLaunchKernelGGL((k_gru_proj<NUv, NTv, RSv>), grid, dim3(128 * NUv), lds, s, in, out, resid, iW, ib, sW, sW2, md, backward, lanes); \
    }
#define PROJ_LAUNCH1R(NUv, NTv)                                                                                              \
    {                                                                                                                        \
        static DevOnce attr_once;                                                                                            \
        if (auto turn_ = attr_once.first(lds > 48 * 1024))                                                                   \
            HIPCHK(hipFuncSetAttribute((const void *)k_gru_proj_res<NUv, NTv>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        hipLaunchKernelGGL((k_gru_proj_res<NUv, NTv>), grid, dim3(128 * NUv), lds, s, in, out, resid, iW, ib, sW, sW2, md, backward, lanes); \
    }
#define PROJ_LAUNCH(NUv, NTv) { if (resid) PROJ_LAUNCH1R(NUv, NTv) else PROJ_LAUNCH1(NUv, NTv, false) }
#ifdef SH_EXPERIMENTS
    const bool stamp = tun().proj_stamp;     /* cycle stamps of one launch on stderr (tuning aid) */
    const bool free_run = SH_GRU_FREE_DEFAULT ? !tun().gru_barrier : tun().gru_free;
    if (free_run) {
        /* no s_barrier in the step loop: projection team free running on a ring of three blocks (k_gru_free) */
        const size_t flds = (two ? 2 : 1) * ((size_t)4 * (NU / 2) * 2 * 64 * 4 + (size_t)3 * 3 * NU * 256) * 4 + 64;
#define FREE_LAUNCH1(NUv, NTv, RSv, STv, DBG)                                                                                \
    {                                                                                                                        \
        static DevOnce attr_once;                                                                                            \
        if (auto turn_ = attr_once.first(flds > 48 * 1024))                                                                           \
            HIPCHK(hipFuncSetAttribute((const void *)k_gru_free<NUv, NTv, RSv, STv>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        hipLaunchKernelGGL((k_gru_free<NUv, NTv, RSv, STv>), grid, dim3(128 * NUv), flds, s, in, out, resid, iW, ib, sW, sW2, md, backward, lanes, DBG); \
    }
#define FREE_LAUNCH(NUv, NTv) { if (resid) FREE_LAUNCH1(NUv, NTv, true, false, (unsigned long long *)nullptr) else FREE_LAUNCH1(NUv, NTv, false, false, (unsigned long long *)nullptr) }
        if (stamp && NU == 6 && two && !resid) {
            static unsigned long long *fdbg = nullptr;
            static int fcalls = 0;
            if (!fdbg) (void)hipMalloc(&fdbg, 1024 * 12 * 16 * 8);
            FREE_LAUNCH1(6, 2, false, true, fdbg)
            if (++fcalls == 7) {
                (void)hipStreamSynchronize(s);
                std::vector<unsigned long long> h((size_t)nwg * 12 * 16);
                (void)hipMemcpy(h.data(), fdbg, h.size() * 8, hipMemcpyDeviceToHost);
                for (int w = 0; w < 12; w++) {
                    unsigned long long *d = &h[((size_t)(nwg / 2) * 12 + w) * 16];
                    if (w < 6) fprintf(stderr, "free stamp wave %2d (recurrence): work %.0f, waiting for gate inputs %.0f, for r*h of all waves %.0f, for h of all waves %.0f cycles per double step (%llu steps)\n",
                                       w, d[0] / (double)d[4], d[1] / (double)d[4], d[2] / (double)d[4], d[3] / (double)d[4], d[4]);
                    else fprintf(stderr, "free stamp wave %2d (projection): work %.0f, waiting for the column's pieces %.0f, for a free ring slot %.0f cycles per double step (%llu steps)\n",
                                 w, d[0] / (double)d[4], d[1] / (double)d[4], d[2] / (double)d[4], d[4]);
                }
            }
            return 0;
        }
        if (two) {
            switch (NU) {
            case 2: FREE_LAUNCH(2, 2) break;
            case 4: FREE_LAUNCH(4, 2) break;
            default: FREE_LAUNCH(6, 2) break;
            }
        } else {
            switch (NU) {
            case 2: FREE_LAUNCH(2, 1) break;
            case 4: FREE_LAUNCH(4, 1) break;
            default: FREE_LAUNCH(6, 1) break;
            }
        }
#undef FREE_LAUNCH1
#undef FREE_LAUNCH
        return 0;
    }
    if (stamp && NU == 6 && two && !resid) {
        static unsigned long long *pdbg = nullptr;
        static int calls = 0;
        if (!pdbg) (void)hipMalloc(&pdbg, 1024 * 12 * 16 * 8);
        static DevOnce once;
        if (auto turn_ = once.first()) HIPCHK(hipFuncSetAttribute((const void *)k_gru_proj<6, 2, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL((k_gru_proj<6, 2, false, true>), grid, dim3(768), lds, s, in, out, resid, iW, ib, sW, sW2, md, backward, lanes, pdbg);
        if (++calls == 7) {
            (void)hipStreamSynchronize(s);
            std::vector<unsigned long long> h((size_t)nwg * 12 * 16);
            (void)hipMemcpy(h.data(), pdbg, h.size() * 8, hipMemcpyDeviceToHost);
            for (int w = 0; w < 12; w++) {
                unsigned long long *d = &h[((size_t)(nwg / 2) * 12 + w) * 16];
                fprintf(stderr, "proj stamp wave %2d (%s): A %.0f bar %.0f B %.0f bar %.0f cycles per double step (%llu steps)", w, w < 6 ? "recurrence" : "projection",
                        d[0] / (double)d[4], d[1] / (double)d[4], d[2] / (double)d[4], d[3] / (double)d[4], d[4]);
                if (w < 6) fprintf(stderr, "; inside B: reads+MFMA issue %.0f, logistic z %.0f, tanh+blend %.0f, store+bookkeeping %.0f, cut+publish %.0f",
                                   d[5] / (double)d[4], d[6] / (double)d[4], d[7] / (double)d[4], d[8] / (double)d[4], d[9] / (double)d[4]);
                fprintf(stderr, "\n");
            }
        }
        return 0;
    }
#endif
    if (two) {
        switch (NU) {
        case 2: PROJ_LAUNCH(2, 2) break;
        case 4: PROJ_LAUNCH(4, 2) break;
        default: PROJ_LAUNCH(6, 2) break;
        }
    } else {
        switch (NU) {
        case 2: PROJ_LAUNCH(2, 1) break;
        case 4: PROJ_LAUNCH(4, 1) break;
        default: PROJ_LAUNCH(6, 1) break;
        }
    }
#undef PROJ_LAUNCH1
#undef PROJ_LAUNCH
    return 0;
}

#ifdef SH_EXPERIMENTS
#ifndef SH_GRU32_DEFAULT
#define SH_GRU32_DEFAULT 0        /* 1: recurrent layers of S = 96 run k_gru_proj32 unless SH_GRU16 is set; 0: k_gru_proj unless SH_GRU32 is set */
#endif
static int use_gru32(const scrappie_hip_engine *e) {      /* 0: 16-read tiles; 1: k_gru_proj32; 2: k_gru_proj32x2 (SH_GRU32=2) */
    if (e->dbg_gru32 >= 0) return e->dbg_gru32;
    if (SH_GRU32_DEFAULT ? tun().gru16 : !tun().gru32) return 0;
    const char *v = getenv("SH_GRU32");
    return (v && atoi(v) == 2) ? 2 : (SH_GRU32_DEFAULT ? SH_GRU32_DEFAULT : 1);
}
/* one recurrent layer of S = 96 on tiles of 32 reads (k_gru_proj32, sh_gru32.h) */
static int launch_gru_proj32x2(hipStream_t s, const float *in, float *out, bool resid, const unsigned *iW, const float *ib, const unsigned *sW,
                               const unsigned *sW2, const ShMeta &md, int backward, const ShGruPairs &pairs, int nwg, size_t ntile) {
    if (nwg <= 0) return 0;
    HIPCHK(hipMemsetAsync(pairs.flag, 0, ntile * 4, s));
    const size_t lds = (size_t)SH_G32X2_LDS_WORDS * 4;
    dim3 grid((unsigned)nwg);
#define G32X2_LAUNCH(RSv, STv, DBG)                                                                                           \
    {                                                                                                                        \
        static DevOnce attr_once;                                                                                            \
        if (auto turn_ = attr_once.first())                                                                                               \
            HIPCHK(hipFuncSetAttribute((const void *)k_gru_proj32x2<RSv, STv>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        hipLaunchKernelGGL((k_gru_proj32x2<RSv, STv>), grid, dim3(512), lds, s, in, out, iW, ib, sW, sW2, md, backward, pairs, DBG); \
    }
    if (tun().gru32_stamp && !resid) {
        static unsigned long long *pdbg = nullptr;
        static int calls = 0;
        if (!pdbg) (void)hipMalloc(&pdbg, 1024 * 8 * 16 * 8);
        G32X2_LAUNCH(false, true, pdbg)
        if (++calls == 7) {
            (void)hipStreamSynchronize(s);
            std::vector<unsigned long long> h((size_t)nwg * 8 * 16);
            (void)hipMemcpy(h.data(), pdbg, h.size() * 8, hipMemcpyDeviceToHost);
            static const char *role[8] = {"R0 chain", "R1 chain", "R2 chain", "C cand-proj", "G0 z/r", "G1 z/r", "G2 z/r", "L loader"};
            for (int w = 0; w < 8; w++) {
                unsigned long long *d = &h[((size_t)(nwg / 2) * 8 + w) * 16];
                fprintf(stderr, "gru32x2 stamp wave %d (%s): work %.0f bar %.0f cycles per interval (%llu intervals = one tile-step of 32 reads each)\n", w, role[w],
                        d[0] / (double)d[4], d[1] / (double)d[4], d[4]);
            }
        }
        return 0;
    }
    if (resid) G32X2_LAUNCH(true, false, (unsigned long long *)nullptr)
    else G32X2_LAUNCH(false, false, (unsigned long long *)nullptr)
#undef G32X2_LAUNCH
    return 0;
}
static int launch_gru_proj32(hipStream_t s, const float *in, float *out, bool resid, const unsigned *iW, const float *ib, const unsigned *sW,
                             const unsigned *sW2, const ShMeta &md, int backward, const ShGruPairs &pairs, int nwg, size_t ntile) {
    if (nwg <= 0) return 0;
    HIPCHK(hipMemsetAsync(pairs.flag, 0, ntile * 4, s));
    const size_t lds = (size_t)SH_G32_LDS_WORDS * 4;
    dim3 grid((unsigned)nwg);
#define G32_LAUNCH(RSv, STv, DBG)                                                                                             \
    {                                                                                                                        \
        static DevOnce attr_once;                                                                                            \
        if (auto turn_ = attr_once.first())                                                                                               \
            HIPCHK(hipFuncSetAttribute((const void *)k_gru_proj32<RSv, STv>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        hipLaunchKernelGGL((k_gru_proj32<RSv, STv>), grid, dim3(512), lds, s, in, out, iW, ib, sW, sW2, md, backward, pairs, DBG); \
    }
    if (tun().gru32_stamp && !resid) {       /* cycle stamps of one launch on stderr (tuning aid) */
        static unsigned long long *pdbg = nullptr;
        static int calls = 0;
        if (!pdbg) (void)hipMalloc(&pdbg, 1024 * 8 * 16 * 8);
        G32_LAUNCH(false, true, pdbg)
        if (++calls == 7) {
            (void)hipStreamSynchronize(s);
            std::vector<unsigned long long> h((size_t)nwg * 8 * 16);
            (void)hipMemcpy(h.data(), pdbg, h.size() * 8, hipMemcpyDeviceToHost);
            static const char *role[8] = {"R0 chain", "R1 chain", "R2 chain", "C cand-proj", "G0 z/r", "G1 z/r", "G2 z/r", "L loader"};
            for (int w = 0; w < 8; w++) {
                unsigned long long *d = &h[((size_t)(nwg / 2) * 8 + w) * 16];
                fprintf(stderr, "gru32 stamp wave %d (%s): A %.0f bar %.0f B %.0f bar %.0f cycles per step (%llu steps)", w, role[w],
                        d[0] / (double)d[4], d[1] / (double)d[4], d[2] / (double)d[4], d[3] / (double)d[4], d[4]);
                if (w < 3) fprintf(stderr, "; A: reads+r products %.0f, logistic*h %.0f, cut+write %.0f; B: reads+candidate products %.0f, z/tanh/blend %.0f, store %.0f, cut+write %.0f",
                                   d[5] / (double)d[4], d[6] / (double)d[4], d[7] / (double)d[4], d[8] / (double)d[4], d[9] / (double)d[4], d[10] / (double)d[4], d[11] / (double)d[4]);
                fprintf(stderr, "\n");
            }
        }
        return 0;
    }
    if (resid) G32_LAUNCH(true, false, (unsigned long long *)nullptr)
    else G32_LAUNCH(false, false, (unsigned long long *)nullptr)
#undef G32_LAUNCH
    return 0;
}

#endif
static int launch_lstm(hipStream_t s, int S, const float *xaff, float *out, const unsigned *sW, const float *pf,
                       const ShMeta &md, int backward, const ShGruLanes &lanes, int nwg) {
    if (nwg <= 0) return 0;
    HIPCHK(hipMemsetAsync(lanes.flag, 0, (size_t)lanes.ntile * 4, s));
    dim3 grid((unsigned)nwg);
    switch (S / 16) {
    case 2: hipLaunchKernelGGL((k_lstm_lanes<2>), grid, dim3(256), 0, s, xaff, out, sW, pf, md, backward, lanes); break;
    case 4: hipLaunchKernelGGL((k_lstm_lanes<4>), grid, dim3(512), 0, s, xaff, out, sW, pf, md, backward, lanes); break;
    case 6: hipLaunchKernelGGL((k_lstm_lanes<6>), grid, dim3(768), 0, s, xaff, out, sW, pf, md, backward, lanes); break;
    default: return set_err("unsupported LSTM size %d (need 32, 64 or 96)", S);
    }
    return 0;
}

/* projection + LSTM recurrence in one kernel (k_lstm_proj): needs the layer input as wide as the state */
static int launch_lstm_proj(hipStream_t s, int S, int I, const float *in, float *out, const unsigned *iW, const float *ib,
                            const unsigned *sW, const float *pf, const ShMeta &md, int backward, const ShGruLanes &lanes, int nwg) {
    if (nwg <= 0) return 0;
    HIPCHK(hipMemsetAsync(lanes.flag, 0, (size_t)lanes.ntile * 4, s));
    const int NU = S / 16;
    const size_t lds = ((size_t)4 * (NU / 2) * 2 * 64 * 4 + (size_t)2 * 4 * NU * 256 + (size_t)3 * NU * 256) * 4;
    dim3 grid((unsigned)nwg);
#define LP_LAUNCH1(NUv, NUIv)                                                                                                \
    {                                                                                                                        \
        static DevOnce attr_once;                                                                                            \
        if (auto turn_ = attr_once.first(lds > 48 * 1024))                                                                            \
            HIPCHK(hipFuncSetAttribute((const void *)k_lstm_proj<NUv, NUIv>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        hipLaunchKernelGGL((k_lstm_proj<NUv, NUIv>), grid, dim3(128 * NUv), lds, s, in, out, iW, ib, sW, pf, md, backward, lanes); \
    }
#define LP_LAUNCH(NUv) { if (I == S) LP_LAUNCH1(NUv, NUv) else LP_LAUNCH1(NUv, 1) }
    if (I != S && I != 16) return set_err("unsupported LSTM input width %d", I);
    switch (NU) {
    case 2: LP_LAUNCH(2) break;
    case 4: LP_LAUNCH(4) break;
    case 6: LP_LAUNCH(6) break;
    default: return set_err("unsupported LSTM size %d (need 32, 64 or 96)", S);
    }
#undef LP_LAUNCH
#undef LP_LAUNCH1
    return 0;
}

static size_t viterbi_lds_bytes(int NH) {
    const int nskip = NH / 16, nslip = std::max(NH / 64, 1);
    return (size_t)NH * 16 * 4 * 2 + (size_t)nskip * 16 * 8 + (size_t)nslip * 16 * 8 + 2 * 16 * 16 * 8;
}

static int launch_viterbi(hipStream_t s, int NH, const ShVitArgs &a, const ShMeta &md, size_t nwg) {
    const size_t lds = viterbi_lds_bytes(NH);
    if (nwg == 0) return 0;
    dim3 grid((unsigned)nwg);
#define VIT_CASE1(NTH, PPT, FIN, SLIP, SK0)                                                                    \
    {                                                                                                       \
        static DevOnce attr_once;                                                                           \
        if (auto turn_ = attr_once.first()) {                                                                                    \
            HIPCHK(hipFuncSetAttribute((const void *)k_viterbi<NTH, PPT, FIN, SLIP, SK0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        }                                                                                                   \
        hipLaunchKernelGGL((k_viterbi<NTH, PPT, FIN, SLIP, SK0>), grid, dim3(NTH), lds, s, a, md);               \
    }
#define VIT_CASE2(NTH, PPT, FIN, SLIP) { if (skip0) VIT_CASE1(NTH, PPT, FIN, SLIP, true) else VIT_CASE1(NTH, PPT, FIN, SLIP, false) }
#define VIT_CASE(NTH, PPT)                                                                                     \
    {                                                                                                       \
        if (fin && slip) VIT_CASE2(NTH, PPT, true, true)                                                    \
        else if (fin) VIT_CASE2(NTH, PPT, true, false)                                                      \
        else if (slip) VIT_CASE2(NTH, PPT, false, true)                                                     \
        else VIT_CASE2(NTH, PPT, false, false)                                                              \
    }
    /* the log-posterior transform is compiled in (exp values + sums in, log always) or out (final log-posterior in) */
    const bool fin = a.sums != nullptr, slip = a.use_slip != 0, skip0 = a.skip_pen == 0.0f;
    if (fin && !a.want_log) return set_err("decode: exp-value input implies log output");
    switch (NH) {
    case 64: VIT_CASE(256, 1) break;
    case 256: VIT_CASE(256, 4) break;
    case 1024:
#ifdef SH_EXPERIMENTS
        if (getenv("SH_VIT_1024")) { VIT_CASE(1024, 4) break; }      /* sixteen waves of four quads each (four waves per SIMD at 128 VGPRs): see DESIGN.md section 5 */
#endif
        VIT_CASE(512, 8) break;
    default: return set_err("unsupported transducer state count %d (need 4^3, 4^4 or 4^5 k-mers)", NH);
    }
#undef VIT_CASE1
#undef VIT_CASE2
#undef VIT_CASE
    return 0;
}

/* S1 inside the decoder.  Two teams of waves (k_ff_viterbi_teams: an S1 producer team, a decoder team, scores updated in place) wherever
 * the slip move is off; the eight do-everything waves of k_ff_viterbi with it, under SH_FV_SINGLE=1 and under the "fv_single" debug option
 * (the parity tests compare the two forms bit for bit). */
static int launch_ff_viterbi(hipStream_t s, const ShFfArgs &f, const ShVitArgs &a, const ShMeta &md, size_t nwg, bool single) {
    const size_t lds = (size_t)SH_FV_LDS_FLOATS * 4;
    if (nwg == 0) return 0;
    dim3 grid((unsigned)nwg);
    if (!a.use_slip && !single) {
        const size_t ldt = (size_t)SH_FVT_LDS_FLOATS * 4;
#define FVT_CASE(SK0, DIV)                                                                                        \
    {                                                                                                             \
        static DevOnce attr_once;                                                                                 \
        if (auto turn_ = attr_once.first())                                                                                    \
            HIPCHK(hipFuncSetAttribute((const void *)k_ff_viterbi_teams<SK0, DIV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldt)); \
        hipLaunchKernelGGL((k_ff_viterbi_teams<SK0, DIV>), grid, dim3(SH_FVT_NTH), ldt, s, f, a, md);             \
    }
        const bool skip0 = a.skip_pen == 0.0f, dv = f.out_div != 1.0f || f.in_div != 1.0f;
        if (skip0) { if (dv) FVT_CASE(true, true) else FVT_CASE(true, false) } else { if (dv) FVT_CASE(false, true) else FVT_CASE(false, false) }
#undef FVT_CASE
        return 0;
    }
#define FV_CASE(SLIP, SK0, DIV)                                                                                   \
    {                                                                                                             \
        static DevOnce attr_once;                                                                                 \
        if (auto turn_ = attr_once.first())                                                                                    \
            HIPCHK(hipFuncSetAttribute((const void *)k_ff_viterbi<SLIP, SK0, DIV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((k_ff_viterbi<SLIP, SK0, DIV>), grid, dim3(512), lds, s, f, a, md);                    \
    }
    const bool slip = a.use_slip != 0, skip0 = a.skip_pen == 0.0f, dv = f.out_div != 1.0f || f.in_div != 1.0f;
    if (slip) { if (skip0) { if (dv) FV_CASE(true, true, true) else FV_CASE(true, true, false) } else { if (dv) FV_CASE(true, false, true) else FV_CASE(true, false, false) } }
    else { if (skip0) { if (dv) FV_CASE(false, true, true) else FV_CASE(false, true, false) } else { if (dv) FV_CASE(false, false, true) else FV_CASE(false, false, false) } }
#undef FV_CASE
    return 0;
}

/* ------------------------------------------------------------------ */
/* the device pipeline                                                  */
/* ------------------------------------------------------------------ */
enum StopAt { STOP_NONE = 0, STOP_TRUNK = 1, STOP_POST = 2 };

struct RunOut {   /* where things are on the device after a run */
    const float *act = nullptr;   /* trunk output [ncb][S/16][256] */
    int act_units = 0;
    const float *E = nullptr;     /* transducer: exp values; rnnrf: normalised transitions */
    const float *sums = nullptr;
};

static int run_pipeline(scrappie_hip_engine *e, Model *m, const float *d_signal, const uint64_t *offsets,
                        const uint32_t *lengths, size_t n, const scrappie_hip_params *p, StopAt stop,
                        int trunk_upto, RunOut *ro) {
    (void)hipSetDevice(e->device);
    if (n == 0) return set_err("empty batch");
    hipStream_t s = e->stream;
    const bool transducer = (m->arch != 1);
    const bool hp_on = transducer && p->homopolymer == HOMOPOLYMER_MEAN && stop == STOP_NONE;
    /* take a free slot (a slot stays taken until scrappie_hip_collect picks it up) */
    {
        int slot = -1;
        for (int k = 0; k < 2; k++) { const int c = (e->cur + 1 + k) & 1; if (!e->pending[c]) { slot = c; break; } }
        if (slot < 0) return set_err("two launch groups are already in flight: call scrappie_hip_collect first");
        e->cur = slot;
    }
    const int slot = e->cur;
    MetaPtrs mp;
    const auto hs0 = std::chrono::steady_clock::now();
    if (build_group(e, m, offsets, lengths, n, hp_on, mp)) return -1;
    if (tun().host_stamp) fprintf(stderr, "host stamp: build_group %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - hs0).count());
    LaunchGroup &lg = e->lgs[slot];
    if (lg.ncb == 0) {
        lg.valid = true; lg.model = (int)(std::find(e->models.begin(), e->models.end(), m) - e->models.begin());
        if (stop == STOP_NONE) { if (e->ev_ok) HIPCHK(hipEventRecord(e->done[slot], s)); e->pending[slot] = true; if (!e->pending[slot ^ 1]) e->oldest = slot; }
        return 0;
    }
    const long long ncb = lg.ncb;
    const int S = m->S, F = m->F;
    const size_t act_bytes = (size_t)ncb * std::max(S, F) * 16 * 4;
    if (e->d_act[1].ensure(act_bytes)) return -1;
    /* gate inputs in HBM: only where projection and recurrence are separate kernels */
    bool any_f32 = false;
    for (bool b : m->layer_f32) any_f32 |= b;
    const bool need_xaff = m->arch == 3 || !gru_proj_ok(F, S) || tun().gru_separate || any_f32;
    if (need_xaff && e->d_xaff.ensure((size_t)ncb * (m->arch == 3 ? 4 : 3) * S * 16 * 4)) return -1;
    if ((m->arch == 2 || m->arch == 3) && e->d_act[2].ensure(act_bytes)) return -1;
    const bool prof = e->profiling && e->ev_ok;
    scrappie_hip_timing &tm = e->slot_timing[slot];
    if (prof) { memset(&tm, 0, sizeof tm); e->evn = 0; e->spans[slot].clear(); }
    int evslot[16] = {0};
    bool bt_on_cs = false;
    enum { F_CONV = 0, F_AFFINE, F_GRU, F_FF, F_DECODE, F_BACKTRACE, F_TOTAL, F_FUSED, F_STITCH };
#define EV(i) do { if (prof && e->evn < 48) { evslot[i] = e->evn++; HIPCHK(hipEventRecord(e->ev[slot][evslot[i]], s)); } } while (0)
#define ACC(field, i, j) do { if (prof) e->spans[slot].push_back({field, evslot[i], evslot[j]}); } while (0)

    /* Prologue on its own (low-priority) stream: metadata and flag clears (build_group) and the convolution.  Nothing in it
     * depends on the previous launch group, and k_conv_act is built to fit beside k_gru_proj's waves (32 of the 512 VGPRs
     * of a SIMD stay free next to three of them), so while group k walks its recurrent layers the convolution of group
     * k + 1 is already running; the main stream only waits for it.  Its output has a buffer per slot. */
    hipStream_t ps = e->ev_ok ? e->pstream : s;
#ifdef SH_EXPERIMENTS
    static const bool rnnrf_main = getenv("SH_RNNRF_CONV_MAIN") != nullptr;
#else
    const bool rnnrf_main = false;
#endif
    if (e->ev_ok && m->arch == 1 && rnnrf_main) {
        /* experiment switch: rnnrf's convolution on the main stream.  With the VALU form of the convolution that was the
         * better place for this model (a step of 19 ms, 80 % of it recurrent layers: 19.33 against 19.65 ms); with
         * k_conv_mfma the prologue stream wins there too (19.75 against 20.1 ms on one box) */
        HIPCHK(hipEventRecord(e->pdone[slot], ps)); HIPCHK(hipStreamWaitEvent(s, e->pdone[slot], 0));
        ps = s;
    }
    if (e->d_conv[slot].ensure(act_bytes)) return -1;
    float *abuf[3] = {e->d_conv[slot].as<float>(), e->d_act[1].as<float>(), e->d_act[2].as<float>()};
    /* SH_CONV_IN_LAYER=1 (experiment; identical results, measured slower: DESIGN.md section 5): the convolution inside the
     * first recurrent layer (k_gru_conv) where the whole path runs as a basecall of an rgrgr model of the shipped shape;
     * everywhere else (hooks that stop after a stage, other shapes, the two-kernel layer forms) it is a kernel of its own. */
    const int kst = (m->WL + 3) / 4;
    bool fuse_conv = false;
#ifdef SH_EXPERIMENTS
    fuse_conv = tun().conv_in_layer && !tun().conv_valu && m->arch == 0 && F == 96 && S == 96 && (kst == 3 || kst == 5) &&
                     stop == STOP_NONE && trunk_upto >= 5 && !tun().gru_separate && !m->layer_f32[0] &&
                     !(SH_GRU_FREE_DEFAULT ? !tun().gru_barrier : tun().gru_free);
    if (fuse_conv) {
        if (e->h_edge[slot].ensure(lg.npad * 16) || e->d_edge[slot].ensure(lg.npad * 16)) return -1;
        int *ew = e->h_edge[slot].as<int>();
        for (size_t i = 0; i < lg.npad && fuse_conv; i++) {
            const int o = lg.order[i];
            if (!conv_edge_words(m->geom, lg.rN[i], lg.rT[i], ew + 4 * i)) fuse_conv = false;
            if (o >= 0 && lg.rT[i] > 0 && offsets[o] + (uint64_t)lg.rN[i] >= ((uint64_t)1 << 30)) fuse_conv = false;    /* 32-bit sample indices in the kernel */
        }
    }
#endif
#define EVP(i) do { if (prof && e->evn < 48) { evslot[i] = e->evn++; HIPCHK(hipEventRecord(e->ev[slot][evslot[i]], ps)); } } while (0)
    EVP(0);
    if (m->arch == 3) {   /* events: the input already is the feature matrix (12 floats per event) */
        int maxT = 0;
        for (size_t i = 0; i < lg.npad; i += 16) maxT = std::max(maxT, lg.rT[i]);
        dim3 grid((unsigned)lg.ntile, (unsigned)std::min(64, (maxT + 3) / 4));
        hipLaunchKernelGGL(k_feat_in, grid, dim3(256), 0, ps, d_signal, mp.md, m->nfeat, abuf[0], ncb, e->d_bad[slot].as<unsigned>());
    } else if (fuse_conv) {
        /* the first recurrent layer computes the convolution itself (k_gru_conv): what it needs to know about each read's
         * right edge goes to the device with the rest of the prologue */
        hipLaunchKernelGGL(k_upload_words, dim3((unsigned)std::min<size_t>((lg.npad + 255) / 256, 256)), dim3(256), 0, ps,
                           (const u32x4 *)e->h_edge[slot].p, e->d_edge[slot].as<u32x4>(), (long long)lg.npad);
    } else {   /* C1 + A1 */
        const int tchunk = tun().conv_tchunk;
        int maxT = 0;
        for (size_t i = 0; i < lg.npad; i += 16) maxT = std::max(maxT, lg.rT[i]);   /* sorted: first read of a tile is longest */
        dim3 grid((unsigned)lg.ntile, (unsigned)std::min(65535, (maxT + tchunk - 1) / tchunk));   /* the kernel strides over y */
        /* something to run under (the other slot's group is in flight): the 48-register build; else the fast one */
        const bool bg = e->ev_ok && e->pending[slot ^ 1] && ps != s;
#ifdef SH_EXPERIMENTS
        static const int fake = getenv("SH_CONV_FAKE") ? atoi(getenv("SH_CONV_FAKE")) : 0;   /* experiment: 1 = a 3 GB fill instead of the convolution, 2 = nothing (results invalid) */
#else
        const int fake = 0;
#endif
        /* on the matrix pipe (k_conv_mfma) for the shapes of the shipped models: 96 filters, 11 or 19 taps */
        /* (one form per model, whichever stream it runs on: the two forms round differently, and a read's call must not depend
         * on whether its launch group had another one to run under) */
        const bool mfma_ok = !tun().conv_valu && F == 96 && (kst == 3 || kst == 5);
        const bool areg = kst == 3 && !bg;       /* taps in registers (70 VGPRs) when the kernel has the GPU to itself, from LDS (<= 56) beside k_gru_proj */
        const size_t lds = ((size_t)m->WL * F + F + (mfma_ok && !areg ? (size_t)6 * kst * 64 : 0) + 16 * ((size_t)(tchunk - 1) * m->stride + m->WL)) * 4;
#define CONV_ARGS grid, dim3(256), lds, ps, d_signal, mp.md, m->conv_W.as<float>(), m->conv_b.as<float>(), m->geom, abuf[0], tchunk, e->d_bad[slot].as<unsigned>()
#define CONV_LAUNCH(K, ACTv) hipLaunchKernelGGL((K<ACTv>), CONV_ARGS)
#define CONV_MFMA(K, ACTv) do { if (areg) hipLaunchKernelGGL((K<ACTv, 6, 3, true>), CONV_ARGS); else if (kst == 3) hipLaunchKernelGGL((K<ACTv, 6, 3, false>), CONV_ARGS); else hipLaunchKernelGGL((K<ACTv, 6, 5, false>), CONV_ARGS); } while (0)
#define CONV_MFMA_BG(K, ACTv) do { if (kst == 3) hipLaunchKernelGGL((K<ACTv, 6, 3, false>), CONV_ARGS); else hipLaunchKernelGGL((K<ACTv, 6, 5, false>), CONV_ARGS); } while (0)     /* (taps from LDS: areg is never set beside the layers) */
        if (fake == 1) HIPCHK(hipMemsetAsync(abuf[0], 0, act_bytes, ps));
        else if (fake == 2) {}
        else if (mfma_ok) {
            if (m->conv_act == 1) { if (bg) CONV_MFMA_BG(k_conv_mfma_bg, 1); else CONV_MFMA(k_conv_mfma, 1); }
            else { if (bg) CONV_MFMA_BG(k_conv_mfma_bg, 0); else CONV_MFMA(k_conv_mfma, 0); }
        }
        else if (m->conv_act == 1) { if (bg) CONV_LAUNCH(k_conv_act_bg, 1); else CONV_LAUNCH(k_conv_act, 1); }
        else { if (bg) CONV_LAUNCH(k_conv_act_bg, 0); else CONV_LAUNCH(k_conv_act, 0); }
#undef CONV_MFMA
#undef CONV_MFMA_BG
#undef CONV_ARGS
#undef CONV_LAUNCH
    }
    EVP(1);
#undef EVP
    ACC(F_CONV, 0, 1);
    if (tun().host_stamp) fprintf(stderr, "host stamp: prologue enqueued at %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - hs0).count());
    EV(14);
    if (e->ev_ok && ps != s) { HIPCHK(hipEventRecord(e->pdone[slot], ps)); HIPCHK(hipStreamWaitEvent(s, e->pdone[slot], 0)); }
    if (tun().host_stamp) fprintf(stderr, "host stamp: wait enqueued at %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - hs0).count());
    EV(13);                      /* the main stream's part of the group starts here */
    ACC(9, 14, 13); ACC(10, 1, 14);
    if (tun().helper_fence && e->ev_ok && e->pending[slot ^ 1]) HIPCHK(hipStreamWaitEvent(s, e->hdone[slot ^ 1], 0));
    int cur = 0;
    if (m->arch == 3) {
        /* events (networks.c:159-181): per level, forward and backward LSTM on the same input,
         * joined by feedforward2_tanh */
        for (int lvl = 0; lvl < 2 && lvl < trunk_upto; lvl++) {
            const int I = (lvl == 0) ? F : S;
            float *in = abuf[cur];
            float *hF = abuf[(cur + 1) % 3], *hB = abuf[(cur + 2) % 3];
            for (int dir = 0; dir < 2; dir++) {
                const int l = 2 * lvl + dir;
                EV(2);
                const bool one_kernel = (I == S || I == 16) && S % 32 == 0 && !tun().gru_separate;
                if (one_kernel) {          /* one kernel per direction (k_lstm_proj) */
                    EV(3);
                    if (launch_lstm_proj(s, S, I, in, dir ? hB : hF, m->iWp[l].as<unsigned>(), m->ibs[l].as<float>(), m->sWp[l].as<unsigned>(), m->lp[l].as<float>(),
                                         mp.md, dir, mp.lanes1, lg.gru1_nwg)) return -1;
                } else {
                if (launch_affine(s, I, in, e->d_xaff.as<float>(), m->iW[l].as<float>(), m->iWp[l].as<unsigned>(), m->ib[l].as<float>(), m->ibs[l].as<float>(), ncb, 4 * S / 16)) return -1;
                EV(3);
                if (launch_lstm(s, S, e->d_xaff.as<float>(), dir ? hB : hF, m->sWp[l].as<unsigned>(), m->lp[l].as<float>(), mp.md, dir, mp.lanes, lg.gru_nwg)) return -1;
                }
                EV(4);
                ACC(F_AFFINE, 2, 3);
                ACC(F_GRU, 3, 4);
                if (prof) {
                    const double af = 2.0 * I * 4 * S * 16.0 * (double)ncb, gf = 2.0 * 4 * S * S * 16.0 * (double)ncb;
                    tm.n_affine_launches++; tm.n_gru_launches++; tm.affine_flops += af; tm.gru_flops += gf;
                    if (one_kernel) { tm.n_fused_launches++; tm.fused_flops += af + gf; }
                }
                if (one_kernel) ACC(F_FUSED, 3, 4);
            }
            EV(2);
            if (launch_affine2(s, S, hF, hB, in, m->ff2W[lvl][0].as<unsigned>(), m->ff2W[lvl][1].as<unsigned>(), m->ff2b[lvl].as<float>(), ncb, S / 16)) return -1;
            EV(3);
            ACC(F_AFFINE, 2, 3);
            if (prof) tm.affine_flops += 2.0 * 2 * S * S * 16.0 * (double)ncb;
        }
    } else if (m->arch == 2) {
        /* N3 raw_r94 (networks.c:196-247): per level, forward and backward GRU on the same
         * input, joined by feedforward2_tanh */
        for (int lvl = 0; lvl < 2 && lvl < trunk_upto; lvl++) {
            const int I = (lvl == 0) ? F : S;
            float *in = abuf[cur];
            float *hF = abuf[(cur + 1) % 3], *hB = abuf[(cur + 2) % 3];
            for (int dir = 0; dir < 2; dir++) {
                const int l = 2 * lvl + dir;
                EV(2);
                const bool f32 = m->layer_f32[l];
                const bool one_kernel = gru_proj_ok(I, S) && !tun().gru_separate && !f32;
#ifdef SH_EXPERIMENTS
                if (one_kernel && use_gru32(e) && S == 96 && I == 96 && m->has32) {
                    EV(3);
                    if (use_gru32(e) == 2 ? launch_gru_proj32x2(s, in, dir ? hB : hF, false, m->iWp32[l].as<unsigned>(), m->ib32[l].as<float>(), m->sWp32[l].as<unsigned>(),
                                                                m->sW2p32[l].as<unsigned>(), mp.md, dir, mp.pairs2, lg.gru32x2_nwg, lg.ntile)
                                          : launch_gru_proj32(s, in, dir ? hB : hF, false, m->iWp32[l].as<unsigned>(), m->ib32[l].as<float>(), m->sWp32[l].as<unsigned>(),
                                                              m->sW2p32[l].as<unsigned>(), mp.md, dir, mp.pairs, lg.gru32_nwg, lg.ntile)) return -1;
                } else
#endif
                if (one_kernel) {           /* one kernel per direction (k_gru_proj) */
                    EV(3);
                    if (launch_gru_proj(s, S, in, dir ? hB : hF, nullptr, m->iWp[l].as<unsigned>(), m->ibs[l].as<float>(), m->sWp[l].as<unsigned>(),
                                        m->sW2p[l].as<unsigned>(), mp.md, dir, mp.lanes1, lg.gru1_nwg, mp.lanes, lg.gru_nwg, lg.gru_two)) return -1;
                } else {
                    if (e->d_xaff.ensure((size_t)ncb * 3 * S * 16 * 4)) return -1;
                    if (launch_affine(s, I, in, e->d_xaff.as<float>(), m->iW[l].as<float>(), m->iWp[l].as<unsigned>(), m->ib[l].as<float>(), m->ibs[l].as<float>(), ncb, 3 * S / 16, f32)) return -1;
                    EV(3);
                    if (launch_gru(s, S, e->d_xaff.as<float>(), dir ? hB : hF, nullptr, m->sW[l].as<float>(), m->sW2[l].as<float>(), m->sWp[l].as<unsigned>(), m->sW2p[l].as<unsigned>(), mp.md, dir, lg.ntile, mp.lanes, lg.gru_nwg, f32)) return -1;
                }
                EV(4);
                ACC(F_AFFINE, 2, 3);
                ACC(F_GRU, 3, 4);
                if (prof) {
                    const double af = 2.0 * I * 3 * S * 16.0 * (double)ncb, gf = 2.0 * 3 * S * S * 16.0 * (double)ncb;
                    tm.n_affine_launches++; tm.n_gru_launches++; tm.affine_flops += af; tm.gru_flops += gf;
                    if (one_kernel) { tm.n_fused_launches++; tm.fused_flops += af + gf; }
                }
                if (one_kernel) ACC(F_FUSED, 3, 4);
            }
            EV(2);
            if (launch_affine2(s, S, hF, hB, in, m->ff2W[lvl][0].as<unsigned>(), m->ff2W[lvl][1].as<unsigned>(), m->ff2b[lvl].as<float>(), ncb, S / 16)) return -1;
            EV(3);
            ACC(F_AFFINE, 2, 3);
            if (prof) tm.affine_flops += 2.0 * 2 * S * S * 16.0 * (double)ncb;
        }
    } else
    {
    /* rgrgr / rnnrf stacks: each layer is one kernel, projection team + recurrence team per workgroup
     * (k_gru_proj), when the layer input is as wide as the state; else projection and recurrence apart */
    for (int l = 0; l < 5 && l < trunk_upto; l++) {
        const int I = (l == 0) ? F : S;
        const bool sep_env = tun().gru_separate;      /* projection and recurrence as two kernels */
        const bool f32 = m->layer_f32[l];            /* weights outside the split products' range: exact-fp32 kernels */
        const bool one_kernel = !sep_env && gru_proj_ok(I, S) && !f32;
        EV(2);
#ifdef SH_EXPERIMENTS
        if (l == 0 && fuse_conv) {
            EV(3);
            ShConvFuse cf;
            cf.sig = d_signal; cf.W = m->conv_W.as<float>(); cf.bias = m->conv_b.as<float>(); cf.edge = e->d_edge[slot].as<int>();
            cf.bad = e->d_bad[slot].as<unsigned>(); cf.g = m->geom;
            if (launch_gru_conv(s, kst, m->conv_act, abuf[cur ^ 1], m->iWp[l].as<unsigned>(), m->ibs[l].as<float>(), m->sWp[l].as<unsigned>(),
                                m->sW2p[l].as<unsigned>(), mp.md, 1, mp.lanes1, lg.gru1_nwg, mp.lanes, lg.gru_nwg, lg.gru_two, cf)) return -1;
        } else if (one_kernel && use_gru32(e) && S == 96 && I == 96 && m->has32) {
            EV(3);
            if (use_gru32(e) == 2 ? launch_gru_proj32x2(s, abuf[cur], abuf[cur ^ 1], m->arch == 1, m->iWp32[l].as<unsigned>(), m->ib32[l].as<float>(), m->sWp32[l].as<unsigned>(),
                                                        m->sW2p32[l].as<unsigned>(), mp.md, (l % 2 == 0) ? 1 : 0, mp.pairs2, lg.gru32x2_nwg, lg.ntile)
                                  : launch_gru_proj32(s, abuf[cur], abuf[cur ^ 1], m->arch == 1, m->iWp32[l].as<unsigned>(), m->ib32[l].as<float>(), m->sWp32[l].as<unsigned>(),
                                                      m->sW2p32[l].as<unsigned>(), mp.md, (l % 2 == 0) ? 1 : 0, mp.pairs, lg.gru32_nwg, lg.ntile)) return -1;
        } else
#endif
        if (one_kernel) {
            EV(3);
            if (launch_gru_proj(s, S, abuf[cur], abuf[cur ^ 1], m->arch == 1 ? abuf[cur] : nullptr,
                                m->iWp[l].as<unsigned>(), m->ibs[l].as<float>(), m->sWp[l].as<unsigned>(), m->sW2p[l].as<unsigned>(), mp.md,
                                (l % 2 == 0) ? 1 : 0, mp.lanes1, lg.gru1_nwg, mp.lanes, lg.gru_nwg, lg.gru_two)) return -1;
        } else {
        if (launch_affine(s, I, abuf[cur], e->d_xaff.as<float>(), m->iW[l].as<float>(), m->iWp[l].as<unsigned>(), m->ib[l].as<float>(), m->ibs[l].as<float>(), ncb, 3 * S / 16, f32)) return -1;
        EV(3);
        if (launch_gru(s, S, e->d_xaff.as<float>(), abuf[cur ^ 1], m->arch == 1 ? abuf[cur] : nullptr,
                       m->sW[l].as<float>(), m->sW2[l].as<float>(), m->sWp[l].as<unsigned>(), m->sW2p[l].as<unsigned>(), mp.md, (l % 2 == 0) ? 1 : 0, lg.ntile, mp.lanes, lg.gru_nwg, f32)) return -1;
        }
        EV(4);
        ACC(F_AFFINE, 2, 3);
        ACC(F_GRU, 3, 4);
        if (prof) {
            const double af = 2.0 * I * 3 * S * 16.0 * (double)ncb, gf = 2.0 * 3 * S * S * 16.0 * (double)ncb;
            tm.n_affine_launches++; tm.n_gru_launches++;
            tm.affine_flops += af;
            tm.gru_flops += gf;
            if (one_kernel) { tm.n_fused_launches++; tm.fused_flops += af + gf; }
        }
        if (one_kernel) ACC(F_FUSED, 3, 4);
        cur ^= 1;
    }
    }
    HIPCHK(hipGetLastError());
    if (tun().host_stamp) fprintf(stderr, "host stamp: layers enqueued at %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - hs0).count());
    if (ro) { ro->act = abuf[cur]; ro->act_units = (trunk_upto == 0) ? F : S; }
    lg.model = (int)(std::find(e->models.begin(), e->models.end(), m) - e->models.begin());
    if (stop == STOP_TRUNK) { lg.valid = true; return 0; }

    /* what the output layer reads: the trunk's output -- or, under scrappie_hip_set_trunk_input, the caller's
     * activations (the network above has run in full either way).  Their chunk-layout image is built once per
     * launch-group shape and re-used. */
    const float *top = abuf[cur];
    if (e->alt_trunk) {
        std::vector<unsigned long long> aoff(lg.npad, ~0ull);
        uint64_t key = 1469598103934665603ull ^ (uint64_t)lg.model ^ ((uint64_t)S << 32);
        for (size_t i = 0; i < lg.npad; i++) {
            const int o = lg.order[i];
            if (o >= 0 && lg.rT[i] > 0) aoff[i] = e->trk_off[(size_t)o % e->trk_off.size()];
            key = (key ^ (uint64_t)(aoff[i] + 0x9e3779b97f4a7c15ull * (uint64_t)(lg.rT[i] + 1))) * 1099511628211ull;
        }
        if (!e->trk_valid || key != e->trk_key) {
            if (e->d_act_alt.ensure((size_t)ncb * S * 16 * 4) || e->d_trkoff.ensure(lg.npad * 8)) return -1;
            HIPCHK(hipMemcpyAsync(e->d_trkoff.p, aoff.data(), lg.npad * 8, hipMemcpyHostToDevice, s));
            HIPCHK(hipStreamSynchronize(s));          /* aoff is a local */
            int maxT = 0;
            for (size_t i = 0; i < lg.npad; i += 16) maxT = std::max(maxT, lg.rT[i]);
            hipLaunchKernelGGL(k_inject_trunk, dim3((unsigned)lg.ntile, (unsigned)std::min(maxT, 1024)), dim3(256), 0, s, e->alt_trunk,
                               e->d_trkoff.as<unsigned long long>(), mp.md, S, e->d_act_alt.as<float>());
            e->trk_key = key; e->trk_valid = true;
        }
        top = e->d_act_alt.as<float>();
    }

    const int mtiles = m->ff_mtiles;
    const bool fused = transducer && stop == STOP_NONE && decoder_fused(e, m);
    if (!fused && e->d_E.ensure((size_t)ncb * mtiles * 256 * 4)) return -1;
    if (e->d_seq[slot].ensure((size_t)std::max<long long>(lg.nseq, 1) * 4) || e->d_fscore[slot].ensure(lg.npad * 4)) return -1;
    if (transducer) {
        if (e->d_sums.ensure((size_t)ncb * 16 * 4)) return -1;
        EV(5);
        if (!fused && launch_ff(s, S, top, e->d_E.as<float>(), e->d_sums.as<float>(), m->ffWp.as<unsigned>(), m->ffbs.as<float>(),
                                ncb, mtiles, m->NS, p->tempW / p->tempb, p->tempb, e->ncu)) return -1;
        EV(6);
        ACC(F_FF, 5, 6);
        if (prof) tm.ff_flops += 2.0 * S * m->NS * 16.0 * (double)ncb;
        const float *E_use = e->d_E.as<float>(), *sums_use = e->d_sums.as<float>();
        if (e->alt_prob) {
            /* measurement / test hook: the decoder (and scrappie_hip_posterior) see the caller's probabilities
             * instead.  Their decoder image is built once per launch-group shape and re-used. */
            std::vector<unsigned long long> aoff(lg.npad, ~0ull);
            uint64_t key = 1469598103934665603ull ^ (uint64_t)lg.model;
            for (size_t i = 0; i < lg.npad; i++) {
                const int o = lg.order[i];
                if (o >= 0 && lg.rT[i] > 0) aoff[i] = e->alt_off[(size_t)o % e->alt_off.size()];
                key = (key ^ (uint64_t)(aoff[i] + 0x9e3779b97f4a7c15ull * (uint64_t)(lg.rT[i] + 1))) * 1099511628211ull;
            }
            if (!e->alt_valid || key != e->alt_key) {
                if (e->d_Ealt.ensure((size_t)ncb * mtiles * 256 * 4) || e->d_sums_alt.ensure((size_t)ncb * 16 * 4) || e->d_altoff.ensure(lg.npad * 8)) return -1;
                HIPCHK(hipMemcpyAsync(e->d_altoff.p, aoff.data(), lg.npad * 8, hipMemcpyHostToDevice, s));
                HIPCHK(hipStreamSynchronize(s));          /* aoff is a local */
                int maxT = 0;
                for (size_t i = 0; i < lg.npad; i += 16) maxT = std::max(maxT, lg.rT[i]);
                hipLaunchKernelGGL(k_inject_prob, dim3((unsigned)lg.ntile, (unsigned)std::min(maxT, 1024)), dim3(256), 0, s, e->alt_prob,
                                   e->d_altoff.as<unsigned long long>(), mp.md, m->NS, mtiles, e->d_Ealt.as<float>(), e->d_sums_alt.as<float>());
                e->alt_key = key; e->alt_valid = true;
            }
            E_use = e->d_Ealt.as<float>(); sums_use = e->d_sums_alt.as<float>();
        }
        if (ro) { ro->E = E_use; ro->sums = sums_use; }
        if (stop == STOP_POST) { HIPCHK(hipGetLastError()); lg.valid = true; return 0; }
        const int NH = m->NS - 1, NQ = NH / 4;
        if (e->d_tb.ensure((size_t)ncb * NQ * 16 * 4) || e->d_tbend.ensure((size_t)ncb * 16 * 4) || e->d_fstate.ensure(lg.npad * 4)) return -1;
        if (hp_on && e->d_hp[slot].ensure((size_t)std::max<long long>(lg.nhp, 1) * 5 * 4)) return -1;
        ShVitArgs va;
        va.E = E_use; va.sums = sums_use;
        va.strideT = (long long)mtiles * 256; va.strideQ = 64; va.strideB = 4;
        va.want_log = 1; va.min_prob = p->min_prob;
        va.stay_pen = p->stay_pen; va.skip_pen = p->skip_pen; va.local_pen = p->local_pen; va.use_slip = p->use_slip;
        va.tb = e->d_tb.as<unsigned>(); va.tb_end = e->d_tbend.as<int>();
        va.final_state = e->d_fstate.as<int>(); va.final_score = e->d_fscore[slot].as<float>();
        va.hp_side = hp_on ? e->d_hp[slot].as<float>() : nullptr; va.hp_off = mp.hp_off;
        va.dbg = nullptr;
        va.dump_final = e->dbg_dump_final ? 1 : 0;
        static unsigned long long *vdbg = nullptr;
        if (tun().vit_stamp) { if (!vdbg) (void)hipMalloc(&vdbg, 4096 * 16 * 8 * 8); va.dbg = vdbg; }
        /* more tiles than CUs: tiles are decoded in pieces that hand their state over through HBM (sh_sched.h) */
        if (e->d_vstate.ensure(std::max<size_t>(lg.ntile, 1) * ((size_t)NH * 16 + 32) * 4) || e->d_vflag.ensure(std::max<size_t>(lg.ntile, 1) * 4)) return -1;
        HIPCHK(hipMemsetAsync(e->d_vflag.p, 0, std::max<size_t>(lg.ntile, 1) * 4, s));
        va.seg = mp.vseg;
        va.vstate = e->d_vstate.as<float>(); va.flag = e->d_vflag.as<unsigned>(); va.err = e->d_gflag[slot].as<unsigned>() + lg.ntile;
        /* the other slot's traceback walk (on the copy stream) reads the buffers this decode overwrites */
        if (e->ev_ok && e->pending[slot ^ 1]) HIPCHK(hipStreamWaitEvent(s, e->done[slot ^ 1], 0));
        if (fused) {
            ShFfArgs fa;
            fa.in = top; fa.wpiece = m->ffWp.as<unsigned>(); fa.bfrag = m->ffbs.as<float>();
            fa.in_div = p->tempW / p->tempb; fa.out_div = p->tempb;
            va.E = nullptr; va.sums = nullptr;
            if (launch_ff_viterbi(s, fa, va, mp.md, (size_t)lg.vit_nwg, tun().fv_single || e->dbg_fv_single)) return -1;
        } else if (launch_viterbi(s, NH, va, mp.md, (size_t)lg.vit_nwg)) return -1;
        if (va.dbg) {
            (void)hipStreamSynchronize(s);
            std::vector<unsigned long long> h((size_t)std::max(lg.vit_nwg, 1) * 16 * 8);
            (void)hipMemcpy(h.data(), vdbg, h.size() * 8, hipMemcpyDeviceToHost);
            const int nwv = (fused && !va.use_slip && !(tun().fv_single || e->dbg_fv_single)) ? 12 : 8;       /* the two-team kernel stamps twelve waves (8-11: the S1 team) */
            for (int w = 0; w < nwv; w++) { unsigned long long *d = &h[((size_t)(lg.vit_nwg / 2) * nwv + w) * 8]; fprintf(stderr, "vit stamp wave %d: phaseB %.0f bar %.0f phaseC %.0f bar %.0f cycles/block\n", w, d[0] / (double)d[4], d[1] / (double)d[4], d[2] / (double)d[4], d[3] / (double)d[4]); }
        }
        EV(7);
        ACC(F_DECODE, 6, 7);
        if (tun().host_stamp) fprintf(stderr, "host stamp: decoder enqueued at %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - hs0).count());
        /* the traceback walk is a chain of dependent loads per read (latency, hardly any CUs): it runs on the
         * copy stream, under the next group's first kernels, in front of the result copies */
        bt_on_cs = e->ev_ok;
        if (bt_on_cs) { HIPCHK(hipEventRecord(e->kdone[slot], s)); HIPCHK(hipStreamWaitEvent(e->cstream, e->kdone[slot], 0)); }
        {
            hipStream_t bs = bt_on_cs ? e->cstream : s;
            if (prof && e->evn < 48) { evslot[10] = e->evn++; HIPCHK(hipEventRecord(e->ev[slot][evslot[10]], bs)); }
            hipLaunchKernelGGL(k_backtrace, dim3((unsigned)((lg.npad + 63) / 64)), dim3(64), 0, bs, e->d_tb.as<unsigned>(), e->d_tbend.as<int>(),
                               e->d_fstate.as<int>(), mp.md, mp.seq_off, e->d_seq[slot].as<int>(), (int)lg.npad, NQ, SH_SEQ_STRIDE);
            if (prof && e->evn < 48) { evslot[8] = e->evn++; HIPCHK(hipEventRecord(e->ev[slot][evslot[8]], bs)); }
        }
        ACC(F_BACKTRACE, 10, 8);
    } else {
        EV(5);
        if (launch_affine(s, S, top, e->d_E.as<float>(), m->ffW.as<float>(), m->ffWp.as<unsigned>(), m->ffb.as<float>(), m->ffbs.as<float>(), ncb, mtiles)) return -1;
        EV(6);
        ACC(F_FF, 5, 6);
        if (prof) tm.ff_flops += 2.0 * S * m->NS * 16.0 * (double)ncb;
        if (e->d_tb.ensure((size_t)ncb * 16 * 8)) return -1;         /* one byte per state (8 per read) and block */
        /* d_tb is shared by the two slots: a transducer group in the other slot may still be walking it
         * (k_backtrace on the copy stream) */
        if (e->ev_ok && e->pending[slot ^ 1]) HIPCHK(hipStreamWaitEvent(s, e->done[slot ^ 1], 0));
        hipLaunchKernelGGL(k_crf, dim3((unsigned)(lg.npad / 16)), dim3(128), 0, s, e->d_E.as<float>(), mp.md, e->d_tb.as<unsigned char>(),
                           mp.seq_off, e->d_seq[slot].as<int>(), e->d_fscore[slot].as<float>(), (int)lg.npad, SH_SEQ_STRIDE);
        EV(7);
        ACC(F_DECODE, 6, 7);
        if (ro) { ro->E = e->d_E.as<float>(); ro->sums = nullptr; }
        if (stop == STOP_POST) { HIPCHK(hipGetLastError()); lg.valid = true; return 0; }
    }
    HIPCHK(hipGetLastError());
    /* results -> pinned host buffers on the copy stream: the per-slot device buffers are not touched
     * again before this slot is collected, so the next group's kernels need not wait for PCIe */
    if (e->h_score[slot].ensure(lg.npad * 4)) return -1;
    if (e->h_err[slot].ensure(4) || e->h_bad[slot].ensure(lg.npad * 4)) return -1;
    EV(9);
    ACC(F_TOTAL, 13, 9);        /* (the convolution ran on the prologue stream, under the previous group) */
    hipStream_t cs = e->ev_ok ? e->cstream : s;
    if (e->ev_ok && !bt_on_cs) { HIPCHK(hipEventRecord(e->kdone[slot], s)); HIPCHK(hipStreamWaitEvent(cs, e->kdone[slot], 0)); }
    /* D2 + D3 on the device (k_stitch, behind the traceback walk on the copy stream): bases, not paths, go to the host */
    bool results_by_kernel = false;
    lg.dev_stitch = !tun().host_stitch;
    lg.dev_pos = lg.dev_stitch && p->want_pos != 0 && transducer;
    if (lg.dev_stitch) {
        const size_t nseq = (size_t)std::max<long long>(lg.nseq, 1), ncap = (size_t)std::max<long long>(lg.nbases_cap, 1);
        if (e->d_bases[slot].ensure(ncap) || e->d_blen[slot].ensure(lg.npad * 4) || e->d_redo[slot].ensure(lg.npad * 4) ||
            e->h_bases[slot].ensure(ncap) || e->h_blen[slot].ensure(lg.npad * 4) || e->h_redo[slot].ensure(lg.npad * 4)) return -1;
        if (lg.dev_pos && (e->d_pos[slot].ensure(nseq * 4 + 16) || e->h_pos[slot].ensure(nseq * 4 + 16))) return -1;
        ShStitchArgs sa;
        sa.seq = e->d_seq[slot].as<int>(); sa.seq_off = mp.seq_off;
        sa.hp = hp_on ? e->d_hp[slot].as<float>() : nullptr; sa.hp_off = mp.hp_off;
        sa.pos = lg.dev_pos ? e->d_pos[slot].as<int>() : nullptr;
        sa.bases = e->d_bases[slot].as<char>(); sa.bases_off = mp.bases_off;
        sa.blen = e->d_blen[slot].as<int>(); sa.redo = e->d_redo[slot].as<unsigned>();
        sa.npad = (int)lg.npad; sa.nstate = m->NS; sa.crf = transducer ? 0 : 1; sa.sstride = SH_SEQ_STRIDE;
        if (prof && e->evn < 48) { evslot[11] = e->evn++; HIPCHK(hipEventRecord(e->ev[slot][evslot[11]], cs)); }
        hipLaunchKernelGGL(k_stitch, dim3((unsigned)((lg.npad + 63) / 64)), dim3(64), 0, cs, sa, mp.md);
        if (prof && e->evn < 48) { evslot[12] = e->evn++; HIPCHK(hipEventRecord(e->ev[slot][evslot[12]], cs)); }
        ACC(F_STITCH, 11, 12);
        if (e->ev_ok) HIPCHK(hipEventRecord(e->hdone[slot], cs));
        if (tun().host_stamp) fprintf(stderr, "host stamp: stitch enqueued at %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - hs0).count());
        /* results into pinned host memory by the device itself (k_results_out): bases of exactly the called length, the
         * per-read words, the error word; pos[] (rarely wanted) likewise, whole */
        ShResultArgs ra;
        ra.d_bases = e->d_bases[slot].as<char>(); ra.h_bases = e->h_bases[slot].as<char>(); ra.bases_off = mp.bases_off;
        ra.d_blen = e->d_blen[slot].as<int>(); ra.h_blen = e->h_blen[slot].as<int>();
        ra.d_redo = e->d_redo[slot].as<unsigned>(); ra.h_redo = e->h_redo[slot].as<unsigned>();
        ra.d_score = e->d_fscore[slot].as<float>(); ra.h_score = e->h_score[slot].as<float>();
        ra.d_bad = e->d_bad[slot].as<unsigned>(); ra.h_bad = e->h_bad[slot].as<unsigned>();
        ra.d_err = e->d_gflag[slot].as<unsigned>() + lg.ntile; ra.h_err = e->h_err[slot].as<unsigned>();
        ra.npad = (int)lg.npad;
        hipLaunchKernelGGL(k_results_out, dim3((unsigned)((lg.npad + 3) / 4)), dim3(256), 0, cs, ra);
        if (lg.dev_pos) {
            const long long n16 = (lg.nseq * 4 + 15) / 16;
            hipLaunchKernelGGL(k_upload_words, dim3((unsigned)std::min<long long>((n16 + 255) / 256, 256)), dim3(256), 0, cs, (const u32x4 *)e->d_pos[slot].p, e->h_pos[slot].as<u32x4>(), n16);
        }
        results_by_kernel = true;
    } else {
        if (e->h_seq[slot].ensure((size_t)std::max<long long>(lg.nseq, 1) * 4)) return -1;
        if (hp_on && e->h_hp[slot].ensure((size_t)std::max<long long>(lg.nhp, 1) * 5 * 4)) return -1;
        HIPCHK(hipMemcpyAsync(e->h_seq[slot].p, e->d_seq[slot].p, (size_t)lg.nseq * 4, hipMemcpyDeviceToHost, cs));
        if (hp_on) HIPCHK(hipMemcpyAsync(e->h_hp[slot].p, e->d_hp[slot].p, (size_t)lg.nhp * 5 * 4, hipMemcpyDeviceToHost, cs));
    }
    if (!results_by_kernel) {
        HIPCHK(hipMemcpyAsync(e->h_score[slot].p, e->d_fscore[slot].p, lg.npad * 4, hipMemcpyDeviceToHost, cs));
        HIPCHK(hipMemcpyAsync(e->h_err[slot].p, e->d_gflag[slot].as<unsigned>() + lg.ntile, 4, hipMemcpyDeviceToHost, cs));
        HIPCHK(hipMemcpyAsync(e->h_bad[slot].p, e->d_bad[slot].p, lg.npad * 4, hipMemcpyDeviceToHost, cs));
    }
    HIPCHK(hipGetLastError());
    if (tun().host_stamp) fprintf(stderr, "host stamp: copies enqueued at %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - hs0).count());
    if (e->ev_ok) HIPCHK(hipEventRecord(e->done[slot], cs));
    lg.valid = true;
    lg.d_signal = d_signal; lg.in_off.assign(offsets, offsets + n); lg.in_len.assign(lengths, lengths + n); lg.params = *p;
    e->pending[slot] = true;
    if (!e->pending[slot ^ 1]) e->oldest = slot;
    return 0;
#undef EV
#undef ACC
}

/* ------------------------------------------------------------------ */
/* public batched surface                                               */
/* ------------------------------------------------------------------ */
extern "C" long scrappie_hip_run_device(scrappie_hip_engine *e, int model, const float *d_signal, const uint64_t *offsets,
                                        const uint32_t *lengths, size_t n, const scrappie_hip_params *p) {
    Model *m = get_model(e, model);
    if (!m) return -1;
    scrappie_hip_params dp = scrappie_hip_default_params();
    if (!p) p = &dp;
    if (n > e->max_launch_reads) { set_err("run_device: %zu reads exceed max_launch_reads %zu", n, e->max_launch_reads); return -1; }
    if (e->dbg_fail_run > 0 && --e->dbg_fail_run == 0) { set_err("run_device: injected failure (debug option fail_run)"); return -1; }
    const auto hs0 = std::chrono::steady_clock::now();
    if (run_pipeline(e, m, d_signal, offsets, lengths, n, p, STOP_NONE, 5, nullptr)) return -1;
    if (tun().host_stamp) fprintf(stderr, "host stamp: run_device %.2f ms on the host\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - hs0).count());
    return (long)e->lgs[e->cur].ncb;
}

/* D2 + D3 of one read on the host (sh_host.c): what k_stitch does on the device.  `path` (T + 1 entries) is consumed. */
static void host_stitch_read(const Model *m, bool hp_on, const float *side, int *path, int T, bool want_pos, scrappie_hip_call &c) {
    int *pos = (int *)calloc((size_t)T + 1, sizeof(int));
    if (!pos) { c.basecall = nullptr; c.basecall_length = 0; c.pos = nullptr; return; }
    char *bases;
    if (m->arch != 1) {
        if (hp_on) sh_homopolymer_side(side, path, T, m->NS);                       /* scrappie_raw.c:293 */
        bases = overlapper(path, (size_t)T + 1, m->NS - 1, pos);                    /* scrappie_raw.c:303 */
    } else {
        bases = crfpath_to_basecall(path, (size_t)T, pos);                          /* scrappie_raw.c:306 */
    }
    c.basecall = bases;
    c.basecall_length = bases ? strlen(bases) : 0;
    if (want_pos && bases) c.pos = pos; else { free(pos); c.pos = nullptr; }
}

static void stitch_range(scrappie_hip_engine *e, int slot, const Model *m, const scrappie_hip_params *p, scrappie_hip_call *out,
                         size_t lo, size_t hi) {
    const LaunchGroup &lg = e->lgs[slot];
    const float *scores = e->h_score[slot].as<float>();
    const unsigned *bad = e->h_bad[slot].as<unsigned>();
    for (size_t i = lo; i < hi; i++) {
        const int o = lg.order[i];
        if (o < 0) continue;
        scrappie_hip_call &c = out[o];
        c.score = NAN; c.nblock = 0; c.basecall = nullptr; c.basecall_length = 0; c.pos = nullptr;
        const int T = lg.rT[i];
        if (T <= 0) continue;
        if (bad[i]) continue;              /* input outside the split products' operand range: no call (reported by stitch_group) */
        c.score = scores[i];
        c.nblock = (size_t)T;
        if (lg.dev_stitch) {
            /* bases (and pos) were made by k_stitch: copy them out of the pinned buffers.  Reads whose homopolymer
             * mean sat on a rounding boundary (redo) are left to stitch_group. */
            if (e->h_redo[slot].as<unsigned>()[i] || e->dbg_redo_all) continue;
            const int len = e->h_blen[slot].as<int>()[i];
            if (len < 0) continue;                                   /* every entry a stay: no call (overlapper returns NULL) */
            char *bases = (char *)malloc((size_t)len + 1);
            if (!bases) continue;
            memcpy(bases, e->h_bases[slot].as<char>() + lg.bases_off[i], (size_t)len);
            bases[len] = 0;
            c.basecall = bases;
            c.basecall_length = (size_t)len;
            if (p->want_pos) {
                int *pos = (int *)malloc(((size_t)T + 1) * sizeof(int));
                if (pos) {
                    if (lg.dev_pos) { const int *src = e->h_pos[slot].as<int>() + lg.seq_off[i]; for (int t = 0; t <= T; t++) pos[t] = src[(size_t)t * SH_SEQ_STRIDE]; }
                    else memset(pos, 0, ((size_t)T + 1) * sizeof(int));             /* CRF: crfpath_to_basecall leaves pos untouched (Q11) */
                }
                c.pos = pos;
            }
            continue;
        }
        int *path = (int *)malloc(((size_t)T + 1) * sizeof(int));
        if (!path) continue;
        { const int *src = e->h_seq[slot].as<int>() + lg.seq_off[i]; for (int t = 0; t <= T; t++) path[t] = src[(size_t)t * SH_SEQ_STRIDE]; }
        host_stitch_read(m, lg.hp_on, lg.hp_on ? e->h_hp[slot].as<float>() + lg.hp_off[i] * 5 : nullptr, path, T, p->want_pos != 0, c);
        free(path);
    }
}

static int stitch_group(scrappie_hip_engine *e, int slot, Model *m, const scrappie_hip_params *p, scrappie_hip_call *out, size_t n);

/* Host threads for stitching a launch group: the CPUs this process may actually use -- its affinity mask and its
 * cgroup CPU quota (a GPU box may report 256 CPUs and grant 16), shared with the other ranks of a torchrun job
 * (LOCAL_WORLD_SIZE) -- at most 32, SCRAPPIE_HIP_HOST_THREADS overrides.  Decided once. */
static unsigned host_threads() {
    static const unsigned n = [] {
        if (const char *ev = getenv("SCRAPPIE_HIP_HOST_THREADS")) { const int v = atoi(ev); if (v > 0) return (unsigned)std::min(v, 256); }
        double cpus = (double)std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) cpus = std::min(cpus, (double)CPU_COUNT(&set));
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32]; double period = 0;
            if (fscanf(f, "%31s %lf", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) cpus = std::min(cpus, atof(q) / period);
            fclose(f);
        }
        if (const char *lw = getenv("LOCAL_WORLD_SIZE")) { const int w = atoi(lw); if (w > 1) cpus /= w; }
        return (unsigned)std::max(1.0, std::min(cpus, 32.0));
    }();
    return n;
}

extern "C" unsigned scrappie_hip_host_thread_budget(void) { return host_threads(); }

extern "C" int scrappie_hip_collect(scrappie_hip_engine *e, const scrappie_hip_params *p, scrappie_hip_call *out, size_t n) {
    if (!e || !out) return set_err("collect: null argument");
    scrappie_hip_params dp = scrappie_hip_default_params();
    if (!p) p = &dp;
    if (!e->pending[0] && !e->pending[1]) return set_err("collect: no launch group in flight");
    const int slot = e->pending[e->oldest] ? e->oldest : (e->oldest ^ 1);
    LaunchGroup &lg = e->lgs[slot];
    if (!lg.valid || lg.n != n) return set_err("collect: oldest launch group has %zu reads, asked for %zu", lg.n, n);
    (void)hipSetDevice(e->device);
    const auto hs0 = std::chrono::steady_clock::now();
    if (e->ev_ok) HIPCHK(hipEventSynchronize(e->done[slot]));
    else HIPCHK(hipStreamSynchronize(e->stream));
    const auto hs1 = std::chrono::steady_clock::now();
    struct StampOut { std::chrono::steady_clock::time_point a, b; bool on; ~StampOut() { if (on) fprintf(stderr, "host stamp: collect waited %.2f ms, then %.2f ms of host work\n", std::chrono::duration<double, std::milli>(b - a).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - b).count()); } } stamp_out{hs0, hs1, tun().host_stamp};
    e->pending[slot] = false;
    e->oldest = slot ^ 1;
    if (!e->spans[slot].empty() && resolve_spans(e, slot)) return -1;
    Model *m = get_model(e, lg.model);
    if (!m) return -1;
    static std::atomic<bool> fake_once{false};
    const bool faked = tun().fake_timeout && e->handover && lg.ncb > 0 && !fake_once.exchange(true);
    if (lg.ncb > 0 && e->h_err[slot].p && (*e->h_err[slot].as<unsigned>() != 0 || faked)) {
        /* a state hand-over between workgroups timed out (sh_wait_flag): the results of this group are
         * invalid.  Run it again scheduled on whole tiles (no inter-workgroup waits at all), behind whatever
         * else is in flight, and stitch that. */
        if (!e->handover) return set_err("launch group failed on the device (error word set without hand-overs)");
        const std::vector<uint64_t> off = lg.in_off;
        const std::vector<uint32_t> len = lg.in_len;
        const scrappie_hip_params pp = lg.params;
        const float *dsig = lg.d_signal;
        const int other_oldest = e->oldest;
        e->handover = false;
        const int rc = run_pipeline(e, m, dsig, off.data(), len.data(), n, &pp, STOP_NONE, 5, nullptr);
        e->handover = true;
        if (rc) return -1;
        const int rs = e->cur;
        if (e->ev_ok) HIPCHK(hipEventSynchronize(e->done[rs])); else HIPCHK(hipStreamSynchronize(e->stream));
        e->pending[rs] = false;
        e->oldest = other_oldest;
        if (!e->spans[rs].empty() && resolve_spans(e, rs)) return -1;
        if (*e->h_err[rs].as<unsigned>() != 0) return set_err("launch group failed on the device even on whole tiles");
        fprintf(stderr, "scrappie_hip: a state hand-over between workgroups timed out; launch group of %zu reads re-run on whole tiles\n", n);
        return stitch_group(e, rs, m, p, out, n);
    }
    return stitch_group(e, slot, m, p, out, n);
}

static int stitch_group(scrappie_hip_engine *e, int slot, Model *m, const scrappie_hip_params *p, scrappie_hip_call *out, size_t n) {
    LaunchGroup &lg = e->lgs[slot];
    for (size_t i = 0; i < n; i++) { out[i].score = NAN; out[i].nblock = 0; out[i].basecall = nullptr; out[i].basecall_length = 0; out[i].pos = nullptr; }
    if (lg.ncb == 0) return 0;
    {   /* reads whose input left the operand range of the split products (k_conv_act / k_feat_in): no call, said aloud */
        const unsigned *bad = e->h_bad[slot].as<unsigned>();
        size_t nbad = 0; long first = -1;
        for (size_t i = 0; i < lg.npad; i++) if (bad[i] && lg.order[i] >= 0) { if (!nbad || lg.order[i] < first) first = lg.order[i]; nbad++; }
        if (nbad) {
            set_err("%zu read(s) of this launch group (first: read %ld of it) hold values outside the supported range (|activation| >= %g "
                    "after the first layer, or non-finite): is the signal trimmed and med/MAD-normalised?  They get no call", nbad, first, (double)SH_ACT_LIMIT);
            fprintf(stderr, "scrappie_hip: %s\n", g_err);
        }
    }
    if (lg.dev_stitch && p->want_pos && !lg.dev_pos && m->arch != 1) {
        /* pos[] was not asked for when the group was enqueued: fetch the paths (and side rows) after all and stitch on the host */
        if (e->h_seq[slot].ensure((size_t)std::max<long long>(lg.nseq, 1) * 4) || (lg.hp_on && e->h_hp[slot].ensure((size_t)std::max<long long>(lg.nhp, 1) * 5 * 4))) return -1;
        HIPCHK(hipMemcpy(e->h_seq[slot].p, e->d_seq[slot].p, (size_t)lg.nseq * 4, hipMemcpyDeviceToHost));
        if (lg.hp_on) HIPCHK(hipMemcpy(e->h_hp[slot].p, e->d_hp[slot].p, (size_t)lg.nhp * 5 * 4, hipMemcpyDeviceToHost));
        lg.dev_stitch = false;
    }
    unsigned nthr = host_threads();
    if (e->host_thread_budget) nthr = std::min(nthr, e->host_thread_budget);
    if (lg.dev_stitch) nthr = std::min(nthr, 4u);             /* copying strings out of pinned memory: a few ms on one thread */
    if (lg.npad < 256) nthr = 1;
    if (nthr == 1) stitch_range(e, slot, m, p, out, 0, lg.npad);
    else {
        std::vector<std::thread> th;
        const size_t per = (lg.npad + nthr - 1) / nthr;
        for (unsigned t = 0; t < nthr; t++) {
            const size_t lo = t * per, hi = std::min(lg.npad, lo + per);
            if (lo >= hi) break;
            th.emplace_back(stitch_range, e, slot, m, p, out, lo, hi);
        }
        for (auto &x : th) x.join();
    }
    if (lg.dev_stitch) {
        /* reads k_stitch would not decide (a homopolymer run's posterior-mean count within rounding noise of a boundary):
         * their path and side rows are still on the device; the host code decides */
        const unsigned *redo = e->h_redo[slot].as<unsigned>();
        const unsigned *bad = e->h_bad[slot].as<unsigned>();
        size_t nredo = 0;
        for (size_t i = 0; i < lg.npad; i++) {
            const int o = lg.order[i], T = lg.rT[i];
            if (o < 0 || T <= 0 || bad[i] || !(redo[i] || e->dbg_redo_all)) continue;
            std::vector<int> path((size_t)T + 1);
            std::vector<float> side(lg.hp_on ? (size_t)T * 5 : 0);
            HIPCHK(hipMemcpy2D(path.data(), 4, e->d_seq[slot].as<int>() + lg.seq_off[i], (size_t)SH_SEQ_STRIDE * 4, 4, (size_t)T + 1, hipMemcpyDeviceToHost));
            if (lg.hp_on) HIPCHK(hipMemcpy(side.data(), e->d_hp[slot].as<float>() + lg.hp_off[i] * 5, (size_t)T * 5 * 4, hipMemcpyDeviceToHost));
            host_stitch_read(m, lg.hp_on, side.data(), path.data(), T, p->want_pos != 0, out[o]);
            nredo++;
        }
        e->n_redo += nredo;
    }
    return 0;
}

/* Launch groups of one call, two in flight: group g+1 is planned, staged and enqueued before the host
 * waits for group g and stitches it.  The call's reads are SORTED BY LENGTH (longest first, stable) before they are
 * cut into groups: a group lasts at least as long as its longest read's serial chain, so a long read should share
 * its group with other long reads, not hold a group of short ones hostage (mixed lengths).
 * stage(k, idx, cnt) makes the signals of reads idx[0..cnt) of the call available on the device and returns the
 * pointer/offset/length arrays to run them with. */
struct GroupArgs { const float *d; const uint64_t *off; const uint32_t *len; };

template <class Stage>
static int run_groups(scrappie_hip_engine *e, int model, const Model *m, const uint32_t *all_len, size_t n,
                      const scrappie_hip_params *p, scrappie_hip_call *out, Stage stage) {
    if (e->pending[0] || e->pending[1]) return set_err("launch groups are already in flight on this engine: collect them first");
    if (n == 0) return 0;
    const int unit = m->arch == 3 ? 1 : std::max(m->stride, 1);    /* events models: lengths already count blocks */
    std::vector<uint32_t> perm(n);
    std::iota(perm.begin(), perm.end(), 0u);
    /* (the decoder-input and trunk-input hooks address their data by a read's position in its launch group: input order) */
    if (!tun().input_order && !e->alt_prob && !e->alt_trunk) std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return all_len[a] > all_len[b]; });
    std::vector<uint32_t> sorted_len(n);
    for (size_t i = 0; i < n; i++) sorted_len[i] = all_len[perm[i]];
    std::vector<size_t> starts(n);
    const long ng = scrappie_hip_plan_groups(sorted_len.data(), n, unit, e->max_launch_reads, launch_block_cap(e, m), starts.data(), n);
    if (ng < 0) return set_err("a read is too long for one launch group on this device");
    starts.resize((size_t)ng); starts.push_back(n);
    auto blank = [&]() { for (size_t i = 0; i < n; i++) { out[i].score = NAN; out[i].nblock = 0; out[i].basecall = nullptr; out[i].basecall_length = 0; out[i].pos = nullptr; } };
    blank();
    auto fail = [&]() {   /* leave the engine drained; a failed call returns nothing: release the calls already stitched */
        (void)hipStreamSynchronize(e->pstream);
        (void)hipStreamSynchronize(e->stream);
        (void)hipStreamSynchronize(e->cstream);
        e->pending[0] = e->pending[1] = false;
        const std::string keep = g_err;
        scrappie_hip_free_calls(out, n);
        blank();
        set_err("%s", keep.c_str());
        return -1;
    };
    std::vector<scrappie_hip_call> tmp;
    auto collect = [&](size_t g) {      /* the group's calls come back in the group's order: hand them to their reads */
        const size_t lo = starts[g], cnt = starts[g + 1] - lo;
        tmp.assign(cnt, scrappie_hip_call{});      /* (never the previous group's pointers: out[] owns those; a failed collect frees what tmp holds) */
        if (scrappie_hip_collect(e, p, tmp.data(), cnt)) { scrappie_hip_free_calls(tmp.data(), cnt); return -1; }
        for (size_t i = 0; i < cnt; i++) out[perm[lo + i]] = tmp[i];
        return 0;
    };
    size_t prev = 0; bool have_prev = false;
    for (long g = 0; g < ng; g++) {
        const size_t lo = starts[g], cnt = starts[g + 1] - lo;
        GroupArgs a;
        int rc = stage((int)(g & 1), perm.data() + lo, cnt, a);
        if (!rc && scrappie_hip_run_device(e, model, a.d, a.off, a.len, cnt, p) < 0) rc = -1;
        if (have_prev && collect(prev)) rc = -1;
        have_prev = false;
        if (rc) return fail();
        prev = (size_t)g; have_prev = true;
    }
    if (have_prev && collect(prev)) return fail();
    return 0;
}

extern "C" int scrappie_hip_basecall_device(scrappie_hip_engine *e, int model, const float *d_signal, const uint64_t *offsets,
                                            const uint32_t *lengths, size_t n, const scrappie_hip_params *p, scrappie_hip_call *out) {
    if (!e || !out) return set_err("basecall_device: null argument");
    Model *m = get_model(e, model);
    if (!m) return -1;
    std::vector<uint64_t> off[2];
    std::vector<uint32_t> len[2];
    return run_groups(e, model, m, lengths, n, p, out, [&](int k, const uint32_t *idx, size_t cnt, GroupArgs &a) {
        off[k].resize(cnt); len[k].resize(cnt);
        for (size_t i = 0; i < cnt; i++) { off[k][i] = offsets[idx[i]]; len[k][i] = lengths[idx[i]]; }
        a.d = d_signal; a.off = off[k].data(); a.len = len[k].data();
        return 0;
    });
}

/* Arenas before the first real call: a launch group of n reads of `samples` samples each (all-zero signals) is run and thrown away, so
 * that every device and pinned buffer a group of that shape needs exists (hipMalloc / hipFree of gigabytes synchronise the device: a
 * call that grows its arenas stalls whatever else is running) and the kernels' code objects are resident.  `scrappie raw` calls it
 * while its loader reads the first full batch. */
extern "C" int scrappie_hip_warm_up(scrappie_hip_engine *e, int model, size_t n, size_t samples) {
    if (!e) return set_err("warm_up: null engine");
    Model *m = get_model(e, model);
    if (!m) return -1;
    if (n == 0 || samples == 0) return 0;
    (void)hipSetDevice(e->device);
    const size_t per = m->arch == 3 ? (size_t)m->nfeat : 1;
    float *d = nullptr;
    HIPCHK(hipMalloc(&d, n * samples * per * 4));
    int rc = hipMemset(d, 0, n * samples * per * 4) == hipSuccess ? 0 : set_err("warm_up: hipMemset failed");
    if (!rc) {
        std::vector<uint64_t> off(n);
        std::vector<uint32_t> len(n, (uint32_t)samples);
        for (size_t i = 0; i < n; i++) off[i] = (uint64_t)i * samples * per;
        std::vector<scrappie_hip_call> calls(n);
        rc = scrappie_hip_basecall_device(e, model, d, off.data(), len.data(), n, nullptr, calls.data());
        if (!rc) scrappie_hip_free_calls(calls.data(), n);
    }
    (void)hipFree(d);
    return rc;
}

/* Chain-bound reads.  A read is a serial chain -- five recurrent layers of alternating direction, then the decoder, the traceback
 * walk and the stitching: SH_CHAIN_NS per block, whatever else the device does -- so a launch group lasts at least as long as its
 * longest read, and a call whose length distribution has a long tail spends most of its time with a few workgroups stepping and the
 * rest of the device idle (profiles/r3_long_tail.txt: 24 000 reads of lognormal(20 000, 0.8) samples, one of 400 000: 912 ms of
 * chain for 470 ms of work).  The reads whose own chain is longer than what the whole call would take at the device's full
 * rate are "long": scrappie_hip_basecall_batch runs them on a helper engine of the same device (its own streams and arenas), beside
 * the launch groups of all the others, so that the call lasts max(longest chain, work) instead of their sum.  Host only:
 * is_long[n] gets 0 / 1; returns the number of long reads (0: do not split).  The reference's schedule(dynamic) loop over whole
 * reads (scrappie_raw.c:355-400) has the same effect on a CPU: a long read occupies one thread while the others go on. */
#ifndef SH_CHAIN_NS
#define SH_CHAIN_NS 11400.0     /* per block of one read: 5 x 1.29 us of recurrent layer + 5.3 decoder + 0.92 traceback walk + 0.34 stitching (lone tiles) */
#endif
#ifndef SH_WORK_NS
#define SH_WORK_NS 3.5          /* per block and read with the device full: 27.9 ms per 10 000 x 800 */
#endif
extern "C" long scrappie_hip_plan_tail(const uint32_t *lengths, size_t n, int stride, size_t max_long_blocks, unsigned char *is_long) {
    if ((!lengths && n) || stride < 1 || !is_long) return -1;
    memset(is_long, 0, n);
    if (n < 2) return 0;
    double W = 0;
    std::vector<uint32_t> T(n);
    for (size_t i = 0; i < n; i++) { T[i] = (uint32_t)(((unsigned long long)lengths[i] + stride - 1) / stride); W += T[i]; }
    const double thr = std::max(4096.0, W * SH_WORK_NS / SH_CHAIN_NS);             /* blocks; never below ~20 000 samples */
    std::vector<uint32_t> cand;
    for (size_t i = 0; i < n; i++) if ((double)T[i] > thr) cand.push_back((uint32_t)i);
    if (cand.empty() || cand.size() == n) return 0;
    std::stable_sort(cand.begin(), cand.end(), [&](uint32_t a, uint32_t b) { return T[a] > T[b]; });
    /* the longest first, as long as they fit the helper's arena in ONE launch group (tiles of 16: blocks of a tile = its longest read's)
     * and stay a small part of the call: a second group would add its own chain, and a call of mostly long reads has no tail to hide */
    double blocks = 0, work = 0;
    long cnt = 0;
    for (size_t k = 0; k < cand.size(); k++) {
        if (k % 16 == 0) { if (max_long_blocks && blocks + T[cand[k]] > (double)max_long_blocks) break; blocks += T[cand[k]]; }
        if (work + T[cand[k]] > 0.15 * W) break;
        work += T[cand[k]];
        is_long[cand[k]] = 1; cnt++;
    }
    return cnt;
}

static int basecall_batch_one(scrappie_hip_engine *e, int model, const raw_table *reads, size_t n, const scrappie_hip_params *p, scrappie_hip_call *out);

static bool tail_two_helpers() {
    /* SCRAPPIE_HIP_TAIL=2: a second helper when the first is busy.  Measured SLOWER on a stream of long-tailed calls (24 x 12 000 lognormal reads:
     * 7.3e8 against 8.7e8 samples/s, profiles/r4_long_tail.txt): two chain-bound groups hold twice the CUs and each serves fewer tickets */
    static const bool on = [] { const char *v = getenv("SCRAPPIE_HIP_TAIL"); return v && atoi(v) == 2; }();
    return on;
}
static bool tail_enabled(const scrappie_hip_engine *e) {
    if (e->is_tail) return false;
    if (e->tail_mode >= 0) return e->tail_mode != 0;
    static const bool env_on = [] { const char *v = getenv("SCRAPPIE_HIP_TAIL"); return !(v && atoi(v) == 0); }();
    return env_on;
}

/* a helper engine: same device, same models at the same indices, same settings */
static scrappie_hip_engine *make_helper(scrappie_hip_engine *e) {
    scrappie_hip_engine *t = scrappie_hip_engine_create(e->device);
    if (!t) return nullptr;
    t->is_tail = true;
    for (const auto &b : e->blobs) {
        t->dbg_force_f32 = b.force_f32;
        const int want = scrappie_hip_find_model(e, b.name.c_str());
        if (load_model_mem_one(t, b.name.c_str(), b.bytes.data(), b.bytes.size()) != want) { scrappie_hip_engine_destroy(t); set_err("helper engine: model '%s' did not load at index %d", b.name.c_str(), want); return nullptr; }
    }
    return t;
}
static void sync_helper(scrappie_hip_engine *e, scrappie_hip_engine *t) {
    t->handover = e->handover; t->max_launch_reads = e->max_launch_reads; t->max_launch_blocks = e->max_launch_blocks;
    t->dbg_ff_separate = e->dbg_ff_separate; t->dbg_fv_single = e->dbg_fv_single; t->dbg_gru32 = e->dbg_gru32; t->dbg_gru_tiles = e->dbg_gru_tiles; t->dbg_redo_all = e->dbg_redo_all;
    t->profiling = false;
}
static scrappie_hip_engine *tail_engine(scrappie_hip_engine *e) {
    if (!e->tail) {
        e->tail = make_helper(e);
        if (!e->tail) return nullptr;
        /* several arenas on one device: a helper's launch groups are a few long tiles */
        e->mem_frac = 0.45; e->tail->mem_frac = 0.15;
    }
    std::lock_guard<std::mutex> lk(e->tail_mu);           /* (a helper's settings change only while it is idle) */
    if (e->tail_busy == 0) {
        sync_helper(e, e->tail);
        if (e->tail2) sync_helper(e, e->tail2);
        if (e->dbg_fail_tail) { e->tail->dbg_fail_run = e->dbg_fail_tail; e->dbg_fail_tail = 0; }
    }
    return e->tail;
}

typedef std::shared_ptr<scrappie_hip_engine::TailTicket> TicketPtr;
static void tail_worker(scrappie_hip_engine *e, scrappie_hip_engine *helper) {
    for (;;) {
        std::unique_lock<std::mutex> lk(e->tail_mu);
        e->tail_cv.wait(lk, [&] { return e->tail_stop || !e->tail_q.empty(); });
        if (e->tail_q.empty()) break;                 /* (stop: what is queued is still served) */
        std::vector<TicketPtr> batch;
        const TicketPtr f = e->tail_q.front();
        while (!e->tail_q.empty() && e->tail_q.front()->model == f->model && memcmp(&e->tail_q.front()->p, &f->p, sizeof f->p) == 0) {
            batch.push_back(e->tail_q.front());
            e->tail_q.pop_front();
        }
        e->tail_busy++;
        lk.unlock();
        std::vector<raw_table> all;
        for (const TicketPtr &t : batch) all.insert(all.end(), t->reads.begin(), t->reads.end());
        std::vector<scrappie_hip_call> calls(all.size());
        const int rc = basecall_batch_one(helper, f->model, all.data(), all.size(), &f->p, calls.data());
        const std::string err = rc ? std::string(g_err) : std::string();
        lk.lock();
        size_t at = 0;
        for (const TicketPtr &t : batch) {
            t->rc = rc; t->err = err;
            if (!rc) t->calls.assign(calls.begin() + (long)at, calls.begin() + (long)(at + t->reads.size()));
            at += t->reads.size();
            t->done = true;
        }
        e->n_tail_groups++;
        e->n_redo_tail += helper->n_redo; helper->n_redo = 0;
        e->tail_busy--;
        lk.unlock();
        e->tail_cv.notify_all();
    }
}
static TicketPtr tail_submit(scrappie_hip_engine *e, int model, const scrappie_hip_params &p, std::vector<raw_table> &&reads, std::vector<float> &&own = std::vector<float>()) {
    if (!e->tail2 && tail_two_helpers()) {
        bool busy;
        { std::lock_guard<std::mutex> lk(e->tail_mu); busy = e->tail_busy > 0 || !e->tail_q.empty(); }
        if (busy) {                                    /* the first helper has a group in hand: a second one for what comes now */
            scrappie_hip_engine *t2 = make_helper(e);
            if (t2) { t2->mem_frac = 0.15; sync_helper(e, t2); std::lock_guard<std::mutex> lk(e->tail_mu); e->tail2 = t2; }
            (void)hipSetDevice(e->device);
        }
    }
    TicketPtr t = std::make_shared<scrappie_hip_engine::TailTicket>();
    t->model = model; t->p = p; t->reads = std::move(reads);
    t->own = std::move(own);          /* (a vector's buffer moves with it: tables that point into it stay good) */
    {
        std::lock_guard<std::mutex> lk(e->tail_mu);
        t->id = e->tail_next++;
        e->tail_open[t->id] = t;
        e->tail_q.push_back(t);
        if (!e->tail_th_live) { e->tail_th = std::thread(tail_worker, e, e->tail); e->tail_th_live = true; }
        else if (e->tail2 && !e->tail_th2_live) { e->tail_th2 = std::thread(tail_worker, e, e->tail2); e->tail_th2_live = true; }
    }
    e->tail_cv.notify_all();
    return t;
}
static void tail_wait(scrappie_hip_engine *e, const TicketPtr &t) {
    std::unique_lock<std::mutex> lk(e->tail_mu);
    e->tail_cv.wait(lk, [&] { return t->done; });
}

/* what scrappie_hip_basecall_batch and _deferred share: lengths, the plan, the helper engine.  Returns the number of long reads (0: no
 * split), -1 on error */
static long tail_plan_len(scrappie_hip_engine *e, Model *m, const uint32_t *len, size_t n, std::vector<unsigned char> &is_long) {
    is_long.assign(n, 0);
    if (!tail_enabled(e) || n < 2 || e->alt_prob || e->alt_trunk || e->blobs.size() != e->models.size() || e->dbg_fail_run) return 0;
    const int unit = m->arch == 3 ? 1 : std::max(m->stride, 1);
    const size_t tail_cap = (size_t)(0.15 * (double)e->total_mem) / bytes_per_block(m, !decoder_fused(e, m));
    const long nl = scrappie_hip_plan_tail(len, n, unit, e->max_launch_blocks ? e->max_launch_blocks : tail_cap, is_long.data());
    if (nl > 0 && !tail_engine(e)) return -1;
    return nl;
}
static long tail_plan(scrappie_hip_engine *e, Model *m, const raw_table *reads, size_t n, std::vector<unsigned char> &is_long) {
    const size_t per = m->arch == 3 ? (size_t)m->nfeat : 1;
    std::vector<uint32_t> len(n);
    for (size_t i = 0; i < n; i++) {
        const raw_table &rt = reads[i];
        len[i] = (uint32_t)(((rt.raw && rt.end > rt.start) ? rt.end - rt.start : 0) / per);
    }
    return tail_plan_len(e, m, len.data(), n, is_long);
}

/* scrappie_hip_basecall_batch that does not wait for the chain-bound reads: their calls are collected later (scrappie_hip_deferred_collect),
 * so that the NEXT call's launch groups run beside them too -- a stream of calls with long-tailed read lengths then runs at the
 * device's rate instead of one longest-read chain per call.  deferred[n] gets 1 for the reads whose out[] entry is still blank.
 * Returns a ticket (> 0) if any read was deferred, 0 if none, -1 on error.  The deferred reads' signals must stay valid until
 * their ticket has been collected. */
extern "C" long scrappie_hip_basecall_batch_deferred(scrappie_hip_engine *e, int model, const raw_table *reads, size_t n,
                                                     const scrappie_hip_params *p, scrappie_hip_call *out, unsigned char *deferred) {
    if (!e || !reads || !out || !deferred) return set_err("basecall_batch_deferred: null argument");
    Model *m = get_model(e, model);
    if (!m) return -1;
    std::vector<unsigned char> is_long;
    const long nl = tail_plan(e, m, reads, n, is_long);
    if (nl < 0) return -1;
    memset(deferred, 0, n);
    if (nl == 0) return basecall_batch_one(e, model, reads, n, p, out) ? -1 : 0;
    std::vector<raw_table> rl, rr;
    std::vector<size_t> ir;
    for (size_t i = 0; i < n; i++) { if (is_long[i]) rl.push_back(reads[i]); else { rr.push_back(reads[i]); ir.push_back(i); } }
    scrappie_hip_params dp = scrappie_hip_default_params();
    const size_t nlong = rl.size();
    auto tk = tail_submit(e, model, p ? *p : dp, std::move(rl));
    std::vector<scrappie_hip_call> orr(rr.size());
    const int rc_main = basecall_batch_one(e, model, rr.data(), rr.size(), p, orr.data());
    for (size_t i = 0; i < n; i++) { out[i].score = NAN; out[i].nblock = 0; out[i].basecall = nullptr; out[i].basecall_length = 0; out[i].pos = nullptr; }
    if (rc_main) {                                    /* nothing is returned: the ticket is withdrawn */
        const std::string keep = g_err;
        tail_wait(e, tk);
        if (!tk->rc) scrappie_hip_free_calls(tk->calls.data(), tk->calls.size());
        { std::lock_guard<std::mutex> lk(e->tail_mu); e->tail_open.erase(tk->id); }
        return set_err("%s", keep.c_str());
    }
    for (size_t k = 0; k < ir.size(); k++) out[ir[k]] = orr[k];
    for (size_t i = 0; i < n; i++) deferred[i] = is_long[i];
    e->n_tail_calls++; e->n_tail_reads += nlong;
    return tk->id;
}
/* The same for signals that are already on the device (prepared there: scrappie_hip_prep_run): the few chain-bound reads are copied
 * back to host memory the ticket owns -- the helper engine stages from there, and the caller may reuse its device buffer as soon as
 * this call returns -- the others run from d_signal as scrappie_hip_basecall_device runs them. */
extern "C" long scrappie_hip_basecall_device_deferred(scrappie_hip_engine *e, int model, const float *d_signal, const uint64_t *offsets,
                                                      const uint32_t *lengths, size_t n, const scrappie_hip_params *p, scrappie_hip_call *out,
                                                      unsigned char *deferred) {
    if (!e || !out || !deferred || (n && (!offsets || !lengths))) return set_err("basecall_device_deferred: null argument");
    Model *m = get_model(e, model);
    if (!m) return -1;
    std::vector<unsigned char> is_long;
    const long nl = m->arch == 3 ? 0 : tail_plan_len(e, m, lengths, n, is_long);
    if (nl < 0) return -1;
    memset(deferred, 0, n);
    if (nl == 0) return scrappie_hip_basecall_device(e, model, d_signal, offsets, lengths, n, p, out) ? -1 : 0;
    (void)hipSetDevice(e->device);
    TicketPtr tk;
    std::vector<uint64_t> off2; std::vector<uint32_t> len2; std::vector<size_t> ir;
    {
        size_t total = 0;
        for (size_t i = 0; i < n; i++) if (is_long[i]) total += lengths[i];
        std::vector<float> own(total);
        std::vector<raw_table> rl;
        size_t at = 0;
        for (size_t i = 0; i < n; i++) {
            if (!is_long[i]) { off2.push_back(offsets[i]); len2.push_back(lengths[i]); ir.push_back(i); continue; }
            if (hipMemcpyAsync(own.data() + at, d_signal + offsets[i], (size_t)lengths[i] * 4, hipMemcpyDeviceToHost, e->ustream) != hipSuccess)      /* (the engine's own stream: the null stream would wait for the helper's group in flight) */
                return set_err("basecall_device_deferred: copying a chain-bound read back failed");
            rl.push_back(raw_table{nullptr, lengths[i], 0, lengths[i], own.data() + at});
            at += lengths[i];
        }
        HIPCHK(hipStreamSynchronize(e->ustream));
        scrappie_hip_params dp = scrappie_hip_default_params();
        tk = tail_submit(e, model, p ? *p : dp, std::move(rl), std::move(own));
    }
    std::vector<scrappie_hip_call> orr(ir.size());
    const int rc_main = scrappie_hip_basecall_device(e, model, d_signal, off2.data(), len2.data(), ir.size(), p, orr.data());
    for (size_t i = 0; i < n; i++) { out[i].score = NAN; out[i].nblock = 0; out[i].basecall = nullptr; out[i].basecall_length = 0; out[i].pos = nullptr; }
    if (rc_main) {                                    /* nothing is returned: the ticket is withdrawn */
        const std::string keep = g_err;
        tail_wait(e, tk);
        if (!tk->rc) scrappie_hip_free_calls(tk->calls.data(), tk->calls.size());
        { std::lock_guard<std::mutex> lk(e->tail_mu); e->tail_open.erase(tk->id); }
        return set_err("%s", keep.c_str());
    }
    for (size_t k = 0; k < ir.size(); k++) out[ir[k]] = orr[k];
    for (size_t i = 0; i < n; i++) deferred[i] = is_long[i];
    e->n_tail_calls++; e->n_tail_reads += (unsigned long long)nl;
    return tk->id;
}
/* The calls of a ticket's deferred reads, in the order those reads had in their call.  wait = 0: returns -2 if they are not ready.
 * Returns their number, -1 on error (unknown ticket, out[] too small, or the helper's launch group failed: the ticket is gone). */
extern "C" long scrappie_hip_deferred_collect(scrappie_hip_engine *e, long ticket, scrappie_hip_call *out, size_t cap, int wait) {
    if (!e || !out) return set_err("deferred_collect: null argument");
    TicketPtr t;
    {
        std::lock_guard<std::mutex> lk(e->tail_mu);
        auto it = e->tail_open.find(ticket);
        if (it == e->tail_open.end()) return set_err("deferred_collect: no such ticket");
        t = it->second;
        if (!t->done && !wait) return -2;
    }
    tail_wait(e, t);
    (void)hipSetDevice(e->device);
    if (!t->rc && t->calls.size() > cap) return set_err("deferred_collect: %zu calls, room for %zu", t->calls.size(), cap);
    { std::lock_guard<std::mutex> lk(e->tail_mu); e->tail_open.erase(ticket); }
    if (t->rc) return set_err("%s", t->err.c_str());
    for (size_t k = 0; k < t->calls.size(); k++) out[k] = t->calls[k];
    return (long)t->calls.size();
}

extern "C" int scrappie_hip_basecall_batch(scrappie_hip_engine *e, int model, const raw_table *reads, size_t n,
                                           const scrappie_hip_params *p, scrappie_hip_call *out) {
    if (!e || !reads || !out) return set_err("basecall_batch: null argument");
    Model *m = get_model(e, model);
    if (!m) return -1;
    {
        std::vector<unsigned char> is_long;
        const long nl = tail_plan(e, m, reads, n, is_long);
        if (nl < 0) return -1;
        if (nl > 0) {
            std::vector<raw_table> rl, rr;
            std::vector<size_t> il, ir;
            for (size_t i = 0; i < n; i++) { if (is_long[i]) { rl.push_back(reads[i]); il.push_back(i); } else { rr.push_back(reads[i]); ir.push_back(i); } }
            std::vector<scrappie_hip_call> orr(rr.size());
            scrappie_hip_params dp = scrappie_hip_default_params();
            auto tk = tail_submit(e, model, p ? *p : dp, std::move(rl));
            const int rc_main = basecall_batch_one(e, model, rr.data(), rr.size(), p, orr.data());
            const std::string err_main = rc_main ? std::string(g_err) : std::string();
            tail_wait(e, tk);
            (void)hipSetDevice(e->device);
            { std::lock_guard<std::mutex> lk(e->tail_mu); e->tail_open.erase(tk->id); }
            if (rc_main || tk->rc) {      /* a failed call returns nothing */
                if (!rc_main) scrappie_hip_free_calls(orr.data(), orr.size());
                if (!tk->rc) scrappie_hip_free_calls(tk->calls.data(), tk->calls.size());
                for (size_t i = 0; i < n; i++) { out[i].score = NAN; out[i].nblock = 0; out[i].basecall = nullptr; out[i].basecall_length = 0; out[i].pos = nullptr; }
                return set_err("%s", rc_main ? err_main.c_str() : tk->err.c_str());
            }
            for (size_t k = 0; k < il.size(); k++) out[il[k]] = tk->calls[k];
            for (size_t k = 0; k < ir.size(); k++) out[ir[k]] = orr[k];
            e->n_tail_calls++; e->n_tail_reads += il.size();
            return 0;
        }
    }
    return basecall_batch_one(e, model, reads, n, p, out);
}

static int basecall_batch_one(scrappie_hip_engine *e, int model, const raw_table *reads, size_t n,
                              const scrappie_hip_params *p, scrappie_hip_call *out) {
    if (!e || !reads || !out) return set_err("basecall_batch: null argument");
    Model *m = get_model(e, model);
    if (!m) return -1;
    (void)hipSetDevice(e->device);
    const size_t per = m->arch == 3 ? (size_t)m->nfeat : 1;    /* events: lengths count events of nfeat floats */
    std::vector<uint32_t> len(n);
    for (size_t i = 0; i < n; i++) {
        const raw_table &rt = reads[i];
        const size_t ns = (rt.raw && rt.end > rt.start) ? rt.end - rt.start : 0;
        len[i] = (uint32_t)(ns / per);
    }
    std::vector<uint64_t> off[2];
    std::vector<uint32_t> glen[2];
    bool used[2] = {false, false};
    return run_groups(e, model, m, len.data(), n, p, out, [&](int k, const uint32_t *idx, size_t cnt, GroupArgs &a) {
        /* staging buffer k was last read by the upload of group g-2 */
        if (used[k] && e->ev_ok) HIPCHK(hipEventSynchronize(e->up[k]));
        off[k].resize(cnt); glen[k].resize(cnt);
        size_t total = 0;
        for (size_t i = 0; i < cnt; i++) { glen[k][i] = len[idx[i]]; off[k][i] = total; total += (size_t)glen[k][i] * per; }
        if (e->h_sig[k].ensure(std::max<size_t>(total, 1) * 4) || e->d_signal[k].ensure(std::max<size_t>(total, 1) * 4)) return -1;
        float *hs = e->h_sig[k].as<float>();
        {   /* gather into the pinned staging buffer on several host threads (160 MB per 10 000 x 4000-sample group: 20 ms on one) */
            const unsigned nthr = (total * 4 > ((size_t)8 << 20)) ? std::max(1u, std::min(host_threads(), 8u)) : 1u;
            auto part = [&](size_t a, size_t b) {
                for (size_t i = a; i < b; i++)
                    if (glen[k][i]) memcpy(hs + off[k][i], reads[idx[i]].raw + reads[idx[i]].start, (size_t)glen[k][i] * per * 4);
            };
            if (nthr == 1) part(0, cnt);
            else {
                std::vector<std::thread> th;
                const size_t step = (cnt + nthr - 1) / nthr;
                for (unsigned t = 0; t < nthr; t++) { const size_t a = t * step, b = std::min(cnt, a + step); if (a < b) th.emplace_back(part, a, b); }
                for (auto &x : th) x.join();
            }
        }
        hipStream_t us = e->ev_ok ? e->ustream : e->stream;
        HIPCHK(hipMemcpyAsync(e->d_signal[k].p, hs, total * 4, hipMemcpyHostToDevice, us));
        if (e->ev_ok) { HIPCHK(hipEventRecord(e->up[k], us)); HIPCHK(hipStreamWaitEvent(e->stream, e->up[k], 0)); HIPCHK(hipStreamWaitEvent(e->pstream, e->up[k], 0)); }
        else HIPCHK(hipStreamSynchronize(e->stream));
        used[k] = true;
        a.d = e->d_signal[k].as<float>(); a.off = off[k].data(); a.len = glen[k].data();
        return 0;
    });
}

/* ------------------------------------------------------------------ */
/* several GPUs: reads handed out dynamically                           */
/* ------------------------------------------------------------------ */
/* The reference's one parallel axis is reads, `#pragma omp parallel for schedule(dynamic)` (scrappie_raw.c:355,387).
 * Here the unit handed out is a LAUNCH GROUP: reads are sorted by length (longest first, so the expensive groups
 * start early and the short ones fill the tail), cut into groups, and every engine's host thread takes the next
 * group from one atomic cursor when it has room for it (two groups in flight per engine).  No exchange between
 * GPUs: weights are replicated, a read lives on one GPU from signal to bases. */
extern "C" long scrappie_hip_plan_dynamic(const uint32_t *lengths, size_t n, int stride, size_t nengine, size_t max_reads,
                                          size_t max_blocks, uint32_t *order, size_t *starts, size_t cap) {
    if ((!lengths && n) || stride < 1 || nengine < 1 || max_reads < 16 || !order) return -1;
    std::vector<uint32_t> idx(n);
    std::iota(idx.begin(), idx.end(), 0u);
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return lengths[a] > lengths[b]; });
    memcpy(order, idx.data(), n * sizeof(uint32_t));
    /* groups small enough that every engine gets several (dynamic balance), large enough to fill a GPU:
     * about n / (4 engines) reads, between 4096 and max_reads */
    size_t per = (n + 4 * nengine - 1) / (4 * nengine);
    per = std::min(max_reads, std::max<size_t>(per, std::min<size_t>(4096, max_reads)));
    std::vector<uint32_t> sorted_len(n);
    for (size_t i = 0; i < n; i++) sorted_len[i] = lengths[idx[i]];
    return scrappie_hip_plan_groups(sorted_len.data(), n, stride, per, max_blocks, starts, cap);
}

extern "C" int scrappie_hip_basecall_batch_multi(scrappie_hip_engine *const *engines, const int *models, size_t nengine,
                                                 const raw_table *reads, size_t n, const scrappie_hip_params *p,
                                                 scrappie_hip_call *out) {
    if (!engines || !models || nengine == 0 || (!reads && n) || !out) return set_err("basecall_batch_multi: null argument");
    if (nengine == 1) return scrappie_hip_basecall_batch(engines[0], models[0], reads, n, p, out);
    scrappie_hip_params dp = scrappie_hip_default_params();
    if (!p) p = &dp;
    std::vector<Model *> ms(nengine);
    for (size_t k = 0; k < nengine; k++) {
        ms[k] = get_model(engines[k], models[k]);
        if (!ms[k]) return -1;
        if (ms[k]->stride != ms[0]->stride || ms[k]->arch != ms[0]->arch || ms[k]->NS != ms[0]->NS)
            return set_err("basecall_batch_multi: the engines hold different models");
        if (engines[k]->pending[0] || engines[k]->pending[1]) return set_err("basecall_batch_multi: launch groups are already in flight on engine %zu", k);
    }
    for (size_t i = 0; i < n; i++) { out[i].score = NAN; out[i].nblock = 0; out[i].basecall = nullptr; out[i].basecall_length = 0; out[i].pos = nullptr; }
    if (n == 0) return 0;
    const Model *m0 = ms[0];
    const size_t per = m0->arch == 3 ? (size_t)m0->nfeat : 1;
    std::vector<uint32_t> len(n), order(n);
    for (size_t i = 0; i < n; i++) {
        const raw_table &rt = reads[i];
        const size_t ns = (rt.raw && rt.end > rt.start) ? rt.end - rt.start : 0;
        len[i] = (uint32_t)(ns / per);
    }
    size_t max_reads = engines[0]->max_launch_reads, max_blocks = launch_block_cap(engines[0], ms[0]);
    for (size_t k = 1; k < nengine; k++) { max_reads = std::min(max_reads, engines[k]->max_launch_reads); max_blocks = std::min(max_blocks, launch_block_cap(engines[k], ms[k])); }
    std::vector<size_t> starts(n + 1);
    const int unit = m0->arch == 3 ? 1 : std::max(m0->stride, 1);
    const long ng = scrappie_hip_plan_dynamic(len.data(), n, unit, nengine, max_reads, max_blocks, order.data(), starts.data(), n);
    if (ng < 0) return set_err("the reads cannot be cut into launch groups (a read longer than a launch group on these devices holds, or invalid planning arguments)");
    starts.resize((size_t)ng); starts.push_back(n);

    std::atomic<long> cursor{0};
    std::atomic<int> failed{0};
    std::vector<std::string> errs(nengine);
    auto worker = [&](size_t k) {
        scrappie_hip_engine *e = engines[k];
        (void)hipSetDevice(e->device);
        struct Flight { long g = -1; std::vector<uint64_t> off; std::vector<uint32_t> len; std::vector<scrappie_hip_call> calls; };
        Flight fl[2];
        int nf = 0;                                /* groups enqueued so far on this engine */
        long prev = -1;                            /* slot index (nf - 1) & 1 holds the group not yet collected */
        auto collect_one = [&](Flight &f) -> int {
            const size_t lo = starts[(size_t)f.g], cnt = starts[(size_t)f.g + 1] - lo;
            if (scrappie_hip_collect(e, p, f.calls.data(), cnt)) return -1;
            for (size_t i = 0; i < cnt; i++) out[order[lo + i]] = f.calls[i];      /* ownership of the strings moves to out[] */
            return 0;
        };
        for (;;) {
            const long g = failed.load() ? ng : cursor.fetch_add(1);
            if (g >= ng) break;
            Flight &f = fl[nf & 1];
            const size_t lo = starts[(size_t)g], cnt = starts[(size_t)g + 1] - lo;
            const int kbuf = nf & 1;
            /* staging buffer kbuf was last read by the upload of this engine's group nf - 2 */
            int rc = 0;
            if (nf >= 2 && e->ev_ok && hipEventSynchronize(e->up[kbuf]) != hipSuccess) rc = set_err("hipEventSynchronize failed");
            f.g = g; f.off.resize(cnt); f.len.resize(cnt); f.calls.assign(cnt, scrappie_hip_call{});
            size_t total = 0;
            for (size_t i = 0; i < cnt; i++) { f.len[i] = len[order[lo + i]]; f.off[i] = total; total += (size_t)f.len[i] * per; }
            if (!rc && (e->h_sig[kbuf].ensure(std::max<size_t>(total, 1) * 4) || e->d_signal[kbuf].ensure(std::max<size_t>(total, 1) * 4))) rc = -1;
            if (!rc) {
                float *hs = e->h_sig[kbuf].as<float>();
                for (size_t i = 0; i < cnt; i++) {
                    const raw_table &rt = reads[order[lo + i]];
                    if (f.len[i]) memcpy(hs + f.off[i], rt.raw + rt.start, (size_t)f.len[i] * per * 4);
                }
                hipStream_t us = e->ev_ok ? e->ustream : e->stream;
                if (hipMemcpyAsync(e->d_signal[kbuf].p, hs, total * 4, hipMemcpyHostToDevice, us) != hipSuccess) rc = set_err("hipMemcpyAsync (signals) failed");
                if (!rc && e->ev_ok && (hipEventRecord(e->up[kbuf], us) != hipSuccess || hipStreamWaitEvent(e->stream, e->up[kbuf], 0) != hipSuccess || hipStreamWaitEvent(e->pstream, e->up[kbuf], 0) != hipSuccess)) rc = set_err("event failed");
                if (!rc && !e->ev_ok && hipStreamSynchronize(e->stream) != hipSuccess) rc = set_err("sync failed");
            }
            if (!rc && scrappie_hip_run_device(e, models[k], e->d_signal[kbuf].as<float>(), f.off.data(), f.len.data(), cnt, p) < 0) rc = -1;
            if (!rc) nf++;
            if (!rc && prev >= 0 && collect_one(fl[(nf - 2) & 1])) rc = -1;      /* the older group, while the new one runs */
            if (rc) { errs[k] = scrappie_hip_last_error(); failed.store(1); break; }
            prev = g;
        }
        if (!failed.load() && prev >= 0 && collect_one(fl[(nf - 1) & 1])) { errs[k] = scrappie_hip_last_error(); failed.store(1); }
        if (failed.load()) {       /* leave the engine drained */
            (void)hipStreamSynchronize(e->pstream);
            (void)hipStreamSynchronize(e->stream);
            (void)hipStreamSynchronize(e->cstream);
            e->pending[0] = e->pending[1] = false;
        }
    };
    /* the engines of one call share the host: each stitches with its share of the CPUs this process may use */
    for (size_t k = 0; k < nengine; k++) engines[k]->host_thread_budget = std::max(1u, host_threads() / (unsigned)nengine);
    std::vector<std::thread> th;
    for (size_t k = 0; k < nengine; k++) th.emplace_back(worker, k);
    for (auto &t : th) t.join();
    for (size_t k = 0; k < nengine; k++) engines[k]->host_thread_budget = 0;
    if (failed.load()) {
        /* out[] owns the strings of every launch group collected before the failure: a failed call returns nothing,
         * so they are released here and out[] is left as it would be for n reads without a call */
        scrappie_hip_free_calls(out, n);
        for (size_t i = 0; i < n; i++) { out[i].score = NAN; out[i].nblock = 0; out[i].basecall = nullptr; out[i].basecall_length = 0; out[i].pos = nullptr; }
        for (size_t k = 0; k < nengine; k++) if (!errs[k].empty()) return set_err("engine %zu: %s", k, errs[k].c_str());
        return set_err("basecall_batch_multi failed");
    }
    return 0;
}

extern "C" int scrappie_hip_set_decoder_input(scrappie_hip_engine *e, const float *d_prob, const uint64_t *prob_off, size_t n_prob) {
    if (!e) return set_err("set_decoder_input: null engine");
    if (e->pending[0] || e->pending[1]) return set_err("set_decoder_input: launch groups are in flight");
    e->alt_valid = false;
    if (!d_prob) { e->alt_prob = nullptr; e->alt_off.clear(); return 0; }
    if (!prob_off || n_prob == 0) return set_err("set_decoder_input: no offsets");
    e->alt_prob = d_prob;
    e->alt_off.assign(prob_off, prob_off + n_prob);
    return 0;
}

extern "C" int scrappie_hip_set_trunk_input(scrappie_hip_engine *e, const float *d_trunk, const uint64_t *trunk_off, size_t n_trunk) {
    if (!e) return set_err("set_trunk_input: null engine");
    if (e->pending[0] || e->pending[1]) return set_err("set_trunk_input: launch groups are in flight");
    e->trk_valid = false;
    if (!d_trunk) { e->alt_trunk = nullptr; e->trk_off.clear(); return 0; }
    if (!trunk_off || n_trunk == 0) return set_err("set_trunk_input: no offsets");
    e->alt_trunk = d_trunk;
    e->trk_off.assign(trunk_off, trunk_off + n_trunk);
    return 0;
}

/* Test hooks.  Options: "ff_separate" (S1 and the decoder as two kernels on this engine, whatever the shape),
 * "dump_final" (the decoders leave every tile's final scores, start and end state in the hand-over buffer). */
extern "C" int scrappie_hip_debug_option(scrappie_hip_engine *e, const char *name, int value) {
    if (!e || !name) return set_err("debug_option: null argument");
    if (e->pending[0] || e->pending[1]) return set_err("debug_option: launch groups are in flight");
    if (!strcmp(name, "ff_separate")) e->dbg_ff_separate = value != 0;
    else if (!strcmp(name, "fv_single")) e->dbg_fv_single = value != 0;
    else if (!strcmp(name, "dump_final")) e->dbg_dump_final = value != 0;
    else if (!strcmp(name, "fail_run")) e->dbg_fail_run = value;
    else if (!strcmp(name, "redo_all")) e->dbg_redo_all = value != 0;
    else if (!strcmp(name, "gru_tiles")) e->dbg_gru_tiles = value;
    else if (!strcmp(name, "force_f32_layers")) e->dbg_force_f32 = value != 0;
#ifdef SH_EXPERIMENTS
    else if (!strcmp(name, "gru32")) e->dbg_gru32 = value;
#endif
    else if (!strcmp(name, "tail")) e->tail_mode = value;
    else if (!strcmp(name, "fail_tail")) e->dbg_fail_tail = value;
    else return set_err("debug_option: unknown option '%s'", name);
    return 0;
}

/* Copy a device buffer of the most recent transducer launch group to the host (everything in flight is drained
 * first): "tb" (one byte per state: [column block][quad][read of tile][state of quad]), "tb_end" (int per column
 * block and read), "final_state" (int per read, tiled order), "final_score" (float per read, tiled order),
 * "final_scores" ([tile][states x 16 reads + 16 start + 16 end] floats; needs the dump_final option),
 * "order" (int per tiled position: index of the read in the call, -1 = padding), "tile_boff" (long long per tile),
 * "n_redo" (unsigned long long: reads k_stitch has left to the host since the engine was created), "gru_tiles" (int:
 * tiles per workgroup of the group's recurrent layers, 1 or 2).
 * Returns the number of bytes the buffer holds (copies min(that, nbytes)), -1 on error. */
extern "C" long long scrappie_hip_debug_fetch(scrappie_hip_engine *e, const char *what, void *dst, size_t nbytes) {
    if (!e || !what) return set_err("debug_fetch: null argument");
    (void)hipSetDevice(e->device);
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipStreamSynchronize(e->cstream));
    const int slot = e->cur;
    const LaunchGroup &lg = e->lgs[slot];
    if (!lg.valid) return set_err("debug_fetch: no launch group has run");
    Model *m = get_model(e, lg.model);
    if (!m) return -1;
    const size_t NH = (size_t)std::max(m->NS - 1, 0);
    const void *src = nullptr; size_t have = 0; bool host = false;
    std::vector<long long> tb;
    const int gru_tiles = lg.gru_two ? 2 : 1;
    if (!strcmp(what, "tb")) { src = e->d_tb.p; have = (size_t)lg.ncb * NH * 16; }
    else if (!strcmp(what, "tb_end")) { src = e->d_tbend.p; have = (size_t)lg.ncb * 16 * 4; }
    else if (!strcmp(what, "final_state")) { src = e->d_fstate.p; have = lg.npad * 4; }
    else if (!strcmp(what, "final_score")) { src = e->d_fscore[slot].p; have = lg.npad * 4; }
    else if (!strcmp(what, "final_scores")) { src = e->d_vstate.p; have = lg.ntile * (NH * 16 + 32) * 4; }
    else if (!strcmp(what, "order")) { src = lg.order.data(); have = lg.npad * 4; host = true; }
    else if (!strcmp(what, "tile_boff")) {
        long long ncb = 0;
        for (size_t t = 0; t < lg.ntile; t++) { tb.push_back(ncb); int mx = 0; for (int k = 0; k < 16; k++) mx = std::max(mx, lg.rT[t * 16 + k]); ncb += mx; }
        src = tb.data(); have = tb.size() * 8; host = true;
    } else if (!strcmp(what, "n_redo")) { static thread_local unsigned long long tot; tot = e->n_redo + e->n_redo_tail; src = &tot; have = 8; host = true; }
    else if (!strcmp(what, "n_tail_groups")) { src = &e->n_tail_groups; have = 8; host = true; }
    else if (!strcmp(what, "n_tail_calls")) { src = &e->n_tail_calls; have = 8; host = true; }
    else if (!strcmp(what, "n_tail_reads")) { src = &e->n_tail_reads; have = 8; host = true; }
    else if (!strcmp(what, "gru_tiles")) { src = &gru_tiles; have = 4; host = true; }
    else if (!strcmp(what, "pinned_bytes")) {      /* pinned host memory this engine holds (staging, results, metadata; both slots) */
        static thread_local unsigned long long tot;
        tot = 0;
        for (int k = 0; k < 2; k++)
            for (const HBuf *h : {&e->h_err[k], &e->h_bad[k], &e->h_edge[k], &e->h_pos[k], &e->h_bases[k], &e->h_blen[k], &e->h_redo[k], &e->h_meta[k],
                                  &e->h_seq[k], &e->h_score[k], &e->h_hp[k], &e->h_sig[k]}) tot += h->cap;
        src = &tot; have = 8; host = true;
    }
    else return set_err("debug_fetch: unknown buffer '%s'", what);
    if (!src && have) return set_err("debug_fetch: buffer '%s' was not allocated", what);
    const size_t cnt = std::min(have, nbytes);
    if (dst && cnt) {
        if (host) memcpy(dst, src, cnt);
        else HIPCHK(hipMemcpy(dst, src, cnt, hipMemcpyDeviceToHost));
    }
    return (long long)have;
}

/* k_stitch on ONE read given on the host (test hook: the device form of homopolymer_path + overlapper /
 * crfpath_to_basecall against the compiled-reference fixtures).  path: nblock + 1 entries; side: [nblock][5] log-
 * posterior rows (homopolymer k-mers of A, C, G, T, stay) or NULL (no homopolymer pass); nstate: 4^k + 1, or 25 with
 * crf != 0.  bases gets at most cap bytes incl. the NUL; pos (nblock + 1 ints) and redo (1: the device left the
 * decision to the host) may be NULL.  Returns the number of bases, -1 = no call, -2 = error. */
extern "C" long scrappie_hip_debug_stitch(scrappie_hip_engine *e, const int *path, const float *side, size_t nblock, int nstate, int crf,
                                          char *bases, size_t cap, int *pos, int *redo) {
    if (!e || !path || !bases || nblock == 0) { set_err("debug_stitch: bad argument"); return -2; }
    (void)hipSetDevice(e->device);
    std::lock_guard<std::mutex> lk(e->mu);
    const size_t T = nblock, npad = 64, bcap = 5 * (T + 1) + 8;
    DBuf dmeta, dseq, dhp, dpos, dbases, dblen, dredo;
    long rc = -2;
    do {
        /* metadata of a group of one read: [seq_off | hp_off | bases_off] (long long x npad each), rT (int x npad) */
        std::vector<char> hm(npad * 8 * 3 + npad * 4, 0);
        int *rT = (int *)(hm.data() + npad * 24);
        rT[0] = (int)T;
        if (dmeta.ensure(hm.size()) || dseq.ensure((T + 1) * 4) || dpos.ensure((T + 1) * 4) || dbases.ensure(bcap) ||
            dblen.ensure(npad * 4) || dredo.ensure(npad * 4) || (side && dhp.ensure(T * 5 * 4))) break;
        hipStream_t s = e->stream;
        if (hipMemcpyAsync(dmeta.p, hm.data(), hm.size(), hipMemcpyHostToDevice, s) != hipSuccess) break;
        if (hipMemcpyAsync(dseq.p, path, (T + 1) * 4, hipMemcpyHostToDevice, s) != hipSuccess) break;
        if (side && hipMemcpyAsync(dhp.p, side, T * 5 * 4, hipMemcpyHostToDevice, s) != hipSuccess) break;
        if (hipMemsetAsync(dredo.p, 0, npad * 4, s) != hipSuccess) break;
        if (hipStreamSynchronize(s) != hipSuccess) break;                 /* sources are pageable caller memory */
        char *d = dmeta.as<char>();
        ShMeta md{};
        md.rT = (const int *)(d + npad * 24);
        ShStitchArgs sa;
        sa.seq = dseq.as<int>(); sa.seq_off = (const long long *)d;
        sa.hp = side ? dhp.as<float>() : nullptr; sa.hp_off = (const long long *)(d + npad * 8);
        sa.pos = pos ? dpos.as<int>() : nullptr;
        sa.bases = dbases.as<char>(); sa.bases_off = (const long long *)(d + npad * 16);
        sa.blen = dblen.as<int>(); sa.redo = dredo.as<unsigned>();
        sa.npad = 1; sa.nstate = nstate; sa.crf = crf; sa.sstride = 1;
        hipLaunchKernelGGL(k_stitch, dim3(1), dim3(64), 0, s, sa, md);
        int len = -1; unsigned rd = 0;
        if (hipMemcpyAsync(&len, dblen.p, 4, hipMemcpyDeviceToHost, s) != hipSuccess) break;
        if (hipMemcpyAsync(&rd, dredo.p, 4, hipMemcpyDeviceToHost, s) != hipSuccess) break;
        if (hipStreamSynchronize(s) != hipSuccess) { set_err("debug_stitch: %s", hipGetErrorString(hipGetLastError())); break; }
        if (redo) *redo = (int)rd;
        if (len >= 0) {
            if ((size_t)len + 1 > cap) { set_err("debug_stitch: %d bases do not fit %zu bytes", len, cap); break; }
            if (len && hipMemcpy(bases, dbases.p, (size_t)len, hipMemcpyDeviceToHost) != hipSuccess) break;
            bases[len] = 0;
            if (pos && !crf && hipMemcpy(pos, dpos.p, (T + 1) * 4, hipMemcpyDeviceToHost) != hipSuccess) break;
        }
        rc = len;
    } while (0);
    for (DBuf *b : {&dmeta, &dseq, &dhp, &dpos, &dbases, &dblen, &dredo}) b->release();
    return rc;
}

extern "C" void scrappie_hip_free_calls(scrappie_hip_call *calls, size_t n) {
    if (!calls) return;
    for (size_t i = 0; i < n; i++) { free(calls[i].basecall); free(calls[i].pos); calls[i].basecall = nullptr; calls[i].pos = nullptr; }
}

/* ------------------------------------------------------------------ */
/* single-read surface on an explicit engine                            */
/* ------------------------------------------------------------------ */
static scrappie_matrix gather_to_host(scrappie_hip_engine *e, const float *src, const float *sums, int T, int nr,
                                      int nchunk, int finalize, int want_log, float min_prob) {
    scrappie_matrix M = make_scrappie_matrix((size_t)nr, (size_t)T);
    if (!M) { set_err("out of host memory"); return nullptr; }
    DBuf tmp;
    const size_t bytes = (size_t)T * M->stride * 4;
    if (tmp.ensure(bytes)) { free_scrappie_matrix(M); return nullptr; }
    bool ok = hipMemsetAsync(tmp.p, 0, bytes, e->stream) == hipSuccess;
    const long long tot = (long long)T * nr;
    hipLaunchKernelGGL(k_gather_read, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, e->stream, src, sums, 0LL, 0, T, nr, nchunk,
                       (int)M->stride, finalize, want_log, min_prob, tmp.as<float>());
    ok = ok && hipMemcpyAsync(M->data.f, tmp.p, bytes, hipMemcpyDeviceToHost, e->stream) == hipSuccess;
    ok = ok && hipStreamSynchronize(e->stream) == hipSuccess;
    tmp.release();
    if (!ok) { set_err("gather failed: %s", hipGetErrorString(hipGetLastError())); return free_scrappie_matrix(M); }
    return M;
}

/* single-read surface: did the read of the launch group just run leave the split products' operand range? */
static bool read_out_of_range(scrappie_hip_engine *e) {
    unsigned flag = 0;
    if (hipMemcpyAsync(&flag, e->d_bad[e->cur].p, 4, hipMemcpyDeviceToHost, e->stream) != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess) {
        set_err("reading the range flag failed: %s", hipGetErrorString(hipGetLastError()));
        return true;
    }
    if (flag) set_err("the read holds values outside the supported range (|activation| >= %g after the first layer, or non-finite): "
                      "is the signal trimmed and med/MAD-normalised?", (double)SH_ACT_LIMIT);
    return flag != 0;
}

static int stage_one(scrappie_hip_engine *e, const raw_table &signal, uint64_t &off, uint32_t &len) {
    if (signal.n == 0 || !signal.raw || signal.end <= signal.start) return set_err("empty read");
    const size_t ns = signal.end - signal.start;
    if (e->d_signal[0].ensure(ns * 4)) return -1;
    HIPCHK(hipMemcpyAsync(e->d_signal[0].p, signal.raw + signal.start, ns * 4, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));   /* source is pageable caller memory */
    off = 0; len = (uint32_t)ns;
    return 0;
}

extern "C" scrappie_matrix scrappie_hip_posterior(scrappie_hip_engine *e, int model, const raw_table signal, float min_prob,
                                                  float tempW, float tempb, bool return_log) {
    Model *m = get_model(e, model);
    if (!m) return nullptr;
    (void)hipSetDevice(e->device);
    std::lock_guard<std::mutex> lk(e->mu);
    uint64_t off; uint32_t len;
    if (stage_one(e, signal, off, len)) return nullptr;
    if (m->arch == 3) len /= (uint32_t)m->nfeat;          /* events: raw holds [nevent][12] features */
    if (len < m->min_samples) { set_err("read of %u samples is below the model minimum %zu", len, m->min_samples); return nullptr; }
    scrappie_hip_params p = scrappie_hip_default_params();
    p.min_prob = min_prob; p.tempW = tempW; p.tempb = tempb;
    RunOut ro;
    if (run_pipeline(e, m, e->d_signal[0].as<float>(), &off, &len, 1, &p, STOP_POST, 5, &ro)) return nullptr;
    if (read_out_of_range(e)) return nullptr;
    const int T = e->lgs[e->cur].rT[0];
    if (m->arch != 1) return gather_to_host(e, ro.E, ro.sums, T, m->NS, m->ff_mtiles, 1, return_log ? 1 : 0, min_prob);
    return gather_to_host(e, ro.E, nullptr, T, m->NS, m->ff_mtiles, 0, 0, 0.f);
}

/* Posteriors of several reads in one launch group (the per-read reference surface called from several host threads at once: the coalescer below).
 * Every request gets its own host matrix or its own error text; a read's posterior does not depend on what it was batched with. */
struct PostReq {
    int model = -1;
    raw_table sig{};
    float min_prob = 0.f, tempW = 1.f, tempb = 1.f;
    bool want_log = true;
    scrappie_matrix out = nullptr;
    size_t nr = 0, nc = 0;           /* shape of the matrix to make */
    const float *src = nullptr;      /* the matrix's bytes in the batch's pinned buffer: the caller makes the matrix and copies them itself (all callers at once) */
    size_t nbytes = 0;
    std::atomic<int> *users = nullptr;
    int phase = 0;                   /* sh_coalesce.h: 0 queued ... 3 done */
    char err[256] = "";
};
struct PostStage { HBuf h; DBuf d; std::atomic<int> users{0}; };
static void post_fail(PostReq *r, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(r->err, sizeof r->err, fmt, ap);
    va_end(ap);
}
static void posterior_batch(scrappie_hip_engine *e, std::vector<PostReq *> &reqs, PostStage *stage) {
    if (reqs.empty()) return;
    Model *m = get_model(e, reqs[0]->model);
    if (!m) { for (PostReq *r : reqs) post_fail(r, "%s", g_err); return; }
    (void)hipSetDevice(e->device);
    std::lock_guard<std::mutex> lk(e->mu);
    std::vector<PostReq *> live;
    std::vector<uint64_t> off;
    std::vector<uint32_t> len;
    size_t total = 0;
    for (PostReq *r : reqs) {
        const raw_table &sg = r->sig;
        if (sg.n == 0 || !sg.raw || sg.end <= sg.start) { post_fail(r, "empty read"); continue; }
        const size_t nf = sg.end - sg.start, ns = m->arch == 3 ? nf / (size_t)m->nfeat : nf;      /* events: raw holds [nevent][12] features */
        if (ns < m->min_samples) { post_fail(r, "read of %zu samples is below the model minimum %zu", ns, m->min_samples); continue; }
        live.push_back(r); off.push_back(total); len.push_back((uint32_t)ns);
        total += nf;
    }
    if (live.empty()) return;
    auto fail_all = [&]() { for (PostReq *r : live) if (!r->src) post_fail(r, "%s", g_err); };
    if (e->h_sig[0].ensure(total * 4) || e->d_signal[0].ensure(total * 4)) { fail_all(); return; }
    float *hs = e->h_sig[0].as<float>();
    for (size_t i = 0; i < live.size(); i++) memcpy(hs + off[i], live[i]->sig.raw + live[i]->sig.start, (live[i]->sig.end - live[i]->sig.start) * 4);
    /* (the group's prologue -- the convolution -- runs on another stream: the signals must be there before it is enqueued) */
    if (hipMemcpyAsync(e->d_signal[0].p, hs, total * 4, hipMemcpyHostToDevice, e->stream) != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess) {
        set_err("upload failed: %s", hipGetErrorString(hipGetLastError())); fail_all(); return;
    }
    scrappie_hip_params p = scrappie_hip_default_params();
    p.tempW = live[0]->tempW; p.tempb = live[0]->tempb;
    RunOut ro;
    if (run_pipeline(e, m, e->d_signal[0].as<float>(), off.data(), len.data(), live.size(), &p, STOP_POST, 5, &ro)) { fail_all(); return; }
    const LaunchGroup &lg = e->lgs[e->cur];
    std::vector<unsigned> bad(lg.npad, 0);
    if (hipMemcpyAsync(bad.data(), e->d_bad[e->cur].p, lg.npad * 4, hipMemcpyDeviceToHost, e->stream) != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess) {
        set_err("reading the range flags failed: %s", hipGetErrorString(hipGetLastError())); fail_all(); return;
    }
    std::vector<long long> tile_boff(lg.ntile, 0);
    { long long ncb = 0; for (size_t t = 0; t < lg.ntile; t++) { int tt = 0; for (int b = 0; b < 16; b++) tt = std::max(tt, lg.rT[t * 16 + b]); tile_boff[t] = ncb; ncb += tt; } }
    /* one staging buffer for the whole group, one synchronisation */
    std::vector<size_t> toff(lg.npad, 0);
    size_t tbytes = 0;
    const size_t mstride = (size_t)((m->NS + 3) / 4) * 4;      /* scrappie_matrix.c:11-42: rows padded to whole vectors */
    for (size_t i = 0; i < lg.npad; i++) {
        const int o = lg.order[i];
        if (o < 0 || lg.rT[i] <= 0) continue;
        PostReq *r = live[(size_t)o];
        if (bad[i]) { post_fail(r, "the read holds values outside the supported range (|activation| >= %g after the first layer, or non-finite): is the signal trimmed and med/MAD-normalised?", (double)SH_ACT_LIMIT); continue; }
        /* (the matrix itself -- 3.3 MB to allocate and clear for a read of 4000 samples -- is made by the caller, beside all the others) */
        r->nr = m->NS; r->nc = lg.rT[i];
        toff[i] = tbytes; tbytes += (size_t)lg.rT[i] * mstride * 4;
    }
    DBuf &tmp = stage->d;            /* (grow-only, kept between launch groups) */
    bool ok = tbytes == 0 || tmp.ensure(tbytes) == 0;
    ok = ok && (tbytes == 0 || hipMemsetAsync(tmp.p, 0, tbytes, e->stream) == hipSuccess);
    /* the buffer's last batch has been copied out by its callers (two buffers in turn: nearly always long ago) */
    while (stage->users.load(std::memory_order_acquire) > 0) std::this_thread::yield();
    ok = ok && (tbytes == 0 || stage->h.ensure(tbytes) == 0);
    for (size_t i = 0; ok && i < lg.npad; i++) {
        const int o = lg.order[i];
        if (o < 0 || lg.rT[i] <= 0 || live[(size_t)o]->nc == 0) continue;
        PostReq *r = live[(size_t)o];
        const int T = lg.rT[i];
        const long long tot = (long long)T * m->NS;
        float *dst = (float *)((char *)tmp.p + toff[i]);
        const bool tr = m->arch != 1;
        hipLaunchKernelGGL(k_gather_read, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, e->stream, ro.E, tr ? ro.sums : nullptr, tile_boff[i >> 4], (int)(i & 15), T, m->NS,
                           m->ff_mtiles, (int)mstride, tr ? 1 : 0, tr && r->want_log ? 1 : 0, tr ? r->min_prob : 0.f, dst);
    }
    /* one copy into pinned memory at the link's rate; the callers take their matrices out of it themselves, all at once */
    ok = ok && (tbytes == 0 || hipMemcpyAsync(stage->h.p, tmp.p, tbytes, hipMemcpyDeviceToHost, e->stream) == hipSuccess);
    ok = ok && hipStreamSynchronize(e->stream) == hipSuccess;
    if (!ok) {
        set_err("gather failed: %s", hipGetErrorString(hipGetLastError()));
        for (PostReq *r : live) if (r->nc) { r->nc = 0; post_fail(r, "%s", g_err); }
        return;
    }
    int nuse = 0;
    for (size_t i = 0; i < lg.npad; i++) {
        const int o = lg.order[i];
        if (o < 0 || lg.rT[i] <= 0 || live[(size_t)o]->nc == 0) continue;
        PostReq *r = live[(size_t)o];
        r->src = (const float *)((const char *)stage->h.p + toff[i]);
        r->nbytes = (size_t)lg.rT[i] * mstride * 4;
        r->users = &stage->users;
        nuse++;
    }
    stage->users.store(nuse, std::memory_order_release);
}

/* The reference calls its network functions from an OpenMP loop over reads (scrappie_raw.c:355,387), one read per call.  One read cannot fill the
 * device -- its five recurrent layers are a serial chain of T steps each -- so calls that arrive while the device is busy (or within a short window
 * of the first) are run as ONE launch group: whoever finds no batch running becomes its leader, takes every waiting request for the same model and
 * temperatures, runs them, hands the matrices out and wakes the others.  SCRAPPIE_HIP_COALESCE=0: every call runs alone, as before;
 * SCRAPPIE_HIP_COALESCE_US / _MAX_US: the leader's waiting windows (defaults 500 / 10000 microseconds); the queue itself is sh_coalesce.h. */
/* how many host threads are inside the coalesced functions (either of them), and the most seen lately: a process that calls from ONE thread must not
 * wait for company that cannot come, one that calls from many should */
static ShPresence g_presence;           /* threads inside the per-read functions: how much company a leader waits for (sh_coalesce.h) */
struct Coalescer : ShCoalescer<PostReq> { PostStage stage[2]; };
static Coalescer g_co;
static bool coalesce_on(char which = 'p') {       /* SCRAPPIE_HIP_COALESCE: 0 neither, p the network calls only, d decode_transducer only; default both */
    static const int on = [] { const char *v = getenv("SCRAPPIE_HIP_COALESCE"); return !v ? 3 : v[0] == '0' ? 0 : v[0] == 'p' ? 1 : v[0] == 'd' ? 2 : 3; }();
    return (on & (which == 'd' ? 2 : 1)) != 0;
}
static scrappie_matrix coalesced_posterior(scrappie_hip_engine *e, int model, const raw_table signal, float min_prob, float tempW, float tempb, bool return_log) {
    constexpr size_t MAX_READS = 4096, MAX_BLOCKS = 200000;          /* per launch group: the posterior is materialised (66 KB per block of 16 reads) */
    ShInside inside(g_presence);
    PostReq r;
    r.model = model; r.sig = signal; r.min_prob = min_prob; r.tempW = tempW; r.tempb = tempb; r.want_log = return_log;
    PostStage *stage = nullptr;
    g_co.run(r, g_presence, MAX_READS, false,
        [&](std::deque<PostReq *> &q, std::vector<PostReq *> &batch) {
            const PostReq *f = q.front();
            size_t blocks = 0;
            for (auto it = q.begin(); it != q.end() && batch.size() < MAX_READS;) {
                PostReq *c = *it;
                const size_t b = (c->sig.end > c->sig.start ? c->sig.end - c->sig.start : 0) / 4 + 1;
                if (c->model == f->model && c->tempW == f->tempW && c->tempb == f->tempb && (batch.empty() || blocks + b <= MAX_BLOCKS)) {
                    batch.push_back(c); blocks += b; it = q.erase(it);
                } else ++it;
            }
            stage = &g_co.stage[g_co.n_batches & 1];
        },
        [](std::vector<PostReq *> &) { return true; }, [](PostReq &) {},
        [&](std::vector<PostReq *> &batch) { posterior_batch(e, batch, stage); });
    if (r.src) {
        r.out = make_scrappie_matrix(r.nr, r.nc);
        if (r.out) memcpy(r.out->data.f, r.src, r.nbytes);
        else post_fail(&r, "out of host memory");
        r.users->fetch_sub(1, std::memory_order_release);
    }
    if (!r.out && r.err[0]) set_err("%s", r.err);
    return r.out;
}
/* (tests, tools) launch groups the coalescer has run, reads in them, the largest group */
extern "C" void scrappie_hip_coalescer_stats(unsigned long long out[3]) {
    std::lock_guard<std::mutex> lk(g_co.mu);
    out[0] = g_co.n_batches; out[1] = g_co.n_reads; out[2] = g_co.max_batch;
    if (getenv("SCRAPPIE_HIP_COALESCE_TIMES")) fprintf(stderr, "posterior coalescer: %llu launch groups took %.1f ms in all\n", g_co.n_batches, g_co.service_us / 1e3);
}

extern "C" scrappie_matrix scrappie_hip_trunk(scrappie_hip_engine *e, int model, const raw_table signal, int upto) {
    Model *m = get_model(e, model);
    if (!m) return nullptr;
    (void)hipSetDevice(e->device);
    std::lock_guard<std::mutex> lk(e->mu);
    uint64_t off; uint32_t len;
    if (stage_one(e, signal, off, len)) return nullptr;
    if (m->arch == 3) len /= (uint32_t)m->nfeat;
    if (len < m->min_samples) { set_err("read too short"); return nullptr; }
    scrappie_hip_params p = scrappie_hip_default_params();
    RunOut ro;
    if (run_pipeline(e, m, e->d_signal[0].as<float>(), &off, &len, 1, &p, STOP_TRUNK, upto, &ro)) return nullptr;
    if (read_out_of_range(e)) return nullptr;
    return gather_to_host(e, ro.act, nullptr, e->lgs[e->cur].rT[0], ro.act_units, ro.act_units / 16, 0, 0, 0.f);
}

/* ------------------------------------------------------------------ */
/* per-read reference surface on the process-default engine             */
/* ------------------------------------------------------------------ */
static scrappie_hip_engine *g_default = nullptr;
static std::mutex g_default_mu;

static scrappie_hip_engine *default_engine() {
    std::lock_guard<std::mutex> lk(g_default_mu);
    if (!g_default) {
        const char *dev = getenv("SCRAPPIE_HIP_DEVICE");
        g_default = scrappie_hip_engine_create(dev ? atoi(dev) : 0);
    }
    return g_default;
}

/* name -> handle on the process-default engine, kept OUTSIDE the engine's own lock: a launch group holds that lock for its whole run, and a caller
 * of the per-read surface that had to take it just to look its model up could not join the queue until the group was over (64 threads then arrive
 * one by one behind every group, and every group holds two reads) */
static std::mutex g_models_mu;
static std::map<std::string, int> g_models;

extern "C" int scrappie_hip_register_model(const char *name, const char *path) {
    scrappie_hip_engine *e = default_engine();
    if (!e || !name) return -1;
    std::lock_guard<std::mutex> lk(g_models_mu);
    const int h = scrappie_hip_load_model(e, name, path);
    if (h >= 0) g_models[name] = h;
    return h;
}

static int default_model(scrappie_hip_engine *e, const char *name) {
    /* find-or-load as one step: the reference's loop calls in from many threads at once, and a second load of the same name would replace -- and
     * free -- the model the first caller is already running */
    std::lock_guard<std::mutex> lk(g_models_mu);
    auto it = g_models.find(name);
    if (it != g_models.end()) return it->second;
    int h = scrappie_hip_find_model(e, name);
    if (h < 0) {
        const char *dir = getenv("SCRAPPIE_MODEL_DIR");
        if (!dir) { set_err("model '%s' is not registered and SCRAPPIE_MODEL_DIR is unset (weights are not compiled in)", name); return -1; }
        std::string path = std::string(dir) + "/" + name + ".scrm";
        h = scrappie_hip_load_model(e, name, path.c_str());
    }
    if (h >= 0) g_models[name] = h;
    return h;
}

static scrappie_matrix named_posterior(const char *name, const raw_table signal, float min_prob, float tempW, float tempb, bool return_log) {
    if (signal.n == 0 || !signal.raw) return nullptr;                 /* networks.c:254-255 */
    scrappie_hip_engine *e = default_engine();
    if (!e) return nullptr;
    const int h = default_model(e, name);
    if (h < 0) return nullptr;
    if (coalesce_on()) return coalesced_posterior(e, h, signal, min_prob, tempW, tempb, return_log);
    return scrappie_hip_posterior(e, h, signal, min_prob, tempW, tempb, return_log);
}

extern "C" scrappie_matrix scrappie_hip_events_posterior(scrappie_hip_engine *e, int model, const float *feature3, size_t nevent,
                                                         float min_prob, float tempW, float tempb, bool return_log) {
    Model *m = get_model(e, model);
    if (!m) return nullptr;
    if (m->arch != 3) { set_err("events_posterior: model '%s' is not an events model", m->name.c_str()); return nullptr; }
    if (!feature3 || nevent == 0) { set_err("events_posterior: empty read"); return nullptr; }
    raw_table rt = {nullptr, nevent * (size_t)m->nfeat, 0, nevent * (size_t)m->nfeat, const_cast<float *>(feature3)};
    return scrappie_hip_posterior(e, model, rt, min_prob, tempW, tempb, return_log);
}

/* networks.c:146: features + window on the host, the network on the device */
extern "C" scrappie_matrix nanonet_posterior(const event_table events, float min_prob, float tempW, float tempb, bool return_log) {
    if (events.n == 0 || !events.event || events.end <= events.start) return nullptr;
    scrappie_hip_engine *e = default_engine();
    if (!e) return nullptr;
    const int model = default_model(e, "nanonet_events");
    if (model < 0) return nullptr;
    const size_t n = events.end - events.start;
    std::vector<float> f3(n * 12);
    if (scrappie_hip_event_features(events, f3.data())) return nullptr;
    if (coalesce_on()) {
        raw_table rt = {nullptr, n * 12, 0, n * 12, f3.data()};
        return coalesced_posterior(e, model, rt, min_prob, tempW, tempb, return_log);
    }
    return scrappie_hip_events_posterior(e, model, f3.data(), n, min_prob, tempW, tempb, return_log);
}

extern "C" scrappie_matrix nanonet_raw_posterior(const raw_table s, float mp, float tw, float tb, bool lg) { return named_posterior("raw_r94", s, mp, tw, tb, lg); }
extern "C" scrappie_matrix nanonet_rgrgr_r94_posterior(const raw_table s, float mp, float tw, float tb, bool lg) { return named_posterior("rgrgr_r94", s, mp, tw, tb, lg); }
extern "C" scrappie_matrix nanonet_rgrgr_r941_posterior(const raw_table s, float mp, float tw, float tb, bool lg) { return named_posterior("rgrgr_r941", s, mp, tw, tb, lg); }
extern "C" scrappie_matrix nanonet_rgrgr_r10_posterior(const raw_table s, float mp, float tw, float tb, bool lg) { return named_posterior("rgrgr_r10", s, mp, tw, tb, lg); }
extern "C" scrappie_matrix nanonet_rnnrf_r94_transitions(const raw_table s, float mp, float tw, float tb, bool lg) { return named_posterior("rnnrf_r94", s, mp, tw, tb, lg); }

extern "C" posterior_function_ptr get_posterior_function(const enum raw_model_type model) {
    switch (model) {
    case SCRAPPIE_MODEL_RAW: return nanonet_raw_posterior;
    case SCRAPPIE_MODEL_RGRGR_R9_4: return nanonet_rgrgr_r94_posterior;
    case SCRAPPIE_MODEL_RGRGR_R9_4_1: return nanonet_rgrgr_r941_posterior;
    case SCRAPPIE_MODEL_RGRGR_R10: return nanonet_rgrgr_r10_posterior;
    case SCRAPPIE_MODEL_RNNRF_R9_4: return nanonet_rnnrf_r94_transitions;
    default:
        /* the reference errx()'s on an invalid enum (networks.c:120-123) */
        fprintf(stderr, "scrappie_hip: model enum %d has no posterior function\n", (int)model);
        exit(EXIT_FAILURE);
    }
}

extern "C" int get_raw_model_stride(const enum raw_model_type model) {
    scrappie_hip_engine *e = default_engine();
    if (!e) return -1;
    const int h = default_model(e, raw_model_string(model));
    return h < 0 ? -1 : e->models[h]->stride;
}

extern "C" int get_raw_model_stride_from_string(const char *modelstr) {      /* python/build.py:34-44 */
    const enum raw_model_type t = get_raw_model(modelstr);
    if (t == SCRAPPIE_MODEL_INVALID) return -1;
    return get_raw_model_stride(t);
}

/* decode_transducer from many host threads at once (the other half of the reference's per-read loop body, scrappie_raw.c:279-287): the calls that are
 * waiting run as one launch -- one workgroup per read, each exactly the single-read form below (a tile whose lanes alias the read's columns), so a call's
 * path and score are the ones it gets alone.  The posteriors are 3.3 MB per read of 4000 samples: every caller copies its own into pinned memory (all at
 * once), one transfer takes them to the device. */
struct DecReq {
    const_scrappie_matrix post = nullptr;
    float stay_pen = 0.f, skip_pen = 0.f, local_pen = 0.f;
    bool slip = false;
    int *seq = nullptr;
    float score = NAN;
    int phase = 0;                  /* 0 queued, 1 asked to copy its posterior to `dst`, 2 copied, 3 done */
    float *dst = nullptr;
};
struct DecCoalescer : ShCoalescer<DecReq> {
    HBuf stage, hseq;
    DBuf d[7];
};
static DecCoalescer g_dc;

/* the batch's posteriors are in g_dc.stage (concatenated, read k at column offset boff[k]); results into the requests */
static void decode_batch(scrappie_hip_engine *e, std::vector<DecReq *> &reqs, const std::vector<long long> &boff, long long ncb) {
    const size_t n = reqs.size();
    const DecReq *f = reqs[0];
    const int NH = (int)f->post->nr - 1, NQ = NH / 4;
    const size_t stride = f->post->stride;
    (void)hipSetDevice(e->device);
    std::lock_guard<std::mutex> lk(e->mu);
    hipStream_t s = e->stream;
    const size_t npad = 16 * n;
    /* metadata: sig_off[npad] | seq_off[npad] | (unused)[npad] | tile_boff[n] | rN[npad] | rT[npad] | tile_T[n] */
    std::vector<char> hm(npad * 24 + n * 8 + npad * 8 + n * 4, 0);
    long long *seq_off = (long long *)(hm.data() + npad * 8);
    long long *tboff = (long long *)(hm.data() + npad * 24);
    int *rN = (int *)(hm.data() + npad * 24 + n * 8), *rT = rN + npad, *tT = rT + npad;
    long long nseq = 0;
    for (size_t k = 0; k < n; k++) {
        const int T = (int)reqs[k]->post->nc;
        tboff[k] = boff[k]; tT[k] = T;
        /* every lane of the tile reads the same columns (strideB = 0) and runs the read, as in the single-read form; lane 0's path is walked back */
        for (int b = 0; b < 16; b++) { rN[k * 16 + b] = 1; rT[k * 16 + b] = T; seq_off[k * 16 + b] = nseq; }
        nseq += T + 1;
    }
    /* (grow-only, kept between launches: seven hipMalloc / hipFree pairs per launch cost more than the transfer) */
    DBuf &dmeta = g_dc.d[0], &dpost = g_dc.d[1], &dtb = g_dc.d[2], &dtbe = g_dc.d[3], &dfs = g_dc.d[4], &dfsc = g_dc.d[5], &dseq = g_dc.d[6];
    bool ok = false;
    do {
        const size_t pbytes = (size_t)ncb * stride * 4;
        if (dmeta.ensure(hm.size()) || dpost.ensure(pbytes) || dtb.ensure((size_t)ncb * NQ * 16 * 4) || dtbe.ensure((size_t)ncb * 16 * 4) ||
            dfs.ensure(npad * 4) || dfsc.ensure(npad * 4) || dseq.ensure((size_t)nseq * 4) || g_dc.hseq.ensure((size_t)nseq * 4 + npad * 4)) break;
        if (hipMemcpyAsync(dmeta.p, hm.data(), hm.size(), hipMemcpyHostToDevice, s) != hipSuccess) break;
        if (hipMemcpyAsync(dpost.p, g_dc.stage.p, pbytes, hipMemcpyHostToDevice, s) != hipSuccess) break;
        char *d = dmeta.as<char>();
        ShMeta md;
        md.sig_off = (const unsigned long long *)d;
        md.tile_boff = (const long long *)(d + npad * 24);
        md.rN = (const int *)(d + npad * 24 + n * 8);
        md.rT = md.rN + npad;
        md.tile_T = md.rT + npad;
        ShVitArgs va;
        va.E = dpost.as<float>(); va.sums = nullptr;
        va.strideT = (long long)stride; va.strideQ = 4; va.strideB = 0;
        va.want_log = 0; va.min_prob = 0.f;
        va.stay_pen = f->stay_pen; va.skip_pen = f->skip_pen; va.local_pen = f->local_pen; va.use_slip = f->slip ? 1 : 0;
        va.tb = dtb.as<unsigned>(); va.tb_end = dtbe.as<int>();
        va.final_state = dfs.as<int>(); va.final_score = dfsc.as<float>();
        va.hp_side = nullptr; va.hp_off = nullptr; va.dbg = nullptr; va.dump_final = 0;
        va.seg = nullptr; va.vstate = nullptr; va.flag = nullptr; va.err = nullptr;   /* one workgroup per read, the whole tile */
        if (launch_viterbi(s, NH, va, md, n)) break;
        /* lane 0 of every tile: the other lanes hold the same path */
        hipLaunchKernelGGL(k_backtrace_lane0, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, dtb.as<unsigned>(), dtbe.as<int>(), dfs.as<int>(), md,
                           (const long long *)(d + npad * 8), dseq.as<int>(), (int)n, NQ);
        int *hseq = g_dc.hseq.as<int>();
        if (hipMemcpyAsync(hseq, dseq.p, (size_t)nseq * 4, hipMemcpyDeviceToHost, s) != hipSuccess) break;
        if (hipMemcpyAsync(hseq + nseq, dfsc.p, npad * 4, hipMemcpyDeviceToHost, s) != hipSuccess) break;
        if (hipStreamSynchronize(s) != hipSuccess) break;
        const float *hsc = (const float *)(hseq + nseq);
        for (size_t k = 0; k < n; k++) {
            memcpy(reqs[k]->seq, hseq + seq_off[k * 16], ((size_t)reqs[k]->post->nc + 1) * 4);
            reqs[k]->score = hsc[k * 16];
        }
        ok = true;
    } while (0);
    if (!ok) { (void)hipGetLastError(); for (DecReq *r : reqs) r->score = NAN; }
}

static float coalesced_decode(scrappie_hip_engine *e, const_scrappie_matrix logpost, float stay_pen, float skip_pen, float local_pen, int *seq, bool allow_slip) {
    constexpr size_t MAX_READS = 1024;
    constexpr long long MAX_BLOCKS = 200000;                          /* 16 KB of traceback + 4 KB of posterior per block */
    ShInside inside(g_presence);
    DecReq r;
    r.post = logpost; r.stay_pen = stay_pen; r.skip_pen = skip_pen; r.local_pen = local_pen; r.slip = allow_slip; r.seq = seq;
    std::vector<long long> boff;
    long long ncb = 0;
    g_dc.run(r, g_presence, MAX_READS, true,
        [&](std::deque<DecReq *> &q, std::vector<DecReq *> &batch) {
            const DecReq *f = q.front();
            boff.clear(); ncb = 0;
            for (auto it = q.begin(); it != q.end() && batch.size() < MAX_READS;) {
                DecReq *c = *it;
                const bool same = c->post->nr == f->post->nr && c->post->stride == f->post->stride && c->stay_pen == f->stay_pen && c->skip_pen == f->skip_pen &&
                                  c->local_pen == f->local_pen && c->slip == f->slip;
                if (same && (batch.empty() || ncb + (long long)c->post->nc <= MAX_BLOCKS)) {
                    batch.push_back(c); boff.push_back(ncb); ncb += (long long)c->post->nc; it = q.erase(it);
                } else ++it;
            }
        },
        [&](std::vector<DecReq *> &batch) {      /* every member copies its own posterior into the launch's pinned buffer, all at once */
            const size_t stride = batch[0]->post->stride;
            if (g_dc.stage.ensure((size_t)ncb * stride * 4) != 0) return false;
            for (size_t k = 0; k < batch.size(); k++) batch[k]->dst = g_dc.stage.as<float>() + (size_t)boff[k] * stride;
            return true;
        },
        [](DecReq &c) { memcpy(c.dst, c.post->data.f, (size_t)c.post->nc * c.post->stride * 4); },
        [&](std::vector<DecReq *> &batch) { decode_batch(e, batch, boff, ncb); });
    return r.score;
}
extern "C" void scrappie_hip_decode_coalescer_stats(unsigned long long out[3]) {
    std::lock_guard<std::mutex> lk(g_dc.mu);
    out[0] = g_dc.n_batches; out[1] = g_dc.n_reads; out[2] = g_dc.max_batch;
    if (getenv("SCRAPPIE_HIP_COALESCE_TIMES")) fprintf(stderr, "decode coalescer: %llu launches took %.1f ms in all\n", g_dc.n_batches, g_dc.service_us / 1e3);
}

/* decode.c:123 on a host posterior: one read = one tile, every lane of the tile
 * aliases the same column data (strideB = 0). */
extern "C" float decode_transducer(const_scrappie_matrix logpost, float stay_pen, float skip_pen, float local_pen, int *seq, bool allow_slip) {
    if (!logpost || !seq) return NAN;
    scrappie_hip_engine *e = default_engine();
    if (!e) return NAN;
    const int NH = (int)logpost->nr - 1, T = (int)logpost->nc;
    if (NH % 64 != 0 || (allow_slip && NH % 256 != 0) || T <= 0) return NAN;
    if (NH != 64 && NH != 256 && NH != 1024) return NAN;
    if (coalesce_on('d')) return coalesced_decode(e, logpost, stay_pen, skip_pen, local_pen, seq, allow_slip);
    (void)hipSetDevice(e->device);
    std::lock_guard<std::mutex> lk(e->mu);
    hipStream_t s = e->stream;
    const int NQ = NH / 4;
    const size_t pbytes = (size_t)T * logpost->stride * 4;
    /* metadata for one tile whose 16 lanes all run the same read */
    const size_t npad = 16;
    std::vector<char> hm(npad * 8 * 3 + 8 + npad * 4 * 2 + 4, 0);
    long long *seq_off = (long long *)(hm.data() + npad * 8);
    int *rN = (int *)(hm.data() + npad * 24 + 8), *rT = rN + npad, *tT = rT + npad;
    for (size_t i = 0; i < npad; i++) { rN[i] = 1; rT[i] = T; seq_off[i] = 0; }
    *tT = T;
    DBuf dmeta, dpost, dtb, dtbe, dfs, dfsc, dseq;
    float score = NAN;
    do {
        if (dmeta.ensure(hm.size()) || dpost.ensure(pbytes) || dtb.ensure((size_t)T * NQ * 16 * 4) || dtbe.ensure((size_t)T * 16 * 4) ||
            dfs.ensure(64) || dfsc.ensure(64) || dseq.ensure(((size_t)T + 1) * 4)) break;
        if (hipMemcpyAsync(dmeta.p, hm.data(), hm.size(), hipMemcpyHostToDevice, s) != hipSuccess) break;
        if (hipMemcpyAsync(dpost.p, logpost->data.f, pbytes, hipMemcpyHostToDevice, s) != hipSuccess) break;
        char *d = dmeta.as<char>();
        ShMeta md;
        md.sig_off = (const unsigned long long *)d;
        md.tile_boff = (const long long *)(d + npad * 24);
        md.rN = (const int *)(d + npad * 24 + 8);
        md.rT = md.rN + npad;
        md.tile_T = md.rT + npad;
        ShVitArgs va;
        va.E = dpost.as<float>(); va.sums = nullptr;
        va.strideT = (long long)logpost->stride; va.strideQ = 4; va.strideB = 0;
        va.want_log = 0; va.min_prob = 0.f;
        va.stay_pen = stay_pen; va.skip_pen = skip_pen; va.local_pen = local_pen; va.use_slip = allow_slip ? 1 : 0;
        va.tb = dtb.as<unsigned>(); va.tb_end = dtbe.as<int>();
        va.final_state = dfs.as<int>(); va.final_score = dfsc.as<float>();
        va.hp_side = nullptr; va.hp_off = nullptr; va.dbg = nullptr; va.dump_final = 0;
        va.seg = nullptr; va.vstate = nullptr; va.flag = nullptr; va.err = nullptr;   /* one workgroup, the whole tile */
        if (launch_viterbi(s, NH, va, md, 1)) break;
        hipLaunchKernelGGL(k_backtrace, dim3(1), dim3(64), 0, s, dtb.as<unsigned>(), dtbe.as<int>(), dfs.as<int>(), md,
                           (const long long *)(d + npad * 8), dseq.as<int>(), 1, NQ, 1);
        if (hipMemcpyAsync(seq, dseq.p, ((size_t)T + 1) * 4, hipMemcpyDeviceToHost, s) != hipSuccess) break;
        float sc = NAN;
        if (hipMemcpyAsync(&sc, dfsc.p, 4, hipMemcpyDeviceToHost, s) != hipSuccess) break;
        if (hipStreamSynchronize(s) != hipSuccess) break;
        score = sc;
    } while (0);
    for (DBuf *b : {&dmeta, &dpost, &dtb, &dtbe, &dfs, &dfsc, &dseq}) b->release();
    return score;
}

/* decode.c:836 on a host transition matrix.  The kernel works on the chunked
 * layout, so the 25 rows are re-laid on the host first. */
__global__ void k_crf_viterbi_only(const float *__restrict__ trans, int stride, int T, unsigned *__restrict__ tbbuf,
                                   int *__restrict__ path, float *__restrict__ score) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float prev[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, curr[5];
    for (int t = 0; t < T; t++) {
        const float *tr = trans + (long long)t * stride;
        unsigned pack = 0;
        for (int to = 0; to < 5; to++) {
            float best = tr[to * 5] + prev[0];
            unsigned from = 0;
            for (int fr = 1; fr < 5; fr++) {
                const float sc = tr[to * 5 + fr] + prev[fr];
                if (sc > best) { best = sc; from = fr; }
            }
            curr[to] = best;
            pack |= from << (3 * to);
        }
        tbbuf[t] = pack;
        for (int i = 0; i < 5; i++) prev[i] = curr[i];
    }
    float best = prev[0];
    int arg = 0;
    for (int i = 1; i < 5; i++) if (prev[i] > best) { best = prev[i]; arg = i; }
    *score = best;
    path[T] = arg;
    for (int blk = T; blk > 0; blk--) { arg = (tbbuf[blk - 1] >> (3 * arg)) & 7u; path[blk - 1] = arg; }
}

/* the same recursion for many reads at once, a thread per read (decode_crf from many host threads: coalesced like decode_transducer) */
__global__ void k_crf_viterbi_batch(const float *__restrict__ trans, int stride, const long long *__restrict__ coff, const int *__restrict__ Ts, int n,
                                    unsigned *__restrict__ tbbuf, int *__restrict__ path, float *__restrict__ score) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int T = Ts[r];
    const float *tr0 = trans + coff[r] * stride;
    unsigned *tb = tbbuf + coff[r];
    int *pth = path + coff[r] + r;                     /* T + 1 entries per read */
    float prev[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, curr[5];
    for (int t = 0; t < T; t++) {
        const float *tr = tr0 + (long long)t * stride;
        unsigned pack = 0;
        for (int to = 0; to < 5; to++) {
            float best = tr[to * 5] + prev[0];
            unsigned from = 0;
            for (int fr = 1; fr < 5; fr++) {
                const float sc = tr[to * 5 + fr] + prev[fr];
                if (sc > best) { best = sc; from = fr; }
            }
            curr[to] = best;
            pack |= from << (3 * to);
        }
        tb[t] = pack;
        for (int i = 0; i < 5; i++) prev[i] = curr[i];
    }
    float best = prev[0];
    int arg = 0;
    for (int i = 1; i < 5; i++) if (prev[i] > best) { best = prev[i]; arg = i; }
    score[r] = best;
    pth[T] = arg;
    for (int blk = T; blk > 0; blk--) { arg = (tb[blk - 1] >> (3 * arg)) & 7u; pth[blk - 1] = arg; }
}

struct CrfReq { const_scrappie_matrix trans = nullptr; int *path = nullptr; float score = NAN; int phase = 0; };
struct CrfCoalescer : ShCoalescer<CrfReq> {
    HBuf hin, hout;
    DBuf d[4];
};
static CrfCoalescer g_cc;

static void crf_batch(scrappie_hip_engine *e, std::vector<CrfReq *> &reqs) {
    const size_t n = reqs.size(), stride = reqs[0]->trans->stride;
    long long ncol = 0;
    for (CrfReq *r : reqs) ncol += (long long)r->trans->nc;
    /* pinned input: [transitions of all reads][column offset per read][T per read] */
    const size_t fbytes = (size_t)ncol * stride * 4, in_bytes = fbytes + n * 8 + n * 4;
    (void)hipSetDevice(e->device);
    std::lock_guard<std::mutex> lk(e->mu);
    DBuf &din = g_cc.d[0], &dtb = g_cc.d[1], &dpath = g_cc.d[2], &dsc = g_cc.d[3];
    bool ok = false;
    do {
        if (g_cc.hin.ensure(in_bytes) || g_cc.hout.ensure(((size_t)ncol + n) * 4 + n * 4) || din.ensure(in_bytes) || dtb.ensure((size_t)ncol * 4) ||
            dpath.ensure(((size_t)ncol + n) * 4) || dsc.ensure(n * 4)) break;
        char *h = g_cc.hin.as<char>();
        long long *coff = (long long *)(h + fbytes);
        int *Ts = (int *)(h + fbytes + n * 8);
        long long c = 0;
        for (size_t k = 0; k < n; k++) {
            const size_t T = reqs[k]->trans->nc;
            memcpy(h + (size_t)c * stride * 4, reqs[k]->trans->data.f, T * stride * 4);
            coff[k] = c; Ts[k] = (int)T; c += (long long)T;
        }
        hipStream_t s = e->stream;
        if (hipMemcpyAsync(din.p, h, in_bytes, hipMemcpyHostToDevice, s) != hipSuccess) break;
        const char *d = din.as<char>();
        hipLaunchKernelGGL(k_crf_viterbi_batch, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, (const float *)d, (int)stride, (const long long *)(d + fbytes),
                           (const int *)(d + fbytes + n * 8), (int)n, dtb.as<unsigned>(), dpath.as<int>(), dsc.as<float>());
        int *hp = g_cc.hout.as<int>();
        if (hipMemcpyAsync(hp, dpath.p, ((size_t)ncol + n) * 4, hipMemcpyDeviceToHost, s) != hipSuccess) break;
        if (hipMemcpyAsync(hp + ncol + n, dsc.p, n * 4, hipMemcpyDeviceToHost, s) != hipSuccess) break;
        if (hipStreamSynchronize(s) != hipSuccess) break;
        const float *hs = (const float *)(hp + ncol + n);
        for (size_t k = 0; k < n; k++) {
            memcpy(reqs[k]->path, hp + coff[k] + (long long)k, ((size_t)Ts[k] + 1) * 4);
            reqs[k]->score = hs[k];
        }
        ok = true;
    } while (0);
    if (!ok) { (void)hipGetLastError(); for (CrfReq *r : reqs) r->score = NAN; }
}

static float coalesced_decode_crf(scrappie_hip_engine *e, const_scrappie_matrix trans, int *path) {
    constexpr size_t MAX_READS = 4096;
    constexpr long long MAX_COLS = 4000000;
    ShInside inside(g_presence);
    CrfReq r;
    r.trans = trans; r.path = path;
    g_cc.run(r, g_presence, MAX_READS, false,
        [&](std::deque<CrfReq *> &q, std::vector<CrfReq *> &batch) {
            long long cols = 0;
            const size_t stride = q.front()->trans->stride;
            for (auto it = q.begin(); it != q.end() && batch.size() < MAX_READS;) {
                CrfReq *c = *it;
                if (c->trans->stride == stride && (batch.empty() || cols + (long long)c->trans->nc <= MAX_COLS)) { batch.push_back(c); cols += (long long)c->trans->nc; it = q.erase(it); }
                else ++it;
            }
        },
        [](std::vector<CrfReq *> &) { return true; }, [](CrfReq &) {},
        [&](std::vector<CrfReq *> &batch) { crf_batch(e, batch); });
    return r.score;
}
extern "C" void scrappie_hip_crf_coalescer_stats(unsigned long long out[3]) {
    std::lock_guard<std::mutex> lk(g_cc.mu);
    out[0] = g_cc.n_batches; out[1] = g_cc.n_reads; out[2] = g_cc.max_batch;
}

extern "C" float decode_crf(const_scrappie_matrix trans, int *path) {
    if (!trans || !path) return NAN;
    if (trans->nr != 25 || trans->nc == 0) return NAN;
    scrappie_hip_engine *e = default_engine();
    if (!e) return NAN;
    if (coalesce_on('d')) return coalesced_decode_crf(e, trans, path);
    (void)hipSetDevice(e->device);
    std::lock_guard<std::mutex> lk(e->mu);
    const int T = (int)trans->nc;
    DBuf dtr, dtb, dpath, dsc;
    float score = NAN;
    do {
        const size_t bytes = (size_t)T * trans->stride * 4;
        if (dtr.ensure(bytes) || dtb.ensure((size_t)T * 4) || dpath.ensure(((size_t)T + 1) * 4) || dsc.ensure(4)) break;
        if (hipMemcpyAsync(dtr.p, trans->data.f, bytes, hipMemcpyHostToDevice, e->stream) != hipSuccess) break;
        hipLaunchKernelGGL(k_crf_viterbi_only, dim3(1), dim3(64), 0, e->stream, dtr.as<float>(), (int)trans->stride, T, dtb.as<unsigned>(),
                           dpath.as<int>(), dsc.as<float>());
        float sc = NAN;
        if (hipMemcpyAsync(path, dpath.p, ((size_t)T + 1) * 4, hipMemcpyDeviceToHost, e->stream) != hipSuccess) break;
        if (hipMemcpyAsync(&sc, dsc.p, 4, hipMemcpyDeviceToHost, e->stream) != hipSuccess) break;
        if (hipStreamSynchronize(e->stream) != hipSuccess) break;
        score = sc;
    } while (0);
    for (DBuf *b : {&dtr, &dtb, &dpath, &dsc}) b->release();
    return score;
}
