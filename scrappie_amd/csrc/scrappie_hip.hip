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
 *                    -> k_walk_stitch_out (walk back + homopolymer pass + stitching + results to pinned host memory, copy stream;
 *                       SH_SPLIT_TAIL, host stitching, no copy stream: k_backtrace -> k_stitch -> k_results_out)   (rnnrf: k_affine -> k_crf -> k_stitch -> k_results_out)
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
 *
 * The translation unit in parts (round 5; the kernels are templates in headers and are instantiated where they are launched, so
 * the engine stays ONE translation unit -- the parts are files of their own for reading, included below in this order):
 *   this file            switches, the engine's state, create / destroy
 *   sh_eng_weights.inc   weights as the device wants them; the Model
 *   sh_eng_load.inc      .scrm container -> Model, settings, the ABI's planning functions
 *   sh_eng_launch.inc    launch-group construction, kernel dispatch helpers
 *   sh_eng_pipeline.inc  run_pipeline: one launch group through its kernels
 *   sh_eng_groups.inc    run_device / collect / stitching, a call cut into launch groups
 *   sh_eng_batch.inc     helper engine for chain-bound reads, host-signal entry points, several GPUs
 *   sh_eng_debug.inc     measurement / test hooks
 *   sh_eng_surface.inc   the reference's per-read functions
 * Separate translation units: sh_p0.hip (signal preparation, k_p0), sh_host.c / sh_fast5.c / sh_h5mini.c (host C);
 * sh_coalesce.h (the per-read functions' queue) and sh_dev.h are plain C++ headers.
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
    bool affine_reg, gru_single, gru_stamp, gru_separate, gru_f32, gru_lanes_stamp, proj_stamp, ff_reg, ff_stamp, vit_stamp, ff_separate, fv_single, host_stamp, host_stitch, split_tail, helper_fence, gru_free, gru_barrier, input_order, conv_valu, conv_in_layer, gru32, gru16, gru32_stamp;
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
        split_tail = on("SH_SPLIT_TAIL");         /* walk back, stitching and result transfer as three kernels (k_backtrace, k_stitch, k_results_out) where k_walk_stitch_out applies */
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

#include "sh_eng_weights.inc"      /* weights as the device wants them: MFMA fragments, fp16 pieces, bias tables; the Model */
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

/* scrappie_hip_basecall_batch from several host threads (sh_eng_batch.inc): the requests that are waiting for one engine run as one engine call */
struct BatchReq {
    scrappie_hip_engine *e = nullptr;
    int model = 0;
    const raw_table *reads = nullptr;
    size_t n = 0;
    scrappie_hip_params p{};
    scrappie_hip_call *out = nullptr;
    int rc = -1;
    std::string err;
    int phase = 0;                   /* sh_coalesce.h: 0 queued ... 3 done */
};
struct BatchCoalescer : ShCoalescer<BatchReq> { BatchCoalescer() { target_pct = 100; window_mul = 4; } };      /* a leader waits for every thread seen inside lately (sh_coalesce.h) */

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
                                        chain-bound reads) lowers it to 0.45 for good -- the helper takes 0.3 (two helpers: 0.15 each; tail_mem_frac) -- so later calls
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
    /* scrappie_hip_basecall_device_stream: the last launch group of the previous call, still in flight */
    struct Carry { bool live = false; int slot = 0; std::vector<uint32_t> perm; scrappie_hip_call *out = nullptr; scrappie_hip_params p{}; } carry;
    std::mutex mu;
    double tail_frac = 0.3;          /* the helper engine's share of the device's memory as settled when it was made (free memory then, at most tail_mem_frac()) */
    double dbg_tail_free_frac = 0;   /* test hook (debug option "tail_free_frac", in 1/1000): the free share the helper's creation sees */
    BatchCoalescer batch_co;         /* this engine's queue of small scrappie_hip_basecall_batch calls, and the threads seen inside them lately */
    ShPresence batch_presence;
    std::mutex call_mu;              /* scrappie_hip_basecall_batch: one call at a time inside the engine (concurrent small calls share one: sh_eng_batch.inc) */
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
/* the NUMA node a device's PCIe root belongs to (-1: unknown / one node): where its engine's and preparer's pinned buffers are placed (sh_numa.h) */
extern "C" int scrappie_hip_device_numa_node(int device) { return sh_device_numa_node(device); }

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
    (void)sh_stream_wait(e->stream);
    if (e->cstream) (void)sh_stream_wait(e->cstream);
    if (e->ustream) (void)sh_stream_wait(e->ustream);
    if (e->pstream) (void)sh_stream_wait(e->pstream);
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

#include "sh_eng_load.inc"      /* the .scrm container -> Model, engine settings, the planning functions of the ABI */
#include "sh_eng_launch.inc"      /* launch-group construction (tiles, metadata, schedules) and the kernel dispatch helpers */
#include "sh_eng_pipeline.inc"      /* run_pipeline: one launch group through its kernels */
#include "sh_eng_groups.inc"      /* run_device / collect / stitching, a call cut into launch groups, device-resident entry points */
#include "sh_eng_batch.inc"      /* chain-bound reads on a helper engine, host-signal entry points, several GPUs */
#include "sh_eng_debug.inc"      /* measurement / test hooks: decoder and trunk inputs, debug_option / debug_fetch / debug_stitch */
#include "sh_eng_surface.inc"      /* the reference's per-read functions: posterior / trunk on an explicit engine, the process-default engine, decode_transducer, decode_crf; all three coalesced (sh_coalesce.h) */