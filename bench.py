#!/usr/bin/env python3
"""bench.py -- headline benchmark of the raw basecalling hot path.

Metric (BASELINE.json): raw samples/s (and kbases/s) for rgrgr_r94-shaped
synthetic 4000-sample reads.  Workload at N=1: BASELINE config[1],
"rgrgr_r94 raw, 10k synthetic 4000-sample reads, batch=64, 1xMI355X": the 10k
reads are handed over in submit batches of 64 and coalesced by the engine into
one launch group (a 64-read launch cannot fill 256 CUs: the recurrence only
parallelises over reads).  One STEP = one pass of the whole hot path over those
10k reads per GPU: conv -> 5x(affine, GRU) -> softmax -> Viterbi -> backtrace
on device, D2H of paths, homopolymer correction + k-mer stitching on the host.
Inputs are resident in HBM when the timed region starts.

    python bench.py --gpus N --steps K --warmup W

For N > 1 launch with torch.distributed.run (one rank per GPU); reads are
sharded across ranks, no data-path collective, weak scaling (10k reads/GPU).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA (2.5 PF; the 5 PF figure is 2:1 sparsity)


def make_reads(n_reads, n_samples, seed0, events=False):
    """Seeded synthetic squiggles, med/MAD normalised.
    64 distinct reads tiled to n_reads keeps set-up time bounded; the kernels'
    cost is data independent (fixed trip counts).  events=True: event tables of n_samples
    events turned into windowed features (12 floats per event) for an events model."""
    from scrappie_amd import synth
    distinct = min(n_reads, 256)
    if events:
        import scrappie_amd as sa
        base = [sa.event_features(synth.synthetic_events(n_samples, seed0 + i)).ravel() for i in range(distinct)]
    else:
        base = [synth.medmad_normalise(synth.synthetic_signal(n_samples, seed0 + i)) for i in range(distinct)]
    flat = np.concatenate([base[i % distinct] for i in range(n_reads)]).astype(np.float32)
    return flat, base


def measured_traffic(kernel, args):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/r1_traffic.json:
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, gfx950 FETCH
    correction applied).  Counters cannot be read from inside the timed run; the figure is
    reported only when the workload is the one it was measured on, else null."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r1_traffic.json")))
        w = t["workload"]
        if (w["model"], w["reads"], w["samples"]) != (args.model, args.reads, args.samples):
            return None
        return t["bytes_per_launch"][kernel]["total"]
    except Exception:
        return None


def hmm_bases_per_read(n_blocks, n=16):
    """transducer models on synthetic weights decode to a handful of bases per read, so
    kbases/s is also quoted for decode driven by HMM-simulated posteriors (55 % stay / 40 % step /
    5 % skip).  Returns the mean bases per read of `n` such posteriors of n_blocks blocks decoded
    through the C ABI (decode_transducer + overlapper on the GPU library).  The decode kernels have
    fixed trip counts, so the device time per read does not depend on which posterior it is."""
    import scrappie_amd as sa
    from scrappie_amd import synth
    tot = 0
    for i in range(n):
        post, _ = synth.simulated_posterior(n_blocks, 900 + i)
        bases, _, _ = sa.decode_post(sa.ScrappyMatrix.from_numpy(post, sloika=False), "rgrgr_r94")
        tot += len(bases or "")
    return tot / float(n)


def cpu_baseline(weights, base_reads, budget_s=12.0):
    """The reference's recipe -- threads over reads, single-threaded OpenBLAS
    (README.md:68-71) -- applied to the oracle (kind 'port': the reference
    sources do not travel and cannot be built without stand-ins).
    The oracle is compiled here with -O3 -march=native for THIS host; its two BLAS
    call shapes go to scipy's bundled OpenBLAS when that is found, else to its own
    loops.  Bounded sample of the same reads: one pass with 1 thread, one with all
    host threads (capped at 64)."""
    import ctypes as C
    import glob
    from concurrent.futures import ThreadPoolExecutor
    import oracle
    so = os.path.join(tempfile.gettempdir(), "liboracle_fast_%d.so" % os.getpid())
    src = os.path.join(ROOT, "oracle", "oracle.c")
    subprocess.run(["gcc", "-O3", "-march=native", "-std=c99", "-fPIC", "-shared", "-ffp-contract=off",
                    "-o", so, src, "-lm"], check=True)
    L = C.CDLL(so)
    oracle._declare(L)
    blas = "own loops"
    try:
        import scipy
        cands = glob.glob(os.path.join(os.path.dirname(scipy.__file__) + ".libs", "libscipy_openblas*.so"))
        if cands:
            B = C.CDLL(cands[0])
            B.scipy_openblas_set_num_threads(1)
            L.orc_set_blas.argtypes = [C.c_void_p, C.c_void_p]
            L.orc_set_blas(C.cast(B.scipy_cblas_sgemv, C.c_void_p), C.cast(B.scipy_cblas_sgemm, C.c_void_p))
            blas = "scipy OpenBLAS (1 thread per call)"
    except Exception:
        pass
    om = oracle.OracleModel(weights)
    p = L.orc_default_params()
    p.do_trim = 0

    def one(x):
        r = oracle.basecall_raw(om, x, p, L=L)
        return len(x), (len(r["bases"]) if r else 0)

    def run(nthreads, budget):
        done, t0 = [], time.perf_counter()
        with ThreadPoolExecutor(nthreads) as ex:
            reads = list(base_reads)
            while time.perf_counter() - t0 < budget:
                done.extend(ex.map(one, reads[:max(nthreads, 8)]))
        dt = time.perf_counter() - t0
        return sum(d[0] for d in done) / dt, sum(d[1] for d in done) / dt / 1e3, len(done), dt

    v1, kb1, n1, dt1 = run(1, budget_s / 2)
    nthr = min(os.cpu_count() or 1, 64)
    vN, kbN, nN, dtN = run(nthr, budget_s / 2)
    os.unlink(so)
    return {"value": vN, "unit": "samples/s", "cores": nthr, "kind": "port",
            "sample": "%d reads x %d samples in %.1f s on %d threads (threads over reads); rgrgr_r94-shaped synthetic "
                      "weights; scalar C oracle, -O3 -march=native, BLAS = %s" % (nN, len(base_reads[0]), dtN, nthr, blas),
            "value_1thread": v1, "kbases_per_s": kbN, "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=10000, help="reads per GPU per step")
    ap.add_argument("--samples", type=int, default=4000)
    ap.add_argument("--model", default="rgrgr_r94")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    distributed = world > 1
    # test hooks (never set by the driver): BENCH_BACKEND=gloo and BENCH_DEVICE=0 let two ranks share
    # one GPU so that the multi-rank path can be exercised on a single-GPU box
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if "BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["BENCH_DEVICE"])
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    import scrappie_amd as sa
    from scrappie_amd import model
    from scrappie_amd.parallel import shard_range

    weights = model.synthetic_model(args.model, seed=1)
    events = weights["arch"] == "events"       # --model nanonet_events: --samples counts events per read
    eng = sa.Engine(local_rank)
    eng.load_model(args.model, weights)
    eng.set_max_launch_reads(max(16384, args.reads))

    # this rank's shard of the global read set (weak scaling: args.reads per GPU)
    total_reads = args.reads * world
    lo, hi = shard_range(total_reads, world, rank)
    n = hi - lo
    flat, base = make_reads(n, args.samples, seed0=1 + 1000 * rank, events=events)
    d_sig = eng.upload(flat)
    off = np.arange(n, dtype=np.uint64) * np.uint64(args.samples * (12 if events else 1))
    ln = np.full(n, args.samples, np.uint32)
    params = eng.default_params()

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        eng.synchronize()

    def enqueue():
        eng.run_device(d_sig, off, ln, args.model, params)

    def finish():
        """collect the OLDEST launch group in flight + the stage timings its events recorded"""
        nb = eng.collect(n, params, raw=True)
        return nb, eng.timing()

    for _ in range(args.warmup):
        enqueue()
        finish()
    eng.set_profiling(True)
    gru_ms, gru_launches, gru_flops, stage = 0.0, 0, 0.0, {}
    fused = [0.0, 0, 0.0]       # ms, launches, FLOPs of the one-kernel layers (k_gru_proj: projection + recurrence)
    nbases = 0
    barrier()
    t0 = time.perf_counter()
    # K steps, software pipelined: the engine holds two launch groups, so step k+1's
    # kernels are enqueued before the host waits for / stitches step k.  Everything
    # of all K steps (enqueue, kernels, D2H, stitching) happens between the barriers.
    if args.steps > 0:
        enqueue()
    for k in range(args.steps):
        if k + 1 < args.steps:
            enqueue()
        nb, tm = finish()
        nbases += nb
        gru_ms += tm["gru_ms"]
        gru_launches += tm["n_gru_launches"]
        gru_flops += tm["gru_flops"]
        fused[0] += tm["fused_ms"]; fused[1] += tm["n_fused_launches"]; fused[2] += tm["fused_flops"]
        for key in ("conv_ms", "affine_ms", "gru_ms", "ff_ms", "decode_ms", "backtrace_ms", "total_ms"):
            stage[key] = stage.get(key, 0.0) + tm[key]
    barrier()
    dt = time.perf_counter() - t0
    eng.set_profiling(False)

    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        nb = torch.tensor([nbases], dtype=torch.float64, device=red_dev)
        dist.all_reduce(nb, op=dist.ReduceOp.SUM)
        nbases = float(nb.item())

    if rank == 0:
        samples_total = float(total_reads) * args.samples * args.steps
        value = samples_total / dt
        d = model.model_dims(weights)
        # dominant kernel: a recurrent layer.  For the GRU stacks a layer is one kernel (k_gru_proj: projection
        # team + recurrence team, FLOPs = projection + recurrence) whose contractions run as split products
        # on the bf16 matrix pipe: six bf16 MFMA flops per algorithmic fp32 flop, so the fp32-equivalent
        # roof of that pipe is its dense bf16 peak / 6.  The events LSTM still runs exact-fp32 MFMAs.
        is_fused = fused[1] > 0
        if is_fused:
            gru_ms, gru_launches, gru_flops = fused
        gru_avg_ms = gru_ms / max(gru_launches, 1)
        achieved = (gru_flops / max(gru_launches, 1)) / (gru_avg_ms * 1e-3) / 1e12 if gru_ms > 0 else 0.0
        split = not events
        peak = BF16_MFMA_PEAK_TFLOPS / 6.0 if split else FP32_MFMA_PEAK_TFLOPS
        out = {
            "metric": ("events/sec, %s bi-LSTM (SURVEY 8(f).4; not the headline metric)" % args.model) if events
                      else "raw samples/sec, rgrgr_r94 4k-sample reads",
            "value": value,
            "unit": "events/s" if events else "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "dtype_note": ("all tensors, accumulators and results are fp32; the projection / recurrence / S1 contractions execute "
                           "as six bf16 partial products of exact 3-way bf16 splits of their fp32 operands (error at or below an "
                           "fp32 FMA chain's: profiles/r1_split_probe.txt); every parity test runs at the fp32 tolerances") if not events
                          else "fp32; the events LSTM runs exact-fp32 MFMAs, its projections and S1 as split products",
            "data": "synthetic",
            "config": {"workload": "%s raw, %d synthetic %d-sample reads per GPU, submit-batch=64 coalesced into one "
                                   "launch group, %dxMI355X" % (args.model, args.reads, args.samples, world),
                       "reads_per_gpu_per_step": args.reads, "samples_per_read": args.samples,
                       "blocks_per_read": (args.samples + d["stride"] - 1) // d["stride"],
                       "dims": d, "weights": "synthetic (reference model headers are missing blobs)"},
            "kbases_per_s": nbases / dt / 1e3,
            "kbases_note": "as called on synthetic weights (degenerate for transducer models: SURVEY.md section 7)",
            "kbases_per_s_hmm_posteriors": None,
            "roofline": {"kernel": ("k_lstm_lanes<%d>" if events else ("k_gru_proj<%d> (projection + recurrence of one layer)" if is_fused else "k_gru_split<%d>")) % (d["S"] // 16),
                         "bound": "mfma", "achieved": achieved,
                         "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak,
                         "peak_note": ("dense bf16 MFMA peak %.0f TFLOP/s / 6: every fp32 product is six bf16 partial products of exact "
                                       "3-way splits, accumulated in fp32 (tools/split_probe.hip: closer to float64 than the fp32 MFMA)"
                                       % BF16_MFMA_PEAK_TFLOPS) if split else "dense fp32 MFMA peak",
                         "achieved_over_f32_mfma_peak": achieved / FP32_MFMA_PEAK_TFLOPS,
                         "traffic": None if events else measured_traffic("k_gru_proj" if is_fused else "k_gru_split", args),
                         "traffic_unit": "HBM bytes per launch (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, profiles/r1_traffic.json)",
                         "algorithmic_bytes": float(total_reads // world) * ((args.samples + d["stride"] - 1) // d["stride"])
                                              * (5.0 if events else (2.0 if is_fused else 4.0)) * d["S"] * 4,
                         "avg_launch_ms": gru_avg_ms,
                         "flops_per_launch": gru_flops / max(gru_launches, 1),
                         "note": "algorithmic FLOPs per read per block = 2*3*S*S (recurrence) + 2*S*3S (the layer's input projection, "
                                 "same kernel), 2*4*S*S (LSTM); bytes = S in + S out (gate inputs stay in LDS), 4S in + S out (LSTM); "
                                 "(SURVEY 8d); HIP events on the engine's stream; rank 0"},
            "stage_ms_per_step": {k: v / args.steps for k, v in stage.items()},
        }
        if world == 1 and not events and d["NS"] == 1025:
            nblk = (args.samples + d["stride"] - 1) // d["stride"]
            bpr = hmm_bases_per_read(nblk)
            out["kbases_per_s_hmm_posteriors"] = value / args.samples * bpr / 1e3
            out["kbases_hmm_note"] = ("reads/s of this run x %.1f bases per read decoded (same kernels, C ABI) from HMM-simulated "
                                      "posteriors of %d blocks (SURVEY 8d); decode cost is data independent" % (bpr, nblk))
        if not args.no_cpu_baseline and world == 1 and not events:
            out["cpu_baseline"] = cpu_baseline(weights, base)
        print(json.dumps(out))
    eng.free(d_sig)
    eng.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
