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


def make_reads(n_reads, n_samples, seed0):
    """Seeded synthetic squiggles, med/MAD normalised.
    64 distinct reads tiled to n_reads keeps set-up time bounded; the kernels'
    cost is data independent (fixed trip counts)."""
    from scrappie_amd import synth
    distinct = min(n_reads, 256)
    base = [synth.medmad_normalise(synth.synthetic_signal(n_samples, seed0 + i)) for i in range(distinct)]
    flat = np.concatenate([base[i % distinct] for i in range(n_reads)]).astype(np.float32)
    return flat, base


def cpu_baseline(weights, base_reads, budget_s=15.0):
    """The oracle (kind 'port': scalar C restatement of the reference path, no
    BLAS) compiled here with -O3 -march=native for THIS host, one thread, timed
    on a bounded sample of the same workload."""
    import ctypes as C
    import oracle
    so = os.path.join(tempfile.gettempdir(), "liboracle_fast_%d.so" % os.getpid())
    src = os.path.join(ROOT, "oracle", "oracle.c")
    subprocess.run(["gcc", "-O3", "-march=native", "-std=c99", "-fPIC", "-shared", "-ffp-contract=off",
                    "-o", so, src, "-lm"], check=True)
    L = C.CDLL(so)
    oracle._declare(L)
    om = oracle.OracleModel(weights)
    p = L.orc_default_params()
    p.do_trim = 0
    nsamp, nbase, nread = 0, 0, 0
    t0 = time.perf_counter()
    for x in base_reads:
        r = oracle.basecall_raw(om, x, p, L=L)
        nsamp += len(x)
        nbase += len(r["bases"]) if r else 0
        nread += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    os.unlink(so)
    return {"value": nsamp / dt, "unit": "samples/s", "cores": 1, "kind": "port",
            "sample": "%d reads x %d samples, rgrgr_r94-shaped synthetic weights, 1 thread, %.1f s; "
                      "scalar C oracle (no BLAS), -O3 -march=native" % (nread, len(base_reads[0]), dt),
            "kbases_per_s": nbase / dt / 1e3, "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
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
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)

    import scrappie_amd as sa
    from scrappie_amd import model
    from scrappie_amd.parallel import shard_range

    weights = model.synthetic_model(args.model, seed=1)
    eng = sa.Engine(local_rank)
    eng.load_model(args.model, weights)
    eng.set_max_launch_reads(max(16384, args.reads))

    # this rank's shard of the global read set (weak scaling: args.reads per GPU)
    total_reads = args.reads * world
    lo, hi = shard_range(total_reads, world, rank)
    n = hi - lo
    flat, base = make_reads(n, args.samples, seed0=1 + 1000 * rank)
    d_sig = eng.upload(flat)
    off = np.arange(n, dtype=np.uint64) * np.uint64(args.samples)
    ln = np.full(n, args.samples, np.uint32)
    params = eng.default_params()

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        eng.synchronize()

    def step():
        eng.run_device(d_sig, off, ln, args.model, params)
        return eng.collect(n, params, raw=True)

    for _ in range(args.warmup):
        step()
    eng.set_profiling(True)
    gru_ms, gru_launches, gru_flops, stage = 0.0, 0, 0.0, {}
    nbases = 0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        nbases += step()
        tm = eng.timing()          # stream already drained by collect(); reads event deltas only
        gru_ms += tm["gru_ms"]
        gru_launches += tm["n_gru_launches"]
        gru_flops += tm["gru_flops"]
        for k in ("conv_ms", "affine_ms", "gru_ms", "ff_ms", "decode_ms", "backtrace_ms", "total_ms"):
            stage[k] = stage.get(k, 0.0) + tm[k]
    barrier()
    dt = time.perf_counter() - t0
    eng.set_profiling(False)

    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        nb = torch.tensor([nbases], dtype=torch.float64, device="cuda")
        dist.all_reduce(nb, op=dist.ReduceOp.SUM)
        nbases = float(nb.item())

    if rank == 0:
        samples_total = float(total_reads) * args.samples * args.steps
        value = samples_total / dt
        d = model.model_dims(weights)
        gru_avg_ms = gru_ms / max(gru_launches, 1)
        achieved = (gru_flops / max(gru_launches, 1)) / (gru_avg_ms * 1e-3) / 1e12 if gru_ms > 0 else 0.0
        out = {
            "metric": "raw samples/sec, rgrgr_r94 4k-sample reads",
            "value": value,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s raw, %d synthetic %d-sample reads per GPU, submit-batch=64 coalesced into one "
                                   "launch group, %dxMI355X" % (args.model, args.reads, args.samples, world),
                       "reads_per_gpu_per_step": args.reads, "samples_per_read": args.samples,
                       "blocks_per_read": (args.samples + d["stride"] - 1) // d["stride"],
                       "dims": d, "weights": "synthetic (reference model headers are missing blobs)"},
            "kbases_per_s": nbases / dt / 1e3,
            "kbases_note": "as called on synthetic weights (degenerate for transducer models: SURVEY.md section 7)",
            "roofline": {"kernel": "k_gru<%d>" % (d["S"] // 16), "bound": "mfma", "achieved": achieved,
                         "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "traffic": None,
                         "avg_launch_ms": gru_avg_ms,
                         "flops_per_launch": gru_flops / max(gru_launches, 1),
                         "note": "algorithmic FLOPs = 2*3*S*S per read per block (SURVEY 8d); rank 0"},
            "stage_ms_per_step": {k: v / args.steps for k, v in stage.items()},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(weights, base)
        print(json.dumps(out))
    eng.free(d_sig)
    eng.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
