#!/usr/bin/env python3
"""bench.py -- headline benchmark of the raw basecalling hot path.

Metric (BASELINE.json): raw samples/s (and kbases/s) for rgrgr_r94-shaped
synthetic 4000-sample reads.  Workload at N=1: BASELINE config[1],
"rgrgr_r94 raw, 10k synthetic 4000-sample reads, 1xMI355X": the 10k reads are
handed to the engine in ONE call and run as one launch group (BASELINE's
"batch=64" is the reference's submit granularity; a 64-read launch cannot fill
256 CUs because the recurrence only parallelises over reads, so the engine's
unit of work is the launch group).  One STEP = one pass of the whole hot path
over those 10k reads per GPU: conv -> 5 x (projection + GRU) -> softmax ->
Viterbi -> backtrace on device, D2H of paths, homopolymer correction + k-mer
stitching on the host.

Three timed regions, each K steps between barriers (only the first one is the
contract's `value`; the other two are extra fields of the same JSON line):
  1. `value`: inputs resident in HBM when the timed region starts (bench contract).
  2. `host_to_host`: SURVEY 8(d)'s definition -- normalised signal in pageable host
     memory to base strings in host memory through scrappie_hip_basecall_batch
     (H2D + kernels + D2H + stitching inside the barriers).
  3. `kbases_per_s_hmm_posteriors`: the same device-resident step on the DEFAULT
     kernels with realistic calls: the output layer is built from state codes and
     reads trunk activations that encode a simulated k-mer path
     (scrappie_hip_set_trunk_input; the network above it still runs in full), because
     random weights decode to ~5 bases per read: S1 + Viterbi (k_ff_viterbi) -> D2H ->
     homopolymer -> overlapper then run on ~400-base calls; bases are COUNTED.

    python bench.py --gpus N --steps K --warmup W

For N > 1 launch with torch.distributed.run (one rank per GPU); reads are
sharded across ranks, no data-path collective, weak scaling (10k reads/GPU).
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
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
F16_MFMA_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense bf16 / f16 MFMA (2.5 PF; the 5 PF figure is 2:1 sparsity)
SPLIT_PRODUCTS = 3                  # f16 MFMA flops issued per algorithmic fp32 flop (sh_kernels.h: a1 b2 + a2 b1 + a1 b1)


def make_reads(first_index, n_reads, n_samples, seed, events=False):
    """Seeded synthetic squiggles, med/MAD normalised: read number i of
    the GLOBAL read set is generated from (seed, i) alone, so a shard holds the same reads whatever the number
    of GPUs, and no input file is needed at any scale.  Piecewise-constant levels ~ N(0, 1), dwell ~
    Geometric(mean 9 samples), N(0, 0.1^2) noise.  events=True: event tables of n_samples events turned into
    windowed features (12 floats per event) for an events model (256 distinct tables, tiled)."""
    from scrappie_amd import synth
    if events:
        import scrappie_amd as sa
        distinct = min(n_reads, 256)
        base = [sa.event_features(synth.synthetic_events(n_samples, seed + first_index + i)).ravel() for i in range(distinct)]
        flat = np.concatenate([base[i % distinct] for i in range(n_reads)]).astype(np.float32)
        return flat, base
    flat = np.empty((n_reads, n_samples), dtype=np.float32)
    nlev = int(n_samples / 9.0 * 1.5) + 16
    for i in range(n_reads):
        rng = np.random.default_rng([seed, first_index + i])
        dwell = rng.geometric(1.0 / 9.0, size=nlev)
        while dwell.sum() < n_samples:
            dwell = np.concatenate([dwell, rng.geometric(1.0 / 9.0, size=nlev)])
        sig = np.repeat(rng.standard_normal(len(dwell)), dwell)[:n_samples] + 0.1 * rng.standard_normal(n_samples)
        med = np.median(sig)
        mad = np.median(np.abs(sig - med)) * 1.4826
        flat[i] = (sig - med) / mad
    return flat.ravel(), [flat[i] for i in range(min(n_reads, 64))]


def csrc_tree_hash():
    """sha256 over the kernel and engine sources (scrappie_amd/csrc: *.h *.hip *.inc + Makefile -- the device code, what launches it and the flags it is
    built with; not the host C files -- names and contents, sorted): what a committed PMC measurement is valid for.  Computed from the files (the GPU box
    has no .git)."""
    import hashlib
    d = os.path.join(ROOT, "scrappie_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f == "Makefile" or f.endswith((".h", ".hip", ".inc")):
            h.update(f.encode() + b"\0")
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def latest_traffic_file():
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")):
        m = re.match(r"r(\d+)_traffic\.json$", os.path.basename(f))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    return best[1] if best else None


_TRAFFIC_NOTE = [None]


def measured_traffic(kernel, args):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/r<N>_traffic.json: separate rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE runs of this same command, gfx950 FETCH correction applied).  Counters cannot be read from inside the timed run, so the
    figure is a measurement of a BUILD: it is reported only when the file records the hash of the kernel sources it was measured on
    (csrc_tree_hash) and that hash is the running tree's, and the workload is the one measured; else null, with the reason in
    roofline.traffic_note (a constant that outlives the kernels it measured is not a measurement)."""
    f = latest_traffic_file()
    if not f:
        _TRAFFIC_NOTE[0] = "no profiles/r*_traffic.json"
        return None
    try:
        t = json.load(open(f))
        w = t["workload"]
        if (w["model"], w["reads"], w["samples"]) != (args.model, args.reads, args.samples):
            _TRAFFIC_NOTE[0] = "%s was measured on another workload" % os.path.basename(f)
            return None
        have, want = t.get("csrc_sha256"), csrc_tree_hash()
        if have != want:
            _TRAFFIC_NOTE[0] = ("%s was measured on kernel sources %s, this tree is %s: re-run tools/profile_round.sh" %
                                (os.path.basename(f), (have or "unrecorded")[:12], want[:12]))
            return None
        _TRAFFIC_NOTE[0] = "%s, measured on this tree's kernel sources (sha256 %s)" % (os.path.basename(f), want[:12])
        return t["bytes_per_launch"][kernel]["total"]
    except Exception as ex:
        _TRAFFIC_NOTE[0] = "unreadable: %s" % ex
        return None


def cli_end_to_end(weights, model_name, n_reads, n_samples, fmt="f32"):
    """`scrappie raw` itself, files in -> FASTA out, on one GPU (the deliverable north_star names): n_reads synthetic raw reads written as files
    (tools/make_reads.c: levels + noise, a quiet stretch in front), then scrappie_amd/scrappie raw --stats over the directory: loader threads
    read each file straight into pinned memory, k_p0 trims and normalises on the GPU, the engine basecalls, records are written.
    fmt = "fast5": the ONLY input format the reference reads (fast5_interface.c:130-217) -- single-read fast5 as MinKNOW writes them (int16
    Signal, chunked, deflate level 1), written through libhdf5 where one is found (the generator needs it; the reader does not: sh_h5mini.c +
    sh_inflate.c).  fmt = "f32": headerless float32 samples.  Returns the CLI's own rates (its --stats lines)."""
    import re
    import shutil
    from scrappie_amd import model as _model
    cli = os.path.join(ROOT, "scrappie_amd", "scrappie")
    src = os.path.join(ROOT, "tools", "make_reads.c")
    if not (os.path.exists(cli) and os.path.exists(src) and shutil.which("gcc")):
        return {"error": "needs scrappie_amd/scrappie, tools/make_reads.c and gcc"}
    tmp = tempfile.mkdtemp(prefix="sh_cli_")
    try:
        gen = os.path.join(tmp, "make_reads")
        if fmt == "fast5":
            built = False
            for inc, lib in (("/opt/conda/include", "/opt/conda/lib"), ("/usr/include/hdf5/serial", "/usr/lib/x86_64-linux-gnu/hdf5/serial"), ("/usr/include", "/usr/lib/x86_64-linux-gnu")):
                if os.path.exists(os.path.join(inc, "hdf5.h")):
                    r = subprocess.run(["gcc", "-O2", "-DWITH_HDF5", "-I" + inc, "-o", gen, src, "-L" + lib, "-lhdf5", "-Wl,-rpath," + lib, "-lm"], capture_output=True)
                    if r.returncode == 0:
                        built = True
                        break
            if not built:
                return {"error": "no libhdf5 development files on this box to WRITE fast5 test files with (the reader needs none)"}
        else:
            subprocess.run(["gcc", "-O2", "-o", gen, src, "-lm"], check=True, capture_output=True)
        rdir = os.path.join(tmp, "reads")
        os.mkdir(rdir)
        t0 = time.time()
        nproc = 8 if fmt == "fast5" else 1                   # (libhdf5 writes ~4500 files per second and process)
        procs = [subprocess.Popen([gen, fmt, rdir, str(min(n_reads, (k + 1) * ((n_reads + nproc - 1) // nproc))), str(n_samples), str(k * ((n_reads + nproc - 1) // nproc))])
                 for k in range(nproc)]
        if any(p.wait() != 0 for p in procs):
            return {"error": "the read generator failed"}
        t_gen = time.time() - t0
        mpath = os.path.join(tmp, model_name + ".scrm")
        _model.save_model(weights, mpath)
        cmd = [cli, "raw", "--model", model_name, "--model-file", mpath, "--stats", "-o", os.path.join(tmp, "out.fa"), rdir]      # all defaults
        # twice, the second run reported: the files were written a moment ago with the GPU idle (its clocks ramp over the first launch
        # groups after an idle spell, as in the bench's own warm-up), and the first run is what a user's first run is
        first_wall = None
        for rep in range(2):
            t0 = time.time()
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, SCRAPPIE_FAST5_READER="own"))
            t_proc = time.time() - t0
            if r.returncode != 0:
                return {"error": "scrappie raw failed: " + r.stderr[-400:]}
            if rep == 0:
                m0 = re.search(r"wall [0-9.]+ s = ([0-9.e+]+) samples/s", r.stderr)
                first_wall = float(m0.group(1)) if m0 else None
        st = " ".join(l for l in r.stderr.splitlines() if l.startswith("scrappie stats:"))
        f = lambda pat: float(re.search(pat, st).group(1))
        import hashlib
        fa = open(os.path.join(tmp, "out.fa"), "rb").read()
        nrec = fa.count(b"\n>") + (1 if fa.startswith(b">") else 0)
        # order-insensitive digest of the sequences (records appear in completion order, as the reference's: scrappie_raw.c:377,402)
        seqs = sorted(l for l in fa.split(b"\n") if l and not l.startswith(b">"))
        return {"value": f(r"wall [0-9.]+ s = ([0-9.e+]+) samples/s"), "unit": "samples/s", "input": fmt, "wall_s": f(r"wall ([0-9.]+) s"),
                "kbases_per_s": f(r"samples/s, ([0-9.]+) kbases/s"), "reads": n_reads, "samples_per_read": n_samples, "records": nrec,
                "sequences_md5": hashlib.md5(b"\n".join(seqs)).hexdigest(),
                "loader_threads": int(re.search(r"prep=\w+, (\d+) host threads", st).group(1)), "read_s": f(r"read ([0-9.]+) s"), "prepare_s": f(r"prepare ([0-9.]+) s"), "engine_s": f(r"engine ([0-9.]+) s"),
                "engine_samples_per_s": f(r"engine [0-9.]+ s \(([0-9.e+]+) samples/s\)"), "loader_samples_per_s": n_reads * n_samples / max(f(r"read ([0-9.]+) s") + f(r"prepare ([0-9.]+) s"), 1e-9),
                "process_s": t_proc, "generate_s": t_gen, "first_run_value": first_wall,
                "note": "scrappie raw --stats on %d %s files of %d samples, all options at their defaults (page cache warm: written a moment before): wall = first "
                        "file opened to last record written, engines and arenas already up (process_s includes start-up, model load and the arena warm-up); three "
                        "stages on three host threads: read (loader team, files straight into pinned memory%s), prepare (k_p0) + streaming engine calls, records; "
                        "engine_s = first engine call to last batch delivered; batches of 16384 reads after a ramp; the second of two runs "
                        "(first_run_value: the first, with the GPU coming out of the idle spell in which the files were written)"
                        % (n_reads, ".fast5 (int16, chunked, deflate 1: as MinKNOW writes them)" if fmt == "fast5" else ".f32", n_samples,
                           "; fast5 through the built-in HDF5-subset reader and inflater, no libhdf5, no zlib" if fmt == "fast5" else "")}
    except Exception as ex:
        return {"error": str(ex)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_baseline(weights, base_reads, budget_s=10.0):
    """The reference's recipe -- `#pragma omp parallel for schedule(dynamic)` over reads
    (scrappie_raw.c:355,387), single-threaded OpenBLAS (README.md:68-71) -- applied to the
    oracle (kind 'port': the reference sources do not travel and cannot be built without
    stand-ins).  The oracle is compiled here with -O3 -march=native -fopenmp for
    THIS host and its OpenMP read loop (orc_basecall_many) run once on 1 thread and once on all
    host CPUs, each for a bounded wall-time budget on the same reads the GPU ran; its two BLAS call
    shapes go to scipy's bundled OpenBLAS when that is found, else to its own loops."""
    import ctypes as C
    import glob
    import oracle
    src = os.path.join(ROOT, "oracle", "oracle.c")

    def build(tag, extra):
        so = os.path.join(tempfile.gettempdir(), "liboracle_%s_%d.so" % (tag, os.getpid()))
        subprocess.run(["gcc", "-O3", "-march=native", "-std=c99", "-fPIC", "-shared", "-ffp-contract=off", "-fopenmp"]
                       + extra + ["-o", so, src, "-lm"], check=True)
        L = C.CDLL(so)
        oracle._declare(L)
        L.orc_basecall_many.restype = C.c_long
        L.orc_basecall_many.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_size_t), C.c_size_t,
                                        C.c_void_p, C.c_int, C.c_double, C.c_size_t,
                                        C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        os.unlink(so)
        return L

    reads = [np.ascontiguousarray(r, dtype=np.float32) for r in base_reads]
    n = len(reads)
    ptrs = (C.POINTER(C.c_float) * n)(*[r.ctypes.data_as(C.POINTER(C.c_float)) for r in reads])
    lens = (C.c_size_t * n)(*[len(r) for r in reads])

    def run(L, nthreads, budget):
        om = oracle.OracleModel(weights)
        p = L.orc_default_params()
        p.do_trim = 0
        ns, nb, el = C.c_double(), C.c_double(), C.c_double()
        done = L.orc_basecall_many(om.ptr, ptrs, lens, n, C.byref(p), nthreads, budget, nthreads,
                                   C.byref(ns), C.byref(nb), C.byref(el))
        return ns.value / el.value, nb.value / el.value / 1e3, done, el.value

    nproc = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        cgroup = open("/sys/fs/cgroup/cpu.max").read().strip()       # "max 100000" = no CPU quota
    except OSError:
        cgroup = "n/a"
    runs = []
    # (a) the oracle's dot products as vectorisable loops, one OpenMP thread per host CPU
    La = build("loops", ["-DORC_FAST_LOOPS"])
    runs.append(("own vectorised loops", nproc) + run(La, nproc, budget_s * 0.4))
    # (b) BLAS call shapes routed to scipy's bundled OpenBLAS, single-threaded per call as README.md:70-71
    #     asks; that build serves at most 64 concurrent callers (its NUM_THREADS), so 64 read threads at most
    v1 = None
    try:
        import scipy
        cands = glob.glob(os.path.join(os.path.dirname(scipy.__file__) + ".libs", "libscipy_openblas*.so"))
        if cands:
            Lb = build("blas", [])
            B = C.CDLL(cands[0])
            B.scipy_openblas_set_num_threads(1)
            Lb.orc_set_blas.argtypes = [C.c_void_p, C.c_void_p]
            Lb.orc_set_blas(C.cast(B.scipy_cblas_sgemv, C.c_void_p), C.cast(B.scipy_cblas_sgemm, C.c_void_p))
            v1 = ("scipy OpenBLAS",) + run(Lb, 1, budget_s * 0.2)
            nb_thr = min(nproc, 64)
            runs.append(("scipy OpenBLAS (1 BLAS thread per call)", nb_thr) + run(Lb, nb_thr, budget_s * 0.4))
    except Exception:
        pass
    if v1 is None:
        v1 = ("own vectorised loops",) + run(La, 1, budget_s * 0.2)
    best = max(runs, key=lambda r: r[2])
    granted = float(nproc)                       # CPUs this job may actually use: affinity mask, capped by the cgroup quota
    try:
        q, per = cgroup.split()
        if q != "max":
            granted = min(granted, float(q) / float(per))
    except ValueError:
        pass
    return {"value": best[2], "unit": "samples/s", "cores": granted, "threads": best[1], "kind": "port",
            "sample": "; ".join("%d reads x %d samples in %.1f s on %d OpenMP threads over reads, BLAS = %s: %.3g samples/s"
                                % (r[4], len(base_reads[0]), r[5], r[1], r[0], r[2]) for r in runs)
                      + "; 1 thread (%s): %.3g samples/s; schedule(dynamic) over reads as scrappie_raw.c:355; rgrgr_r94-shaped "
                        "synthetic weights; scalar C oracle -O3 -march=native; value = the faster configuration" % (v1[0], v1[1]),
            "value_1thread": v1[1], "kbases_per_s": best[3], "host_cpus": nproc, "cgroup_cpu_max": cgroup,
            "all_runs": [{"blas": r[0], "threads": r[1], "samples_per_s": r[2]} for r in runs]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--reads", type=int, default=10000, help="reads per GPU per step")
    ap.add_argument("--samples", type=int, default=4000)
    ap.add_argument("--model", default="rgrgr_r94")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the host-to-host and HMM-posterior regions")
    args = ap.parse_args()

    # HIP gives a process 4 hardware queues by default and maps its streams onto them round robin; an engine owns 4
    # streams, torch and RCCL bring their own, and two of the engine's streams on one queue serialise work that is
    # meant to overlap (measured under the process-group path: 29.7 against 28.2 ms per step).  Must be set before the
    # HIP runtime initialises, i.e. before torch is imported; scrappie_amd sets the same default when it loads.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    distributed = world > 1 or bool(os.environ.get("BENCH_FORCE_DIST"))
    # test hooks (never set by the driver): BENCH_BACKEND=gloo and BENCH_DEVICE=0 let two ranks share
    # one GPU so that the multi-rank path can be exercised on a single-GPU box; BENCH_FORCE_DIST=1 takes the
    # process-group path (RCCL communicator and its streams) with a single rank
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
    numa_info = None
    try:                         # where this rank's GPU hangs (the library places its pinned staging there: sh_numa.h)
        L_ = sa.lib()
        L_.scrappie_hip_device_numa_node.restype = C.c_int
        node = int(L_.scrappie_hip_device_numa_node(local_rank))
        numa_info = {"rank0_gpu_numa_node": node, "host_threads_per_engine": int(L_.scrappie_hip_host_thread_budget()),
                     "local_world_size": int(os.environ.get("LOCAL_WORLD_SIZE", "1"))}
    except Exception:
        pass
    eng.load_model(args.model, weights)
    eng.set_max_launch_reads(max(16384, args.reads))

    # this rank's shard of the global read set (weak scaling: args.reads per GPU)
    total_reads = args.reads * world
    lo, hi = shard_range(total_reads, world, rank)
    n = hi - lo
    flat, base = make_reads(lo, n, args.samples, seed=1, events=events)
    d_sig = eng.upload(flat)
    off = np.arange(n, dtype=np.uint64) * np.uint64(args.samples * (12 if events else 1))
    ln = np.full(n, args.samples, np.uint32)
    params = eng.default_params()

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        eng.synchronize()

    run_model = [args.model]

    def enqueue():
        eng.run_device(d_sig, off, ln, run_model[0], params)

    def finish():
        """collect the OLDEST launch group in flight + the stage timings its events recorded"""
        nb = eng.collect(n, params, raw=True)
        return nb, eng.timing()

    rank_dt = [None]            # the last timed region's wall time on every rank (a SCALE run should show skew, not just the maximum)

    def reduce_max_sum(dt, nb):
        rank_dt[0] = [dt]
        if distributed:
            mine = torch.zeros(world, dtype=torch.float64, device=red_dev)
            mine[rank] = dt
            dist.all_reduce(mine, op=dist.ReduceOp.SUM)
            rank_dt[0] = [float(v) for v in mine.tolist()]
            t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            b = torch.tensor([nb], dtype=torch.float64, device=red_dev)
            dist.all_reduce(b, op=dist.ReduceOp.SUM)
            return float(t.item()), float(b.item())
        return dt, float(nb)

    def device_resident_region(steps, acc=None):
        """K steps, software pipelined: the engine holds two launch groups, so step k+1's kernels are
        enqueued before the host waits for / stitches step k.  Everything of all K steps (enqueue,
        kernels, D2H, stitching) happens between the barriers."""
        nbases = 0
        marks = []
        barrier()
        t0 = time.perf_counter()
        if steps > 0:
            enqueue()
        for k in range(steps):
            if k + 1 < steps:
                enqueue()
            nb, tm = finish()
            nbases += nb
            if acc is not None:
                acc(tm)
            marks.append(time.perf_counter() - t0)
        barrier()
        dt = time.perf_counter() - t0
        if os.environ.get("SH_BENCH_MARKS"):
            print("step ends (ms): " + " ".join("%.1f" % (m * 1e3) for m in marks) + "; region %.1f" % (dt * 1e3), file=sys.stderr)
        return reduce_max_sum(dt, nbases)

    # warm-up: the same pipelined form as the timed region, stage-timing events included (the first launch
    # groups after an idle spell run at ramping clocks: measured 38 / 36 / 39 ms before the steady 30.5)
    eng.set_profiling(True)
    device_resident_region(args.warmup)
    gru = [0.0, 0, 0.0]         # ms, launches, FLOPs of the recurrent kernels
    fused = [0.0, 0, 0.0]       # ... of the one-kernel layers (k_gru_proj: projection + recurrence)
    stage = {}

    def acc(tm):
        gru[0] += tm["gru_ms"]; gru[1] += tm["n_gru_launches"]; gru[2] += tm["gru_flops"]
        fused[0] += tm["fused_ms"]; fused[1] += tm["n_fused_launches"]; fused[2] += tm["fused_flops"]
        for key in ("conv_ms", "affine_ms", "gru_ms", "ff_ms", "decode_ms", "backtrace_ms", "stitch_ms", "total_ms"):
            stage[key] = stage.get(key, 0.0) + tm[key]

    # ---- region 1: the contract's timed region (inputs resident in HBM)
    dt, nbases = device_resident_region(args.steps, acc)
    value_rank_dt = list(rank_dt[0])
    eng.set_profiling(False)
    gru_ms, gru_launches, gru_flops = gru

    # ---- region 2: SURVEY 8(d) host -> host: pageable host signals in, base strings out
    h2h = None
    if not args.no_extra and not events and args.steps > 0:
        L = sa.lib()
        G = max(1, min(args.steps, 40))                          # launch groups (= steps' worth of reads) in the one call: all K steps
        host_sig = np.tile(flat, G)                              # ordinary pageable memory, G x the step's reads
        ntot = n * G
        rts = (sa._RawTable * ntot)()
        base_addr = host_sig.ctypes.data
        # (filled through a numpy view of the table: a Python loop over 400 000 ctypes structs takes seconds)
        rt_np = np.frombuffer(rts, dtype=np.dtype([("uuid", np.uint64), ("n", np.uint64), ("start", np.uint64), ("end", np.uint64),
                                                   ("raw", np.uint64)]))
        assert rt_np.itemsize == C.sizeof(sa._RawTable)
        rt_np["uuid"] = 0; rt_np["n"] = args.samples; rt_np["start"] = 0; rt_np["end"] = args.samples
        rt_np["raw"] = base_addr + 4 * args.samples * np.arange(ntot, dtype=np.uint64)
        calls = (sa._Call * ntot)()
        eng.set_max_launch_reads(max(n, 16))                     # one step's reads per launch group, as in region 1

        def h2h_call(cnt):
            if L.scrappie_hip_basecall_batch(eng._h, eng._models[args.model], rts, cnt, C.byref(params), calls) != 0:
                raise RuntimeError("basecall_batch: " + sa.last_error())
            nb = int(np.frombuffer(calls, dtype=np.uint64).reshape(ntot, C.sizeof(sa._Call) // 8)[:cnt, 3].sum())
            L.scrappie_hip_free_calls(calls, cnt)
            return nb

        h2h_call(min(ntot, 2 * n))                               # warm-up: both staging buffers
        barrier()
        t0 = time.perf_counter()
        nb2 = h2h_call(ntot)
        barrier()
        dt2, nb2 = reduce_max_sum(time.perf_counter() - t0, nb2)
        eng.set_max_launch_reads(max(16384, args.reads))
        h2h = {"value": float(total_reads) * G * args.samples / dt2, "unit": "samples/s",
               "ms_per_step": dt2 / G * 1e3, "steps": G, "kbases_per_s": nb2 / dt2 / 1e3,
               "note": "SURVEY 8(d) definition: normalised signal in pageable host memory -> base strings in host memory; ONE "
                       "scrappie_hip_basecall_batch call over %d steps' worth of reads between barriers (gather into pinned "
                       "staging, H2D, kernels, D2H, host stitching; the engine cuts the call into launch groups of one step's "
                       "reads and keeps two in flight)" % G}

    # ---- BASELINE config 2 AS WRITTEN ("batch=64"): the same reads handed over 64 at a time through scrappie_hip_basecall_batch, from one
    # host thread and from 64 (the reference's `#pragma omp parallel for schedule(dynamic)`, scrappie_raw.c:355,387, with a batched body)
    b64 = None
    if not args.no_extra and not events and args.steps > 0 and world == 1:
        import threading
        L = sa.lib()
        per = 64
        n64 = (n // per) * per
        host64 = np.ascontiguousarray(flat[:n64 * args.samples])
        rts64 = (sa._RawTable * n64)()
        rt_np = np.frombuffer(rts64, dtype=np.dtype([("uuid", np.uint64), ("n", np.uint64), ("start", np.uint64), ("end", np.uint64), ("raw", np.uint64)]))
        rt_np["uuid"] = 0; rt_np["n"] = args.samples; rt_np["start"] = 0; rt_np["end"] = args.samples
        rt_np["raw"] = host64.ctypes.data + 4 * args.samples * np.arange(n64, dtype=np.uint64)
        calls64 = (sa._Call * n64)()
        ncalls = n64 // per
        szr, szc = C.sizeof(sa._RawTable), C.sizeof(sa._Call)
        mh = eng._models[args.model]

        def one_call(k):
            r = (sa._RawTable * per).from_address(C.addressof(rts64) + k * per * szr)
            c = (sa._Call * per).from_address(C.addressof(calls64) + k * per * szc)
            if L.scrappie_hip_basecall_batch(eng._h, mh, r, per, C.byref(params), c) != 0:
                raise RuntimeError("basecall_batch: " + sa.last_error())

        def run_threads(nthr, calls_to_make):
            nxt = [0]
            lock = threading.Lock()
            errs = []

            def body():
                try:
                    while True:
                        with lock:
                            k = nxt[0]; nxt[0] += 1
                        if k >= calls_to_make:
                            return
                        one_call(k)
                except Exception as ex:          # noqa: BLE001
                    errs.append(repr(ex))
            barrier()
            t0 = time.perf_counter()
            th = [threading.Thread(target=body) for _ in range(nthr)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            barrier()
            dtb = time.perf_counter() - t0
            if errs:
                raise RuntimeError(errs[0])
            nb = int(np.frombuffer(calls64, dtype=np.uint64).reshape(n64, szc // 8)[:calls_to_make * per, 3].sum())
            L.scrappie_hip_free_calls(calls64, calls_to_make * per)
            return dtb, nb
        try:
            st0 = (C.c_ulonglong * 3)(); st1 = (C.c_ulonglong * 3)()
            run_threads(64, min(ncalls, 128))                  # warm-up (threads seen, arenas)
            L.scrappie_hip_batch_coalescer_stats(st0)
            dt64, nb64 = run_threads(64, ncalls)
            L.scrappie_hip_batch_coalescer_stats(st1)
            eng_calls64, caller_calls64 = int(st1[0] - st0[0]), int(st1[1] - st0[1])
            run_threads(256, ncalls)                           # (warm-up for the wider team)
            L.scrappie_hip_batch_coalescer_stats(st0)
            dt256, nb256 = run_threads(256, ncalls)
            st2 = (C.c_ulonglong * 3)()
            L.scrappie_hip_batch_coalescer_stats(st2)
            n1 = min(ncalls, 24)                               # one thread: a call is a launch group of 64 reads -- its chain, ~12 ms, whatever it holds
            dt1, _ = run_threads(1, n1)
            b64 = {"reads_per_call": per, "calls": ncalls, "reads": n64,
                   "threads_64": {"value": n64 * args.samples / dt64, "unit": "samples/s", "wall_ms": dt64 * 1e3,
                                  "engine_calls": eng_calls64, "caller_calls": caller_calls64, "kbases_per_s": nb64 / dt64 / 1e3},
                   "threads_256": {"value": n64 * args.samples / dt256, "unit": "samples/s", "wall_ms": dt256 * 1e3,
                                   "engine_calls": int(st2[0] - st0[0]), "caller_calls": int(st2[1] - st0[1])},
                   "threads_1": {"value": n1 * per * args.samples / dt1, "unit": "samples/s", "calls_timed": n1, "ms_per_call": dt1 / n1 * 1e3},
                   "note": "BASELINE config 2 as written: %d reads submitted as %d scrappie_hip_basecall_batch calls of 64 (host signals in, base strings "
                           "out, between barriers).  64 host threads taking the next call from a shared counter (the reference's schedule(dynamic) loop, "
                           "scrappie_raw.c:355,387): concurrent small calls share launch groups through the coalescing queue (sh_coalesce.h; "
                           "engine_calls = engine calls actually made).  What a caller gets is bounded by the reads in flight = threads x 64 (a launch group lasts one read's chain "
                           "whatever it holds; the device has 512 tile slots of 16 reads): 64 threads keep 4096 reads in flight, 256 threads 16 384.  One thread: each call is a launch group of 64 reads -- 4 of the device's 512 tile "
                           "slots for one chain's duration; the streaming / deferred entry points are the single-thread form" % (n64, ncalls)}
            # ... and from C: the reference's OpenMP loop itself (tools/batch64_omp.c, a process of its own with its own engine on the same GPU)
            try:
                import re
                import shutil
                from scrappie_amd import model as _model
                tmpd = tempfile.mkdtemp(prefix="sh_b64_")
                try:
                    exe = os.path.join(tmpd, "batch64_omp")
                    subprocess.run(["gcc", "-O2", "-std=gnu11", "-fopenmp", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "batch64_omp.c"), "-o", exe,
                                    "-L" + os.path.join(ROOT, "scrappie_amd"), "-lscrappie_hip", "-Wl,-rpath," + os.path.join(ROOT, "scrappie_amd"), "-lm"], check=True, capture_output=True)
                    mpath, spath = os.path.join(tmpd, "m.scrm"), os.path.join(tmpd, "s.f32")
                    _model.save_model(weights, mpath)
                    host64.tofile(spath)
                    omp = {}
                    for nthr in (64, 256):
                        # (passive waiting: the threads that have no call left wait in the loop's closing barrier, and 256 of them spinning on 16 CPUs starve the one that leads the launch)
                        r = subprocess.run([exe, mpath, spath, str(n64), str(args.samples), str(nthr), "3"], capture_output=True, text=True, timeout=300,
                                           env=dict(os.environ, OMP_WAIT_POLICY="passive", GOMP_SPINCOUNT="0"))
                        if r.returncode != 0:
                            omp["threads_%d" % nthr] = {"error": r.stderr[-300:]}
                            continue
                        runs = [(float(m_.group(1)), int(m_.group(2))) for m_ in re.finditer(r"samples_per_s ([0-9.e+]+) engine_calls (\d+)", r.stdout)]
                        best = max(runs)
                        omp["threads_%d" % nthr] = {"value": best[0], "unit": "samples/s", "engine_calls": best[1], "runs": [v for v, _ in runs]}
                    b64["openmp"] = omp
                    b64["openmp"]["note"] = ("the same 156 calls of 64 reads from `#pragma omp parallel for schedule(dynamic)` in C (tools/batch64_omp.c; the best of three "
                                             "repetitions): OpenMP threads reach the queue together, Python threads over milliseconds")
                finally:
                    shutil.rmtree(tmpd, ignore_errors=True)
            except Exception as ex:              # noqa: BLE001
                b64["openmp"] = {"error": str(ex)}
        except Exception as ex:                  # noqa: BLE001
            b64 = {"error": str(ex)}

    # ---- region 3: realistic calls through the default kernels (decode driven by HMM-like posteriors)
    hmm = None
    if not args.no_extra and not events and model.model_dims(weights)["NS"] == 1025 and model.model_dims(weights)["S"] >= 21 and args.steps > 0:
        from scrappie_amd import synth
        d_ = model.model_dims(weights)
        nblk = (args.samples + weights["stride"] - 1) // weights["stride"]
        w_hmm = dict(weights)
        w_hmm["ff_W"], w_hmm["ff_b"] = synth.hmm_output_layer(S=d_["S"])
        eng.load_model("hmm_output_layer", w_hmm)
        trunks = [synth.hmm_trunk(nblk, 900 + i + 100 * rank, S=d_["S"], plant_homopolymers=4)[0] for i in range(32)]
        eng.set_trunk_input(trunks)
        run_model[0] = "hmm_output_layer"
        enqueue(); finish()                                      # builds the chunk-layout image of the activations (untimed)
        eng.set_profiling(True)
        st3 = {}

        def acc3(tm):
            for key in ("decode_ms", "backtrace_ms", "stitch_ms", "total_ms", "gru_ms"):
                st3[key] = st3.get(key, 0.0) + tm[key]
        dt3, nb3 = device_resident_region(args.steps, acc3)
        eng.set_profiling(False)
        eng.set_trunk_input(None)
        run_model[0] = args.model
        hmm = {"kbases_per_s": nb3 / dt3 / 1e3, "samples_per_s": float(total_reads) * args.samples * args.steps / dt3,
               "ms_per_step": dt3 / args.steps * 1e3, "bases_per_read": nb3 / (float(total_reads) * args.steps),
               "stage_ms_per_step": {k: v / args.steps for k, v in st3.items()},
               "note": "same device-resident step on the DEFAULT kernels (k_ff_viterbi: S1 inside the decoder): the whole network "
                       "runs, then the output layer -- built from +-1 state codes (scrappie_amd.synth.hmm_output_layer) -- reads trunk "
                       "activations encoding a simulated k-mer path (55 %% stay / 40 %% step / 5 %% skip, planted homopolymers, 32 "
                       "distinct, %d blocks) through scrappie_hip_set_trunk_input; posteriors are computed by the production S1 "
                       "(mean max p ~0.55); bases are counted from the calls" % nblk}

    # ---- region 4: the same step with the five recurrent layers on the exact-fp32 kernels (v_mfma_f32_16x16x4_f32 throughout:
    # k_affine<.., F32> + k_gru_lanes), the path a layer with |w| >= 255 takes: the like-for-like reading of "fp32 peak"
    f32r = None
    if not args.no_extra and not events and weights["arch"] in ("rgrgr", "rnnrf") and args.steps > 0:
        eng.debug_option("force_f32_layers", 1)
        eng.load_model("exact_fp32", weights)
        eng.debug_option("force_f32_layers", 0)
        k4 = max(2, min(args.steps, 10))
        run_model[0] = "exact_fp32"
        enqueue(); finish()
        eng.set_profiling(True)
        st4 = {}

        def acc4(tm):
            for key in ("affine_ms", "gru_ms", "decode_ms", "total_ms", "affine_flops", "gru_flops"):
                st4[key] = st4.get(key, 0.0) + tm[key]
        dt4, _ = device_resident_region(k4, acc4)
        eng.set_profiling(False)
        run_model[0] = args.model
        lay_ms = (st4["affine_ms"] + st4["gru_ms"]) / k4
        lay_tf = (st4["affine_flops"] + st4["gru_flops"]) / k4 / (lay_ms * 1e-3) / 1e12
        dp = None
        try:                                                 # the two forms' posteriors of the same reads
            dp = 0.0
            for r in base[:4]:
                a = np.asarray(eng.posterior(r, model=args.model, log=False)); b = np.asarray(eng.posterior(r, model="exact_fp32", log=False))
                dp = max(dp, float(np.abs(a - b).max()))
        except Exception as ex:                              # (models without a posterior surface)
            dp = None
        f32r = {"ms_per_step": dt4 / k4 * 1e3, "value": float(total_reads) * args.samples * k4 / dt4, "unit": "samples/s", "steps": k4,
                "recurrent_layers_ms_per_step": lay_ms, "recurrent_layers_tflops": lay_tf,
                "frac_of_157_TFLOPs": lay_tf / FP32_MFMA_PEAK_TFLOPS,
                "roofline": {"kernel": "k_affine<.., F32> + k_gru_lanes (projection and recurrence of the five layers on v_mfma_f32_16x16x4_f32)", "bound": "mfma",
                             "achieved": lay_tf, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": lay_tf / FP32_MFMA_PEAK_TFLOPS,
                             "peak_note": "dense fp32 MFMA peak (MI355X_MICROARCH.md); algorithmic FLOPs of the layers / HIP-event time of their kernels"},
                "max_abs_dp_vs_split_products": dp,
                "note": "the five recurrent layers (projection + recurrence) on exact-fp32 MFMAs (k_affine<..,F32> + k_gru_lanes: what a layer "
                        "with a weight outside the split products' range runs on; gate inputs through HBM); S1 + decoder unchanged; "
                        "frac = the layers' algorithmic FLOP/s over the dense fp32 MFMA peak"}

    # ---- the per-read reference surface from many host threads (the reference's own loop body, scrappie_raw.c:265-315): not the metric, a record
    prs = None
    if not args.no_extra and not events and rank == 0 and world == 1 and weights["arch"] == "rgrgr" and args.model in sa._model_fn_ and args.steps > 0 \
            and model.model_dims(weights)["NS"] == 1025:
        from concurrent.futures import ThreadPoolExecutor
        try:
            mpath = os.path.join(tempfile.mkdtemp(), args.model + ".scrm")
            model.save_model(weights, mpath)
            sa.register_model(args.model, mpath)
            rts = [sa.RawTable(r) for r in base[:64]]

            def body(i):
                post = sa.calc_post(rts[i % len(rts)], args.model, min_prob=1e-5, log=True)
                return sa._decode_post(post, local_pen=150.0)
            for i in range(2):
                body(i)
            t0 = time.time()
            n1 = 24
            for i in range(n1):
                body(i)
            dt1 = time.time() - t0
            nthr, nrd = 64, 1024
            t0 = time.time()
            with ThreadPoolExecutor(nthr) as pool:
                calls = list(pool.map(body, range(nrd)))
            dtn = time.time() - t0
            prs = {"threads": nthr, "reads_per_s": nrd / dtn, "samples_per_s": nrd * args.samples / dtn, "reads_per_s_one_thread": n1 / dt1,
                   "mean_call_length": float(np.mean([len(c[0] or "") for c in calls])),
                   "note": "nanonet_%s_posterior + decode_transducer + overlapper per read (a 1025 x T host matrix per read, as the reference's API has it), "
                           "called from Python threads; concurrent calls are coalesced into launch groups (INTEGRATION.md, profiles/r4_per_read_surface.txt)" % args.model}
        except Exception as ex:
            prs = {"error": str(ex)}

    if rank == 0:
        samples_total = float(total_reads) * args.samples * args.steps
        value = samples_total / dt
        d = model.model_dims(weights)
        # dominant kernel: a recurrent layer.  For the GRU stacks a layer is one kernel (k_gru_proj: projection
        # team + recurrence team, FLOPs = projection + recurrence) whose contractions run as split products
        # on the f16 matrix pipe: three f16 MFMA flops per algorithmic fp32 flop, so the fp32-equivalent
        # roof of that pipe is its dense f16 peak / 3.  The events LSTM still runs exact-fp32 MFMAs.
        is_fused = fused[1] > 0
        if is_fused:
            gru_ms, gru_launches, gru_flops = fused
        gru_avg_ms = gru_ms / max(gru_launches, 1)
        achieved = (gru_flops / max(gru_launches, 1)) / (gru_avg_ms * 1e-3) / 1e12 if gru_ms > 0 else 0.0
        split = True                              # every recurrent layer (GRU and LSTM) runs split products
        # the whole step's algorithmic FLOPs per GPU (convolution + projections + recurrences + output layer)
        nblk_ = (args.samples + d["stride"] - 1) // d["stride"]
        whole_flops = float(model.flops_per_block(weights)) * float(total_reads // world) * nblk_
        peak = F16_MFMA_PEAK_TFLOPS / SPLIT_PRODUCTS
        out = {
            "metric": ("events/sec, %s bi-LSTM (SURVEY 8(f).4; not the headline metric)" % args.model) if events
                      else "raw samples/sec, %s %dk-sample reads%s" % (args.model, args.samples // 1000,
                                                                       "" if args.model == "rgrgr_r94" else " (not the headline model)"),
            "value": value,
            "unit": "events/s" if events else "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "ms_per_step_per_rank": {"max": max(value_rank_dt) / args.steps * 1e3, "min": min(value_rank_dt) / args.steps * 1e3,
                                     "all": [v / args.steps * 1e3 for v in value_rank_dt],
                                     "note": "wall time of the timed region on each rank / steps (barrier to barrier: ranks that finish early wait in the "
                                             "closing barrier, so the spread is what the slowest rank's GPU or host share costs the others)"} if args.steps > 0 else None,
            "numa": numa_info,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 (3xf16 split products, 22-bit operands, fp32 accumulate)",
            "dtype_note": "all tensors, accumulators and results are fp32; the projection / recurrence / S1 contractions execute "
                          "as three f16 partial products of two-piece fp16 splits (22 bits) of their up-scaled fp32 operands, accumulated in "
                          "fp32 (error at or below an fp32 FMA chain's: profiles/r2_split_probe.txt), inside the operand range checked at "
                          "model load (|w| < 255; else the exact-fp32 kernels run); every parity test runs at the fp32 tolerances, incl. "
                          "against independent float64 fixtures, and on adversarial operands against float64 and the exact-fp32 layer "
                          "(tests/test_gpu_round6.py::test_split_products_worst_case); exact_fp32 below is the same step with the layers on exact-fp32 MFMAs",
            "data": "synthetic",
            "config": {"workload": "%s raw, %d synthetic %d-sample reads per GPU per step, handed to the engine in one call "
                                   "(one launch group), %dxMI355X" % (args.model, args.reads, args.samples, world),
                       "reads_per_gpu_per_step": args.reads, "samples_per_read": args.samples,
                       "blocks_per_read": (args.samples + d["stride"] - 1) // d["stride"],
                       "dims": d, "weights": "synthetic (reference model headers are missing blobs)"},
            "kbases_per_s": nbases / dt / 1e3,
            "kbases_note": "as called on synthetic weights (degenerate for transducer models: SURVEY.md section 7); "
                           "kbases_per_s_hmm_posteriors is the measured rate with a realistic decode",
            "value_note": "inputs resident in HBM when the timed region starts: the bench contract defines `value` so and rules "
                          "out a PCIe-inclusive figure there; SURVEY 8(d)'s host-to-host rate (pageable host signal -> base strings), "
                          "measured in the same run, is host_to_host.value",
            "kbases_per_s_hmm_posteriors": None,
            "roofline": {"kernel": (("k_lstm_proj<%d> (projection + peephole LSTM of one direction of one level)" if is_fused else "k_lstm_lanes<%d>") if events
                                    else ("k_gru_proj<%d> (projection + recurrence of one layer)" if is_fused else "k_gru_split<%d>")) % (d["S"] // 16),
                         "bound": "mfma", "achieved": achieved,
                         "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak,
                         "whole_step_frac": whole_flops / (dt / args.steps) / 1e12 / peak if not events else None,
                         "whole_step_tflops": whole_flops / (dt / args.steps) / 1e12 if not events else None,
                         "peak_note": ("dense f16 MFMA peak %.0f TFLOP/s / 3: every fp32 product is three f16 partial products of "
                                       "two-piece splits, accumulated in fp32 (tools/split_probe.hip: closer to float64 than the fp32 MFMA)"
                                       % F16_MFMA_PEAK_TFLOPS) if split else "dense fp32 MFMA peak",
                         "achieved_over_f32_mfma_peak": achieved / FP32_MFMA_PEAK_TFLOPS,
                         "traffic": None if events else measured_traffic("k_gru_proj" if is_fused else "k_gru_split", args),
                         "traffic_unit": "HBM bytes per launch (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)",
                         "traffic_note": None,
                         "algorithmic_bytes": float(total_reads // world) * ((args.samples + d["stride"] - 1) // d["stride"])
                                              * ((2.0 if is_fused else 5.0) if events else (2.0 if is_fused else 4.0)) * d["S"] * 4,
                         "avg_launch_ms": gru_avg_ms,
                         # "also report GRU steps/s per CU since latency ... is the practical limiter":
                         # recurrence steps (one read, one block, one layer) per second and CU
                         "gru_read_steps_per_s_per_cu": (float(total_reads // world) * ((args.samples + d["stride"] - 1) // d["stride"])
                                                         / (gru_avg_ms * 1e-3) / 256.0) if gru_avg_ms > 0 else None,
                         "flops_per_launch": gru_flops / max(gru_launches, 1),
                         "note": "algorithmic FLOPs per read per block = 2*3*S*S (recurrence) + 2*S*3S (the layer's input projection, "
                                 "same kernel), 2*4*S*S (LSTM); bytes = S in + S out (gate inputs stay in LDS), 4S in + S out (LSTM); "
                                 "(SURVEY 8d); the first events level reads 12 features instead of S; HIP events on the engine's stream; rank 0"},
            "stage_ms_per_step": {k: v / args.steps for k, v in stage.items() if not k.startswith("_")},
            "stitching": "host threads (SH_HOST_STITCH=1: paths + 5 posterior rows over PCIe, 24 B per block)" if os.environ.get("SH_HOST_STITCH")
                         else "device (one copy-stream kernel behind the decoder, k_walk_stitch_out: traceback walk, homopolymer pass, stitching, "
                              "and the called bases written to pinned host memory -- only they cross PCIe; stitch_ms is that kernel and "
                              "backtrace_ms 0 (SH_SPLIT_TAIL=1: k_backtrace, k_stitch, k_results_out timed apart); conv_ms is the NEXT group's "
                              "convolution on the prologue stream beside this group's recurrent layers; none of the three is part of "
                              "total_ms, which is the main stream's layers + decoder)",
        }
        out["roofline"]["traffic_note"] = _TRAFFIC_NOTE[0]
        s1_in_decoder = (not events and stage.get("decode_ms") and stage.get("ff_ms", 0.0) / args.steps < 0.05
                         and d["NS"] == 1025 and d["S"] == 96)
        if s1_in_decoder:
            # k_ff_viterbi_teams: S1 + decoder in one kernel (an S1 producer team and a decoder team), the posterior never in memory.  Its HBM
            # traffic is the trunk output in and one traceback byte per state out
            nblk = (args.samples + d["stride"] - 1) // d["stride"]
            cols = float(total_reads // world) * nblk
            fv_bytes = cols * (d["S"] * 4.0 + (d["NS"] - 1) + 4.0)
            fv_ms = stage["decode_ms"] / args.steps
            s1_flops = 2.0 * d["S"] * ((d["NS"] + 15) // 16 * 16) * cols
            tr = measured_traffic("k_ff_viterbi_teams", args)
            out["roofline_other"] = {
                "k_ff_viterbi_teams": {"bound": "the decoder team's instruction streams (8 decoder waves + 4 S1 producer waves per workgroup; 26.3 lane-instructions per state "
                                          "and block, VALU 69 % busy; profiles/r5_decoder_teams_v2.txt)",
                                 "avg_launch_ms": fv_ms,
                                 "algorithmic_bytes": fv_bytes, "hbm_achieved_GBps": fv_bytes / (fv_ms * 1e-3) / 1e9,
                                 "hbm_frac": fv_bytes / (fv_ms * 1e-3) / 1e9 / 8000.0,
                                 "traffic": tr,
                                 "s1_flops_per_launch": s1_flops,
                                 "mfma_frac": s1_flops / (fv_ms * 1e-3) / 1e12 / (F16_MFMA_PEAK_TFLOPS / SPLIT_PRODUCTS),
                                 "replaces": "k_ff_lds (33.6 GB written) + k_viterbi (33.6 GB read): 67 GB of posterior per launch "
                                             "that no longer exist; SH_FF_SEPARATE=1 runs that form (identical results)"},
                "note": "algorithmic bytes per launch = S floats in + 1 traceback byte per state + end pointer out; stage time from HIP events; "
                        "traffic from the committed PMC passes (roofline.traffic_note)"}
        elif not events and stage.get("ff_ms") and stage.get("decode_ms") and d["NS"] > 25:
            # the two HBM-bound kernels: algorithmic bytes per launch / HIP-event time of the stage
            nblk = (args.samples + d["stride"] - 1) // d["stride"]
            cols = float(total_reads // world) * nblk
            mt = (d["NS"] + 15) // 16 * 16
            s1_bytes = cols * (d["S"] + mt + 1) * 4.0                  # S in + 1040 exp values + 1 row sum out
            vit_bytes = cols * ((mt + 1) * 4.0 + (d["NS"] - 1) + 4.0)  # 1040 exp values + sum in, 1 traceback byte per state + end pointer out
            out["roofline_hbm"] = {
                "k_ff_lds": {"bound": "hbm", "achieved": s1_bytes / (stage["ff_ms"] / args.steps * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s"},
                "k_viterbi": {"bound": "hbm", "achieved": vit_bytes / (stage["decode_ms"] / args.steps * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s"},
                "note": "algorithmic bytes per launch / stage time from HIP events; peak = HBM3E spec (MI355X_MICROARCH.md: 8 TB/s, "
                        "6.3 TB/s measured for a float4 copy)"}
            for k in ("k_ff_lds", "k_viterbi"):
                out["roofline_hbm"][k]["frac"] = out["roofline_hbm"][k]["achieved"] / 8000.0
        if hmm:
            out["kbases_per_s_hmm_posteriors"] = hmm["kbases_per_s"]
            out["hmm_posteriors"] = hmm
        if h2h:
            out["host_to_host"] = h2h
            out["value_host_to_host"] = h2h["value"]       # SURVEY 8(d)'s metric, promoted: `value` is HBM-resident by the bench contract
        if b64:
            out["batch64"] = b64
            for k_ in ("threads_64", "threads_256"):
                if k_ in b64:
                    b64[k_]["frac_of_value"] = b64[k_]["value"] / value
                if k_ in b64.get("openmp", {}) and "value" in b64["openmp"][k_]:
                    b64["openmp"][k_]["frac_of_value"] = b64["openmp"][k_]["value"] / value
        if f32r:
            out["exact_fp32"] = f32r
        if prs:
            out["per_read_surface"] = prs
    eng.free(d_sig)
    eng.close()
    if rank == 0:
        if not args.no_extra and not events and world == 1 and weights["arch"] in ("rgrgr", "rnnrf") and args.steps > 0 and args.samples >= 1000:
            # the command line end to end, with the GPU to itself (this process's engine is gone)
            # on the one input format the reference reads (fast5), and on headerless float32 files; same reads, so the same sequences
            e5 = cli_end_to_end(weights, args.model, 40 * args.reads, args.samples, "fast5")
            e32 = cli_end_to_end(weights, args.model, 40 * args.reads, args.samples, "f32")
            for rec in (e5, e32):
                if "value" in rec:
                    rec["frac_of_value"] = rec["value"] / out["value"]
            if "value" in e5:
                out["cli_end_to_end"], out["cli_end_to_end_f32"] = e5, e32
                if "sequences_md5" in e32:
                    e5["same_sequences_as_f32_input"] = e5["sequences_md5"] == e32["sequences_md5"]
            else:
                out["cli_end_to_end"], out["cli_end_to_end_fast5"] = e32, e5
        if not args.no_cpu_baseline and world == 1 and not events:
            out["cpu_baseline"] = cpu_baseline(weights, base)
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
