#!/usr/bin/env python3
"""Wall time of every step of a pipelined run (two launch groups in flight), to see where a constant
per-region cost comes from.  usage: python tools/step_times.py [steps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scrappie_amd as sa
from scrappie_amd import model
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_reads
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mode = sys.argv[2] if len(sys.argv) > 2 else ""
if "libfirst" in mode:
    sa.lib()                      # binds /opt/rocm's HIP runtime before torch brings its bundled one
if "torch" in mode or "import" in mode:
    import torch
if "devsync" in mode:
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
n, ns = 10000, 4000
eng = sa.Engine(0)
eng.load_model("rgrgr_r94", model.synthetic_model("rgrgr_r94", seed=1))
eng.set_max_launch_reads(16384)
flat, base = make_reads(0, n, ns, seed=1, events=False)
d_sig = eng.upload(flat)
off = np.arange(n, dtype=np.uint64) * np.uint64(ns)
ln = np.full(n, ns, np.uint32)
params = eng.default_params()
for _ in range(2):
    eng.run_device(d_sig, off, ln, "rgrgr_r94", params); eng.collect(n, params, raw=True)
def throttled():
    try:
        return {k: int(v) for k, v in (l.split() for l in open("/sys/fs/cgroup/cpu.stat"))}
    except Exception:
        return {}
import threading
print("threads after setup:", threading.active_count(), "os threads:", len(os.listdir("/proc/self/task")))
for prof in (False, True):
    c0 = throttled()
    eng.set_profiling(prof)
    if "torch" in mode:
        torch.cuda.synchronize()
    if "devsync" in mode:
        hip.hipDeviceSynchronize()
    eng.synchronize()
    t0 = time.perf_counter(); ts = []
    eng.run_device(d_sig, off, ln, "rgrgr_r94", params)
    ts.append(time.perf_counter() - t0)
    for k in range(steps):
        if k + 1 < steps:
            eng.run_device(d_sig, off, ln, "rgrgr_r94", params)
        t1 = time.perf_counter()
        eng.collect(n, params, raw=True)
        if "timing" in mode:
            eng.timing()
        ts.append((t1 - t0, time.perf_counter() - t0))
    eng.synchronize()
    c1 = throttled()
    print("cpu.stat delta:", {k: c1[k] - c0[k] for k in c1 if k in c0 and c1[k] != c0[k]})
    print("profiling", prof, "first enqueue %.2f ms;" % (ts[0] * 1e3), " ".join("[%.1f %.1f]" % (a * 1e3, b * 1e3) for a, b in ts[1:]), "end %.1f" % ((time.perf_counter() - t0) * 1e3))
