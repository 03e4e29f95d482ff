#!/usr/bin/env python3
"""Host load of a step with REALISTIC calls: the bench's device-resident step through the
default kernels with the HMM output layer + trunk-input hook (~414 bases per read), K steps software pipelined,
for the number of stitching threads given by SCRAPPIE_HIP_HOST_THREADS.  SH_HOST_STAMP=1 makes the engine print,
per launch group, how long scrappie_hip_collect waited for the device and how long the host work took.
    SCRAPPIE_HIP_HOST_THREADS=2 SH_HOST_STAMP=1 python tools/host_stitch.py [steps] [realistic=1]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scrappie_amd as sa
from scrappie_amd import model, synth
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
realistic = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n, ns = 10000, 4000
w = model.synthetic_model("rgrgr_r94", seed=1)
if realistic:
    w["ff_W"], w["ff_b"] = synth.hmm_output_layer()
eng = sa.Engine(0)
eng.load_model("m", w)
flat, _ = bench.make_reads(0, n, ns, seed=1)
d = eng.upload(flat)
off = np.arange(n, dtype=np.uint64) * np.uint64(ns)
ln = np.full(n, ns, np.uint32)
if realistic:
    eng.set_trunk_input([synth.hmm_trunk(800, 900 + i, plant_homopolymers=4)[0] for i in range(32)])
p = eng.default_params()
for rep in range(2):            # first pass = warm-up
    eng.synchronize()
    t0 = time.perf_counter()
    eng.run_device(d, off, ln, "m", p)
    nb = 0
    for k in range(steps):
        if k + 1 < steps:
            eng.run_device(d, off, ln, "m", p)
        nb += eng.collect(n, p, raw=True)
    eng.synchronize()
    dt = time.perf_counter() - t0
print("threads=%s realistic=%d: %.2f ms per step, %.0f bases per read" % (os.environ.get("SCRAPPIE_HIP_HOST_THREADS", "default"), realistic, dt / steps * 1e3, nb / float(n * steps)))
