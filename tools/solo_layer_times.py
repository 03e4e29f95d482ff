#!/usr/bin/env python3
"""Stage times with the engine's stages serialised (set_profiling: nothing runs beside anything), for two models on one box:
what a recurrent layer costs ALONE in its plain and in its residual form.  usage: python tools/solo_layer_times.py [model ...]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scrappie_amd as sa
from scrappie_amd import model
import bench
names = sys.argv[1:] or ["rgrgr_r94", "rnnrf_r94"]
n, ns = 10000, 4000
flat, _ = bench.make_reads(0, n, ns, seed=1)
off = np.arange(n, dtype=np.uint64) * np.uint64(ns); ln = np.full(n, ns, np.uint32)
for rnd in range(2):
    for name in names:
        eng = sa.Engine(0); eng.load_model(name, model.synthetic_model(name, seed=1))
        d = eng.upload(flat)
        eng.set_profiling(True)
        for rep in range(4):
            eng.run_device(d, off, ln, name); eng.collect(n, raw=True)
            t = eng.timing()
        print(name, {k: round(v, 3) for k, v in t.items() if k.endswith("_ms")}, flush=True)
        eng.close()
