#!/usr/bin/env python3
"""PCIe-inclusive rate of the hot path (DESIGN.md section 7 note; never bench.py's `value`):
scrappie_hip_basecall_batch on host buffers = H2D of the signals + kernels + D2H + stitching, the call
cut into launch groups of 10 000 reads, two in flight (uploads and downloads on their own streams)."""
import ctypes as C
import sys
import time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scrappie_amd as sa
from scrappie_amd import model, synth

n, ns = (int(sys.argv[1]) if len(sys.argv) > 1 else 40000), 4000
w = model.synthetic_model("rgrgr_r94", seed=1)
eng = sa.Engine(0)
eng.load_model("rgrgr_r94", w)
eng.set_max_launch_reads(10000)
base = [synth.medmad_normalise(synth.synthetic_signal(ns, 1 + i)) for i in range(64)]
keep = [np.ascontiguousarray(base[i % 64], dtype=np.float32) for i in range(n)]
rts = (sa._RawTable * n)()
for i, s in enumerate(keep):
    rts[i] = sa._RawTable(None, len(s), 0, len(s), s.ctypes.data_as(C.POINTER(C.c_float)))
calls = (sa._Call * n)()
p = eng.default_params()
L = sa.lib()
for it in range(4):
    t0 = time.perf_counter()
    rc = L.scrappie_hip_basecall_batch(eng._h, eng._models["rgrgr_r94"], rts, n, C.byref(p), calls)
    dt = time.perf_counter() - t0
    assert rc == 0
    L.scrappie_hip_free_calls(calls, n)
    print("basecall_batch on host buffers: %.1f ms per %d reads -> %.3e samples/s" % (dt * 1e3, n, n * ns / dt))
