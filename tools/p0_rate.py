#!/usr/bin/env python3
"""k_p0 alone: signal preparation (trim_and_segment_raw + medmad_normalise_array) of N raw reads of NS samples on an idle GPU.
usage: p0_rate.py [reads=10000] [samples=4000]      (profiles/r5_p0.txt)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scrappie_amd as sa
from scrappie_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
base = [synth.synthetic_signal(ns, 100 + i, raw_units=True).astype(np.float32) for i in range(64)]
for b in base[::2]:
    b[:250] = b[:250] * np.float32(0.02) + np.float32(90)
raws = [base[i % 64] for i in range(n)]
prep = sa.Prep(0)
for rep in range(3):
    t0 = time.perf_counter()
    d, off, ln, st, en = prep.run(raws, stage_capacity=int(1.1 * n * ns))
    dt = time.perf_counter() - t0
    t = prep.timing()
print("k_p0, %d reads x %d samples (%.0f M samples, %.2f GB in + out): kernel %.3f ms = %.2e samples/s = %.2f TB/s of algorithmic traffic; host-to-device copy %.2f ms (%.1f GB/s); "
      "whole scrappie_hip_prep_run call incl. the Python-side staging copies %.1f ms; mean window %d..%d of %d" %
      (n, ns, n * ns / 1e6, 8e-9 * n * ns, t["k_p0_ms"], n * ns / (t["k_p0_ms"] * 1e-3), 8.0 * n * ns / (t["k_p0_ms"] * 1e-3) / 1e12, t["h2d_ms"],
       4e-9 * n * ns / (t["h2d_ms"] * 1e-3), dt * 1e3, int(st.mean()), int(en.mean()), ns))
prep.close()
