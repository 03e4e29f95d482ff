#!/usr/bin/env python3
"""Soak: random mixed-length batches through the whole path, every batch twice (the two runs must agree exactly:
hand-overs, pieces, two groups in flight are all timing dependent) and a sample of reads alone.
usage: python tools/soak.py [iterations] [model]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scrappie_amd as sa
from scrappie_amd import model, synth
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
name = sys.argv[2] if len(sys.argv) > 2 else "rgrgr_r94"
eng = sa.Engine(0)
eng.load_model(name, model.synthetic_model(name, seed=1))
rng = np.random.default_rng(123)
base = [synth.medmad_normalise(synth.synthetic_signal(int(n), 4000 + i)) for i, n in enumerate(rng.integers(30, 30000, 96))]
base += [synth.medmad_normalise(synth.synthetic_signal(int(n), 4200 + i)) for i, n in enumerate((150001, 90007))]   # two very long reads
key = lambda c: None if c is None else (c["bases"], np.float32(c["score"]).tobytes(), c["nblock"])
t0 = time.time()
for it in range(iters):
    n = int(rng.integers(300, 9000))
    pick = rng.integers(0, 96 if it % 3 else len(base), n)      # every third batch may hold the long reads
    reads = [base[j][: max(1, int(len(base[j]) * rng.uniform(0.3, 1.0)))] if rng.random() < 0.3 else base[j] for j in pick]
    kw = [dict(), dict(use_slip=1), dict(tempW=1.2, tempb=0.9), dict(homopolymer=0, local_pen=1.0)][it % 4]
    a = [key(c) for c in eng.basecall(reads, name, eng.default_params(**kw))]
    b = [key(c) for c in eng.basecall(reads, name, eng.default_params(**kw))]
    assert a == b, ("nondeterministic", it, sum(x != y for x, y in zip(a, b)))
    sub = rng.integers(0, n, 8)
    solo = [key(c) for c in eng.basecall([reads[i] for i in sub], name, eng.default_params(**kw))]
    assert solo == [a[i] for i in sub], ("batch dependence", it)
    print("iteration %d: %d reads, %s ok (%.1f s)" % (it, n, kw, time.time() - t0), flush=True)
print("soak ok")
