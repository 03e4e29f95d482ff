#!/usr/bin/env python3
"""Do launch groups of two engines on ONE device run side by side?  Each engine gets one chain-bound group (a few
long reads + many short ones, inputs resident in HBM); timed one after the other, then from two host threads at once.
usage: concurrent_probe.py [reads=3000] [long=200000] [short=4000]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scrappie_amd as sa
from scrappie_amd import model, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
name = "rgrgr_r94"
lens = np.full(n, ns, np.uint32)
lens[:16] = nl
long_sig = synth.medmad_normalise(synth.synthetic_signal(nl + 64 * 7, 5))
off = np.zeros(n, np.uint64)
off[1:] = np.cumsum(lens[:-1].astype(np.uint64))
flat = np.empty(int(lens.sum()), np.float32)
for i in range(n):
    s = (i % 64) * 7
    flat[int(off[i]):int(off[i]) + int(lens[i])] = long_sig[s:s + int(lens[i])]
engs, bufs = [], []
for k in range(2):
    e = sa.Engine(0)
    e.load_model(name, model.synthetic_model(name, seed=1))
    engs.append(e); bufs.append(e.upload(flat))


def one(k, reps):
    for r in range(reps):
        engs[k].run_device(bufs[k], off, lens, name)
        engs[k].collect(n, raw=True)


for k in range(2):
    one(k, 1)
t0 = time.perf_counter(); one(0, 2); one(1, 2); t_seq = time.perf_counter() - t0
th = [threading.Thread(target=one, args=(k, 2)) for k in range(2)]
t0 = time.perf_counter()
for t in th: t.start()
for t in th: t.join()
t_par = time.perf_counter() - t0
print("%d reads (16 of %d samples, the rest %d): 4 groups one after the other %.1f ms, two engines side by side %.1f ms" % (n, nl, ns, t_seq * 1e3, t_par * 1e3))
