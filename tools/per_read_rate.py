#!/usr/bin/env python3
"""Rate of the per-read reference surface (nanonet_rgrgr_r94_posterior: one read per call, a host matrix per call) driven from
T host threads, as the reference's OpenMP loop drives it (scrappie_raw.c:355,387).  SCRAPPIE_HIP_COALESCE=0: every call alone.
usage: python tools/per_read_rate.py [threads] [reads] [samples]"""
import ctypes as C, os, sys, tempfile, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scrappie_amd as sa
from scrappie_amd import model, synth
T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
N = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
MODEL = os.environ.get("PER_READ_MODEL", "rgrgr_r94")
w = model.synthetic_model(MODEL, seed=1)
path = os.path.join(tempfile.mkdtemp(), MODEL + ".scrm")
model.save_model(w, path)
sa.register_model(MODEL, path)
sigs = [sa.RawTable(synth.medmad_normalise(synth.synthetic_signal(N, 100 + i))) for i in range(64)]
fn = getattr(sa.lib(), sa._model_fn_[MODEL])
free = sa.lib().free_scrappie_matrix
WHOLE = os.environ.get("PER_READ_DECODE", "0") != "0"      # the loop body of scrappie_raw.c:265-315: posterior, decode_transducer, overlapper
def one(i):
    if WHOLE:
        post = sa.calc_post(sigs[i % 64], MODEL, min_prob=1e-5, log=True)
        return sa._decode_post_crf(post) if MODEL == "rnnrf_r94" else sa._decode_post(post, local_pen=150.0)
    m = fn(sigs[i % 64].data(), 1e-5, 1.0, 1.0, True)
    assert m, sa.last_error()
    free(m)
for i in range(4): one(i)       # arena, kernels
t0 = time.time()
with ThreadPoolExecutor(T) as pool:
    list(pool.map(one, range(R)))
dt = time.time() - t0
st = (C.c_ulonglong * 3)()
sa.lib().scrappie_hip_coalescer_stats.argtypes = [C.POINTER(C.c_ulonglong)]
sa.lib().scrappie_hip_coalescer_stats(st)
if WHOLE:
    sa.lib().scrappie_hip_decode_coalescer_stats.argtypes = [C.POINTER(C.c_ulonglong)]
    st2 = (C.c_ulonglong * 3)()
    sa.lib().scrappie_hip_decode_coalescer_stats(st2)
    print("posterior + decode_transducer + overlapper per read; decode coalescer: %d launches, largest %d reads" % (st2[0], st2[2]))
print(MODEL, "%d threads, %d reads x %d samples, SCRAPPIE_HIP_COALESCE=%s: %.2f s -> %.0f reads/s, %.3g samples/s (posterior to host memory: %.2f GB/s); coalescer: %d launch groups, largest %d reads"
      % (T, R, N, os.environ.get("SCRAPPIE_HIP_COALESCE", "1"), dt, R / dt, R * N / dt, R * (N // 5) * 1028 * 4 / dt / 1e9, st[0], st[2]))
