import sys, numpy as np
sys.path.insert(0, "/root/repo")
import scrappie_amd as sa
from scrappie_amd import model, synth
import bench
for realistic in (0, 1):
    w = model.synthetic_model("rgrgr_r94", seed=1)
    if realistic:
        w["ff_W"], w["ff_b"] = synth.hmm_output_layer()
    eng = sa.Engine(0); eng.load_model("m", w)
    n, ns = 10000, 4000
    flat, _ = bench.make_reads(0, n, ns, seed=1)
    d = eng.upload(flat)
    off = np.arange(n, dtype=np.uint64) * np.uint64(ns); ln = np.full(n, ns, np.uint32)
    if realistic:
        eng.set_trunk_input([synth.hmm_trunk(800, 900 + i, plant_homopolymers=4)[0] for i in range(32)])
    eng.set_profiling(True)
    for rep in range(3):
        eng.run_device(d, off, ln, "m"); eng.collect(n, raw=True)
        t = eng.timing()
    print("realistic", realistic, {k: round(v, 3) for k, v in t.items() if k.endswith("_ms")})
    eng.close()
