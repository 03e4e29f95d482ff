#!/usr/bin/env python3
"""debug helper: where do the two one-kernel decoder forms (k_ff_viterbi_teams / k_ff_viterbi) differ?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scrappie_amd as sa
from scrappie_amd import model, synth

w = model.synthetic_model("rgrgr_r94", seed=1)
eng = sa.Engine(0)
eng.load_model("rgrgr_r94", w)
nreads = int(sys.argv[1]) if len(sys.argv) > 1 else 16
kw = dict(local_pen=float(sys.argv[2])) if len(sys.argv) > 2 else dict()
reads = [synth.medmad_normalise(synth.synthetic_signal(400 + 37 * (i % 29), 9000 + i)) for i in range(nreads)]
p = eng.default_params(**kw)
ln = np.array([len(x) for x in reads], np.uint32)
off = np.concatenate([[0], np.cumsum(ln[:-1], dtype=np.uint64)]).astype(np.uint64)
d = eng.upload(np.concatenate(reads))
st = []
eng.debug_option("dump_final", 1)
for single in (0, 1):
    eng.debug_option("fv_single", single)
    eng.run_device(d, off, ln, "rgrgr_r94", p)
    calls = eng.collect(len(reads), p)
    st.append({k: eng.debug_fetch(k, dt) for k, dt in (("tb", np.uint8), ("tb_end", np.int32), ("final_state", np.int32), ("final_score", np.uint32),
                                                     ("final_scores", np.uint32), ("order", np.int32), ("tile_boff", np.int64))})
a, b = st
ncb = len(a["tb_end"]) // 16
ta, tb = a["tb"].reshape(ncb, 256, 16, 4), b["tb"].reshape(ncb, 256, 16, 4)
df = np.argwhere(ta != tb)
print("ncb", ncb, "differing bytes", len(df), "of", ta.size)
if len(df):
    print("first blocks with differences:", np.unique(df[:, 0])[:20])
    print("quads:", np.unique(df[:, 1])[:64])
    print("reads:", np.unique(df[:, 2]))
    for x in df[:12]:
        print(x, "teams", ta[tuple(x)], "single", tb[tuple(x)])
print("tb_end differ:", int(np.sum(a["tb_end"] != b["tb_end"])))
fa, fb = a["final_scores"], b["final_scores"]
dd = np.argwhere(fa != fb)
print("final scores differ:", len(dd), dd[:10].ravel())
