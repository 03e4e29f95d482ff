import numpy as np, sys
sys.path.insert(0, '.')
import scrappie_amd as sa
from scrappie_amd import model, synth
eng = sa.Engine(0)
w = model.synthetic_model("nanonet_events", seed=17, size=32)
eng.load_model("events32", w)
base = [sa.event_features(synth.synthetic_events(60 + 3 * (i % 29), 300 + i)).ravel() for i in range(61)]
key = lambda c: None if c is None else (c["bases"], c["score"], c["nblock"])
ref = [key(c) for c in eng.basecall(base, "events32")]
for n in (1000, 4000, 4200, 9000):
    reads = [base[(i * 7) % 61] for i in range(n)]
    whole = [key(c) for c in eng.basecall(reads, "events32")]
    bad = [i for i in range(n) if whole[i] != ref[(i * 7) % 61]]
    print(n, "mismatches", len(bad), bad[:10])
    if bad:
        i = bad[0]; print(whole[i][1:], ref[(i*7)%61][1:], whole[i][0][:40], ref[(i*7)%61][0][:40])
