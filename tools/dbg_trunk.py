import sys; sys.path.insert(0,'.')
import numpy as np, scrappie_amd as sa, oracle
from scrappie_amd import model, synth
x = synth.medmad_normalise(synth.synthetic_signal(1500, 21))
for variant in ("zero_bias", "zero_iW", "zero_sW", "plain"):
    w = model.synthetic_model("rgrgr_r94", seed=11)
    if variant == "zero_bias": w["gru0_b"][:] = 0
    if variant == "zero_iW": w["gru0_iW"][:] = 0
    if variant == "zero_sW": w["gru0_sW"][:] = 0; w["gru0_sW2"][:] = 0
    e = sa.Engine(0); e.load_model("rgrgr_r94", w); om = oracle.OracleModel(w)
    got = e.trunk(x, "rgrgr_r94", 1); want = oracle.trunk(om, x, 1)
    d = np.abs(got-want); i = d.argmax()
    print(variant, "max err", d.max(), "got", got.ravel()[i], "want", want.ravel()[i], "mean abs got", np.abs(got).mean(), "want", np.abs(want).mean())
    e.close()
