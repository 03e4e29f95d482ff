"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through
the C ABI (ctypes over libscrappie_hip.so), against the CPU oracle and the
compiled-reference golden fixtures.

Bars (north_star / SURVEY.md section 8d):
  * integer decode path on an identical posterior: bit-exact path, bases, pos, score;
  * posterior: max |dp| <= 1e-5, max |dlogp| <= 1e-4 where p > 1e-4;
  * CRF transitions: max |d| <= 1.2e-5 = 2 x the 6.2e-6 SURVEY 8d measured between two CPU BLAS builds of
    the reference itself (measured here, HIP vs oracle: printed by test_rnnrf_transitions; HIP vs the
    independent float64 statement: 2.9e-6, tests/test_net_f64.py).
"""
import ctypes as C
import os

import numpy as np
import pytest

import scrappie_amd as sa
from scrappie_amd import model, synth

pytestmark = pytest.mark.gpu
ip = C.POINTER(C.c_int)

P_TOL, LOGP_TOL, ACT_TOL, CRF_TOL = 1e-5, 1e-4, 2e-5, 1.2e-5


@pytest.fixture(scope="module")
def eng():
    e = sa.Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def models(eng, orc, tmp_path_factory):
    """name -> (weights, OracleModel); also registered for the per-read surface."""
    out = {}
    d = tmp_path_factory.mktemp("models")
    for name, size in (("rgrgr_r94", 96), ("rgrgr_r10", 96), ("rnnrf_r94", 96)):
        w = model.synthetic_model(name, seed=11, size=size)
        eng.load_model(name, w)
        path = str(d / (name + ".scrm"))
        model.save_model(w, path)
        sa.register_model(name, path)
        out[name] = (w, orc.OracleModel(w))
    w = model.synthetic_model("raw_r94", seed=13, size=32)                  # bi-GRU (N3)
    eng.load_model("raw_r94", w)
    path = str(d / "raw_r94.scrm")
    model.save_model(w, path)
    sa.register_model("raw_r94", path)
    out["raw_r94"] = (w, orc.OracleModel(w))
    w = model.synthetic_model("rgrgr_r94", seed=12, size=32, nstate=65)     # small: 3-mers
    eng.load_model("small", w)
    out["small"] = (w, orc.OracleModel(w))
    for name, size in (("nanonet_events", 96), ("events32", 32)):           # events bi-LSTM (8(f).4)
        w = model.synthetic_model("nanonet_events", seed=17, size=size)
        eng.load_model(name, w)
        path = str(d / (name + ".scrm"))
        model.save_model(w, path)
        sa.register_model(name, path)
        out[name] = (w, orc.OracleModel(w))
    return out


def sig(n, seed):
    return synth.medmad_normalise(synth.synthetic_signal(n, seed))


# ------------------------------------------------------------------ network
@pytest.mark.parametrize("upto", [0, 1, 2, 5])
def test_trunk_layer_by_layer(eng, orc, models, upto):
    w, om = models["rgrgr_r94"]
    x = sig(1500, 21)
    got = eng.trunk(x, "rgrgr_r94", upto)
    want = orc.trunk(om, x, upto)
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) <= ACT_TOL


@pytest.mark.parametrize("name,N", [("rgrgr_r94", 2000), ("rgrgr_r94", 2003), ("rgrgr_r10", 2000), ("rgrgr_r10", 1998)])
def test_transducer_posterior(eng, orc, models, name, N):
    w, om = models[name]
    x = sig(N, 31 + N % 7)
    got = eng.posterior(x, name, min_prob=1e-5)
    want = orc.posterior(om, x, min_prob=1e-5)
    assert got.shape == want.shape == ((N + 4) // 5, 1025)
    p_got, p_want = np.exp(got.astype(np.float64)), np.exp(want.astype(np.float64))
    assert np.max(np.abs(p_got - p_want)) <= P_TOL
    big = p_want > 1e-4
    assert np.max(np.abs(got[big] - want[big])) <= LOGP_TOL
    # probabilities (return_log = False) and temperatures
    L = sa.lib()
    rt = sa.RawTable(x)
    m = L.scrappie_hip_posterior(eng._h, eng._models[name], rt.data(), 1e-5, 1.5, 0.75, False)
    gp = sa.ScrappyMatrix(m).data(as_numpy=True, sloika=False)
    wp = orc.posterior(om, x, min_prob=1e-5, tempW=1.5, tempb=0.75, log=False)
    assert np.max(np.abs(gp - wp)) <= P_TOL and abs(gp.sum(axis=1) - 1).max() < 1e-4


def test_raw_r94_bigru(eng, orc, models):
    """N3 (networks.c:196-247): conv/tanh -> {GRU fwd, GRU bwd -> feedforward2_tanh} x 2 -> softmax"""
    w, om = models["raw_r94"]
    x = sig(1500, 51)
    for upto in (0, 1, 2):
        got, want = eng.trunk(x, "raw_r94", upto), orc.trunk(om, x, upto)
        assert got.shape == want.shape and np.max(np.abs(got - want)) <= ACT_TOL, upto
    got, want = eng.posterior(x, "raw_r94"), orc.posterior(om, x)
    assert np.max(np.abs(np.exp(got.astype(np.float64)) - np.exp(want.astype(np.float64)))) <= P_TOL
    calls = eng.basecall([x, sig(903, 52)], "raw_r94", eng.default_params(want_pos=1))
    for xx, c in zip((x, sig(903, 52)), calls):
        post = eng.posterior(xx, "raw_r94")
        wsc, wseq = orc.decode_transducer(post)
        rc, wseq = orc.homopolymer_path(post, wseq)
        wb, wpos = orc.overlapper(wseq, 1024)
        assert (c["bases"] if c else None) == wb
    # per-read reference surface: get_posterior_function(SCRAPPIE_MODEL_RAW)
    pm = sa.calc_post(sa.RawTable(x), "raw_r94", min_prob=1e-5)
    assert np.array_equal(pm.data(as_numpy=True, sloika=False), got)


def test_events_bilstm(eng, orc, models):
    """SURVEY 8(f).4 (networks.c:146-193): windowed event features -> {LSTM fwd, LSTM bwd ->
    feedforward2_tanh} x 2 -> softmax; peephole LSTM cell layers.c:806-830."""
    for name in ("nanonet_events", "events32"):
        w, om = models[name]
        for n, seed in ((700, 61), (257, 62), (2, 63)):
            f3 = sa.event_features(synth.synthetic_events(n, seed))
            for upto in (1, 2):
                got, want = eng.trunk(f3.ravel(), name, upto), orc.events_trunk(om, f3, upto)
                assert got.shape == want.shape and np.max(np.abs(got - want)) <= ACT_TOL, (name, n, upto)
            got, want = eng.posterior(f3.ravel(), name), orc.events_posterior(om, f3)
            assert got.shape == want.shape == (n, 1025)
            assert np.max(np.abs(np.exp(got.astype(np.float64)) - np.exp(want.astype(np.float64)))) <= P_TOL
    # batched basecall of ragged event reads == decoding the engine's own posterior with the oracle
    w, om = models["nanonet_events"]
    reads = [sa.event_features(synth.synthetic_events(n, 70 + i)) for i, n in enumerate((900, 333, 1, 64, 1200, 17))]
    calls = eng.basecall([r.ravel() for r in reads], "nanonet_events", eng.default_params(want_pos=1))
    assert calls[2] is None                                   # one event: below the two-column minimum
    for r, c in zip(reads, calls):
        if len(r) < 2:
            continue
        post = eng.posterior(r.ravel(), "nanonet_events")
        wsc, wseq = orc.decode_transducer(post)
        rc, wseq = orc.homopolymer_path(post, wseq)
        wb, wpos = orc.overlapper(wseq, 1024)
        assert c["bases"] == wb and c["nblock"] == len(r)
    # the reference's entry point: nanonet_posterior(event_table, ...) on the registered model
    import ctypes as C
    L = sa.lib()
    ev = np.ascontiguousarray(synth.synthetic_events(300, 81))
    et = sa._EventTable(len(ev), 0, len(ev), C.cast(ev.ctypes.data, C.POINTER(sa._Event)))
    L.nanonet_posterior.restype = C.POINTER(sa._Mat)
    L.nanonet_posterior.argtypes = [sa._EventTable, C.c_float, C.c_float, C.c_float, C.c_bool]
    pm = L.nanonet_posterior(et, 1e-5, 1.0, 1.0, True)
    assert pm
    got = sa.ScrappyMatrix(pm).data(as_numpy=True, sloika=False)
    want = eng.posterior(sa.event_features(ev).ravel(), "nanonet_events")
    assert np.array_equal(got, want)


def test_events_lane_handover(eng, models):
    """More event reads than lanes (one LSTM lane per workgroup: > 256 tiles): tiles are cut
    between lanes and h / cell state handed over through HBM; calls must not change."""
    n = 4200
    base = [sa.event_features(synth.synthetic_events(60 + 3 * (i % 29), 300 + i)).ravel() for i in range(61)]
    reads = [base[(i * 7) % 61] for i in range(n)]
    key = lambda c: None if c is None else (c["bases"], c["score"], c["nblock"])
    whole = [key(c) for c in eng.basecall(reads, "events32")]
    ref = [key(c) for c in eng.basecall(base, "events32")]
    assert all(whole[i] == ref[(i * 7) % 61] for i in range(n))


def test_events_one_kernel_layer_vs_two_kernel_form(eng, models, tmp_path):
    """k_lstm_proj (projection + LSTM in one kernel, gate inputs in LDS) against k_affine + k_lstm_lanes
    (SH_GRU_SEPARATE=1, a second process): same posterior to a tenth of the tolerance -- the first level's
    projection is an exact-fp32 product in one form and a split product in the other, so not the same bits."""
    import subprocess
    import sys
    w, _ = models["nanonet_events"]
    mpath = str(tmp_path / "ev.scrm")
    model.save_model(w, mpath)
    ev = synth.synthetic_events(700, 4242)
    f3 = sa.event_features(ev)
    np.save(str(tmp_path / "f3.npy"), f3)
    code = """
import sys, numpy as np
sys.path.insert(0, %r)
import scrappie_amd as sa
e = sa.Engine(0); e.load_model("nanonet_events", %r)
f3 = np.load(%r)
np.save(%r, e.posterior(f3.ravel(), "nanonet_events", min_prob=1e-5))
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), mpath, str(tmp_path / "f3.npy"), str(tmp_path / "post2.npy"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SH_GRU_SEPARATE="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    two = np.load(str(tmp_path / "post2.npy"))
    one = eng.posterior(f3.ravel(), "nanonet_events", min_prob=1e-5)
    assert one.shape == two.shape
    assert np.max(np.abs(np.exp(one.astype(np.float64)) - np.exp(two.astype(np.float64)))) <= P_TOL / 10


def test_conv_right_edge_quirk_all_residues(eng, orc, models):
    """quirk Q1: every residue of N mod (stride * ceil(WL/stride)), both window lengths"""
    for name, span in (("rgrgr_r94", 15), ("rgrgr_r10", 20)):
        w, om = models[name]
        for N in range(600, 600 + span):
            x = sig(N, N)
            got, want = eng.trunk(x, name, 0), orc.trunk(om, x, 0)
            assert got.shape == want.shape, (name, N)
            assert np.max(np.abs(got - want)) <= ACT_TOL, (name, N)


def test_conv_matrix_pipe_form_against_valu_form(eng, models):
    """k_conv_mfma (the convolution as exact-fp32 MFMAs, the form every shipped shape runs) against k_conv_act (VALU
    multiplications and additions, SH_CONV_VALU=1 in a second process): the same windows -- Q1's right edge on every
    residue, both window lengths, the left edge, short and long reads -- the same activation, values within
    rounding of each other (the two forms associate the taps differently), and the same reads flagged for leaving the
    operand range."""
    import json
    import subprocess
    import sys
    code = """
import sys, json, numpy as np
sys.path.insert(0, %r)
import scrappie_amd as sa
from scrappie_amd import synth, model
e = sa.Engine(0)
out = {}
for name in ("rgrgr_r94", "rgrgr_r10"):
    e.load_model(name, model.synthetic_model(name, seed=1))
    for N in list(range(700, 720)) + [80, 97, 4001]:
        x = synth.medmad_normalise(synth.synthetic_signal(N, 3000 + N))
        out["%%s %%d" %% (name, N)] = e.trunk(x, name, 0).astype(np.float64).ravel().tolist()
    bad = synth.medmad_normalise(synth.synthetic_signal(900, 5)); bad[450] = np.inf
    calls = e.basecall([bad, synth.medmad_normalise(synth.synthetic_signal(900, 6))], name)
    out["%%s flagged" %% name] = [c is None for c in calls]
print(json.dumps(out))
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),)
    got = []
    for extra in ({}, {"SH_CONV_VALU": "1"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        got.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert got[0].keys() == got[1].keys()
    worst = 0.0
    for k in got[0]:
        if k.endswith("flagged"):
            assert got[0][k] == got[1][k] == [True, False], k
            continue
        a, b = np.array(got[0][k]), np.array(got[1][k])
        assert a.shape == b.shape and a.size > 0, k
        worst = max(worst, float(np.max(np.abs(a - b))))
    assert 0.0 < worst <= 2e-6, worst          # two roundings of the same sum; not the same instructions


def test_convolution_inside_first_layer_equals_separate_kernel(eng, models):
    """k_gru_conv (SH_CONV_IN_LAYER=1, a second process: the first recurrent layer of the rgrgr models computing the
    convolution itself, chunk by chunk, from the raw signal; measured 1 ms per step slower than the convolution as a kernel
    of its own, profiles/r3_conv_in_layer_stamps.txt) performs k_conv_mfma's operations in k_conv_mfma's order: calls
    identical bit for bit -- every residue of N (Q1's right edge), both window lengths, one and two tiles per workgroup,
    tiles cut between lanes, a read out of the operand range."""
    import json
    import subprocess
    import sys
    code = """
import sys, json, hashlib, numpy as np
sys.path.insert(0, %r)
import scrappie_amd as sa
from scrappie_amd import synth, model
e = sa.Engine(0)
out = {}
key = lambda c: None if c is None else (c["bases"], np.float32(c["score"]).tobytes().hex(), c["nblock"])
for name in ("rgrgr_r94", "rgrgr_r10"):
    e.load_model(name, model.synthetic_model(name, seed=1))
    p = e.default_params(local_pen=120.0)
    edge = [synth.medmad_normalise(synth.synthetic_signal(N, 3000 + N)) for N in list(range(700, 740)) + [97, 113, 4001, 30011]]
    out[name + " edges"] = [key(c) for c in e.basecall(edge, name, p)]
    base = [synth.medmad_normalise(synth.synthetic_signal(300 + 7 * (i %% 41), 9000 + i)) for i in range(97)]
    h = hashlib.sha256()
    for c in e.basecall([base[(i * 13) %% 97] for i in range(9100)], name, p):         # two tiles per workgroup, cut tiles
        h.update(repr(key(c)).encode())
    out[name + " many"] = h.hexdigest()
    bad = synth.medmad_normalise(synth.synthetic_signal(900, 5)); bad[450] = np.inf
    out[name + " flagged"] = [c is None for c in e.basecall([bad, edge[3]], name)]
print(json.dumps(out))
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),)
    got = []
    for extra in ({}, {"SH_CONV_IN_LAYER": "1", "SCRAPPIE_HIP_LIB": sa.EXP_LIB_PATH}):      # the product library; the experiments build with the switch
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        got.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert got[0] == got[1]
    assert got[0]["rgrgr_r94 flagged"] == [True, False] and sum(c is not None and len(c[0]) > 0 for c in got[0]["rgrgr_r94 edges"]) > 30


def test_rnnrf_transitions(eng, orc, models):
    w, om = models["rnnrf_r94"]
    for N in (2000, 1333):
        x = sig(N, 41)
        got = eng.posterior(x, "rnnrf_r94")
        want = orc.posterior(om, x)
        assert got.shape == want.shape == ((N + 4) // 5, 25)
        print("rnnrf N=%d: max |HIP - oracle| = %.3g" % (N, np.max(np.abs(got - want))))
        assert np.max(np.abs(got - want)) <= CRF_TOL


def test_crf_kernel_short_and_ragged_tiles(eng, orc, models):
    """k_crf keeps a ring of 8 columns and groups of 16 traceback words in flight and runs its block loops to the TILE's block count
    (sh_crf.h): reads of 8 .. 40 blocks -- fewer blocks than the ring / one walk-back group, every residue of the block count modulo
    both, tiles whose reads end at different blocks -- through the basecall path (k_crf<false>: transitions not written back) and the
    posterior surface (k_crf<true>), against the oracle's globalnorm + decode_crf on the same transitions."""
    w, om = models["rnnrf_r94"]
    min_n = eng.min_samples("rnnrf_r94")
    lens = [min_n + 5 * k + (k % 3) for k in range(33)] + [min_n, min_n, 199, 87, 123, 41, 160, 45]
    sigs = [sig(int(n), 9100 + i) for i, n in enumerate(lens)]
    calls = eng.basecall(sigs, "rnnrf_r94", eng.default_params(want_pos=1))
    worst = 0.0
    for x, c in zip(sigs, calls):
        post = eng.posterior(x, "rnnrf_r94")
        want = orc.posterior(om, x)
        assert post.shape == want.shape == ((len(x) + 4) // 5, 25)
        worst = max(worst, float(np.max(np.abs(post - want))))
        wsc, path = orc.decode_crf(post)
        assert c is not None and c["nblock"] == post.shape[0]
        assert c["bases"] == orc.crfpath_to_basecall(path, post.shape[0]), len(x)
        assert abs(c["score"] - wsc) <= 2e-3 * max(1.0, abs(wsc))
    assert worst <= CRF_TOL, worst
    # the same reads in another company (other tiles, other tile block counts): the same calls
    order = np.random.default_rng(5).permutation(len(sigs))
    again = eng.basecall([sigs[j] for j in order] + [sig(4000, 9200)], "rnnrf_r94", eng.default_params(want_pos=1))
    for k, j in enumerate(order):
        assert (again[k]["bases"], again[k]["score"]) == (calls[j]["bases"], calls[j]["score"])


@pytest.mark.parametrize("size", [32, 64, 96])
def test_rnnrf_residual_layers_other_widths_one_and_two_tiles(eng, orc, size):
    """The residual layers (networks.c:583) take their input column from an LDS ring the projection team fills (sh_gru.h, RLDS): ring slots,
    unit tiles and tile slots at layer widths 32 / 64 / 96 (two, four, six unit tiles), alone (one tile per workgroup) and in a launch group
    of 4200 reads (263 tiles: two tiles per workgroup, cut tiles) -- posterior against the oracle, calls identical alone and in company."""
    name = "rnnrf_w%d" % size
    w = model.synthetic_model("rnnrf_r94", seed=23 + size, size=size)
    eng.load_model(name, w)
    om = orc.OracleModel(w)
    min_n = eng.min_samples(name)
    base = [sig(min_n + 37 * k + (k % 5), 7300 + k) for k in range(23)] + [sig(1500, 7400), sig(803, 7401)]
    worst = 0.0
    for x in base[::4] + base[-2:]:
        got = eng.posterior(x, name)
        want = orc.posterior(om, x)
        assert got.shape == want.shape
        worst = max(worst, float(np.max(np.abs(got - want))))
    print("rnnrf width %d: max |HIP - oracle| = %.3g" % (size, worst))
    assert worst <= CRF_TOL, worst
    p = eng.default_params(want_pos=1)
    alone = [eng.basecall([x], name, p)[0] for x in base]
    many = eng.basecall([base[(i * 7) % len(base)] for i in range(4200)], name, p)
    for i, c in enumerate(many):
        a = alone[(i * 7) % len(base)]
        assert c is not None and (c["bases"], c["score"], c["nblock"]) == (a["bases"], a["score"], a["nblock"]), i


# ------------------------------------------------------------------ decode (integer path)
def test_decode_transducer_bit_exact_vs_reference_fixture(golden):
    """GPU Viterbi on the SAME posterior as the compiled reference decode.c:
    path and score must be identical (incl. slip, penalties, 3/4/5-mers)."""
    g = golden["ref_decode"]
    for (T, seed, klen, stay, skip, local, slip, hp) in g["transducer_cases"]:
        T, seed, klen, hp = int(T), int(seed), int(klen), int(hp)
        post = synth.fixture_posterior(T, seed, klen, hp)
        pm = sa.ScrappyMatrix.from_numpy(post, sloika=False)
        seq, score, pos = None, None, None
        bases, score, pos = sa._decode_post(pm, stay, skip, local, bool(slip))
        path = np.zeros(T + 1, np.int32)
        sc = sa.lib().decode_transducer(pm.data(), stay, skip, local, path.ctypes.data_as(ip), bool(slip))
        assert np.array_equal(path, g["seq_%d" % seed]), seed
        assert np.float32(sc) == g["score_%d" % seed], seed
        assert (bases or "") == str(g["bases_%d" % seed])
        assert np.array_equal(pos, g["pos_%d" % seed])
        if not slip:
            # the reference's own cross-check (src/test/test_scrappie_decoding.c:33-52): decode_transducer
            # must equal sloika_viterbi -- here k_viterbi's output against the compiled sloika_viterbi's
            # (path exact over nblock entries, score to 1e-5)
            # (wherever the compiled reference's two functions agree themselves: in case 116 they break one exact
            # tie differently; k_viterbi follows decode_transducer, asserted above)
            if np.array_equal(g["seq_%d" % seed][:T], g["sloika_seq_%d" % seed][:T]):
                assert np.array_equal(path[:T], g["sloika_seq_%d" % seed][:T]), seed
            assert abs(float(sc) - float(g["sloika_score_%d" % seed])) <= 1e-5 * max(1.0, abs(float(sc))), seed


def test_decode_transducer_ties_vs_oracle(orc):
    """floored posteriors give many exactly tied scores: tie-break order (quirk Q7)"""
    for i in range(6):
        T = 60 + 37 * i
        post, _ = synth.simulated_posterior(T, 500 + i, plant_homopolymers=2)
        post = np.maximum(post, np.float32(np.log(np.float32(4e-5)))).astype(np.float32)
        pm = sa.ScrappyMatrix.from_numpy(post, sloika=False)
        for slip in (False, True):
            path = np.zeros(T + 1, np.int32)
            sc = sa.lib().decode_transducer(pm.data(), 0.0, 0.5 * (i % 2), 2.0, path.ctypes.data_as(ip), slip)
            wsc, wseq = orc.decode_transducer(post, 0.0, 0.5 * (i % 2), 2.0, slip)
            assert np.array_equal(path, wseq) and np.float32(sc) == np.float32(wsc)


def test_decode_crf_bit_exact_vs_reference_fixture(golden):
    g = golden["ref_decode"]
    for T, seed in g["crf_cases"]:
        T, seed = int(T), int(seed)
        tr = synth.simulated_crf_transitions(T, seed)
        pm = sa.ScrappyMatrix.from_numpy(tr, sloika=False)
        bases, score, pos = sa._decode_post_crf(pm)
        path = np.zeros(T + 1, np.int32)
        sc = sa.lib().decode_crf(pm.data(), path.ctypes.data_as(ip))
        assert np.array_equal(path, g["crf_path_%d" % seed])
        assert np.float32(sc) == g["crf_score_%d" % seed]
        assert bases == str(g["crf_bases_%d" % seed])


# ------------------------------------------------------------------ whole path, batched
def _oracle_call(orc, om, x, **kw):
    p = orc.lib().orc_default_params()
    p.do_trim = 0
    for k, v in kw.items():
        setattr(p, k, v)
    return orc.basecall_raw(om, x, p)


# Decode parameter sets with an EXPENSIVE start state (local_pen >= 100): synthetic transducer weights give flat
# posteriors, and with the default local_pen = 2 the best path of a read sits in the start state (~5 bases per
# read) -- no step / skip / slip move, no one-addition fast path, no piece hand-over of k-mer scores ever decides
# a call.  With these the path must run through the k-mer states (decode.c:326-349): hundreds of bases per read.
REAL_PATH_KW = [dict(local_pen=120.0), dict(local_pen=250.0, use_slip=1, skip_pen=0.3, stay_pen=0.1),
                dict(local_pen=100.0, skip_pen=0.5), dict(local_pen=1000.0, use_slip=1)]


@pytest.mark.parametrize("kw", [dict(), dict(homopolymer=0)] + REAL_PATH_KW)
def test_batch_integer_path_exact_given_gpu_posterior(eng, orc, models, kw):
    """Engine.basecall (device Viterbi + host homopolymer/stitching) must equal
    the oracle's decode applied to the engine's own posterior, bit for bit."""
    w, om = models["rgrgr_r94"]
    lens = [1500, 1203, 777, 2000, 1500, 901, 350, 1777, 1500, 640, 1234, 1999, 1500, 1001, 1502, 888, 1640, 455]
    sigs = [sig(n, 100 + i) for i, n in enumerate(lens)]
    params = eng.default_params(want_pos=1, **kw)
    calls = eng.basecall(sigs, "rgrgr_r94", params)
    nb = nt = 0
    for x, c in zip(sigs, calls):
        post = eng.posterior(x, "rgrgr_r94", min_prob=1e-5)
        wsc, wseq = orc.decode_transducer(post, params.stay_pen, params.skip_pen, params.local_pen, bool(params.use_slip))
        if params.homopolymer:
            rc, wseq = orc.homopolymer_path(post, wseq)
        wb, wpos = orc.overlapper(wseq, 1024)
        if wb is None:
            assert c is None
            continue
        assert c["bases"] == wb
        assert np.array_equal(c["pos"], wpos)
        assert np.float32(c["score"]) == np.float32(wsc)
        assert c["nblock"] == post.shape[0]
        nb += len(wb); nt += post.shape[0]
    if params.local_pen >= 100:
        assert nb > 0.3 * nt, (nb, nt)              # the path really runs through the k-mer states


def test_decoder_input_hook_hmm_posteriors(eng, orc, models):
    """scrappie_hip_set_decoder_input (the measurement hook behind bench.py's kbases_per_s_hmm_posteriors):
    with HMM-simulated posteriors in place of the S1 output, the batched path (device Viterbi on pieces,
    homopolymer rows, host stitching) must give, bit for bit, the oracle's decode of the posterior the
    engine reports for the same read -- and those calls are real ones (~0.5 bases per block)."""
    T = 800
    probs = [synth.simulated_posterior(T, 4000 + i, plant_homopolymers=4, log=False)[0] for i in range(5)]
    sigs = [sig(5 * T, 600 + i) for i in range(37)]                 # read i decodes probs[i % 5]
    try:
        # the posterior the engine reports for each supplied matrix (a one-read group takes matrix 0)
        posts = []
        for k in range(5):
            eng.set_decoder_input([probs[k]])
            post = eng.posterior(sigs[0], "rgrgr_r94", min_prob=1e-5)
            want = np.log(np.float32(1e-5) + np.float32(1 - 1e-5) * probs[k].astype(np.float64))
            assert np.max(np.abs(post - want)) < 5e-6       # = log(min_prob + (1 - min_prob) p) of the supplied p
            posts.append(post)
        eng.set_decoder_input(probs)
        params = eng.default_params(want_pos=1)
        calls = eng.basecall(sigs, "rgrgr_r94", params)
        nb = 0
        for i, c in enumerate(calls):
            if i >= 10:
                continue
            post = posts[i % 5]
            wsc, wseq = orc.decode_transducer(post)
            rc, wseq = orc.homopolymer_path(post, wseq)
            wb, wpos = orc.overlapper(wseq, 1024)
            assert c["bases"] == wb and np.array_equal(c["pos"], wpos) and np.float32(c["score"]) == np.float32(wsc)
            nb += len(wb)
        assert nb / 10.0 > 0.3 * T                                  # a realistic decode, not the 5-base degenerate one
        assert all(calls[i]["bases"] == calls[i % 5]["bases"] for i in range(37))
    finally:
        eng.set_decoder_input(None)
    # hook off again: the network's own posterior is back
    c0 = eng.basecall(sigs[:2], "rgrgr_r94")
    assert all(len(c["bases"]) < 0.1 * T for c in c0 if c)


@pytest.fixture(scope="module")
def hmm_model(eng, orc):
    """rgrgr_r94-shaped weights whose OUTPUT LAYER is built from state codes (synth.hmm_output_layer): fed trunk
    activations that encode a k-mer path (scrappie_hip_set_trunk_input) it yields HMM-like posteriors."""
    w = model.synthetic_model("rgrgr_r94", seed=11)
    w["ff_W"], w["ff_b"] = synth.hmm_output_layer()
    eng.load_model("hmm", w)
    return w, orc.OracleModel(w)


def _tb_state(eng):
    """every buffer the transducer decoder leaves behind, for the most recent launch group"""
    return {k: eng.debug_fetch(k, dt) for k, dt in (("tb", np.uint8), ("tb_end", np.int32), ("final_state", np.int32),
                                                   ("final_score", np.uint32), ("final_scores", np.uint32),
                                                   ("order", np.int32), ("tile_boff", np.int64))}


def _same_decoder_state(a, b, ca, cb, ln):
    """two decoder forms left identical traceback bytes, end-state pointers and final scores behind (live blocks of real reads)"""
    assert np.array_equal(a["order"], b["order"]) and np.array_equal(a["tile_boff"], b["tile_boff"])
    live = _live_mask(a, (ln + 4) // 5)
    ncb = live.shape[0]
    ta, tb = a["tb"].reshape(ncb, 256, 16, 4), b["tb"].reshape(ncb, 256, 16, 4)
    m = live[:, None, :, None]
    assert np.array_equal(ta & m, tb & m), "traceback bytes differ"
    assert np.array_equal(a["tb_end"].reshape(ncb, 16)[live], b["tb_end"].reshape(ncb, 16)[live])
    real = a["order"] >= 0
    assert np.array_equal(a["final_state"][real], b["final_state"][real])
    assert np.array_equal(a["final_score"][real], b["final_score"][real])
    ntile = len(a["order"]) // 16
    fa, fb = a["final_scores"].reshape(ntile, 1024 * 16 + 32), b["final_scores"].reshape(ntile, 1024 * 16 + 32)
    rm = np.repeat(real.reshape(ntile, 1, 16), 256, axis=1)[..., None].repeat(4, axis=3).reshape(ntile, -1)
    assert np.array_equal(fa[:, :1024 * 16][rm], fb[:, :1024 * 16][rm]), "final scores of the k-mer states differ"
    r2 = np.concatenate([real.reshape(ntile, 16)] * 2, axis=1)
    assert np.array_equal(fa[:, 1024 * 16:][r2], fb[:, 1024 * 16:][r2]), "start / end state scores differ"
    key = lambda c: None if c is None else (c["bases"], c["score"], c["nblock"])
    assert [key(c) for c in ca] == [key(c) for c in cb]


def _live_mask(st, blocks):
    """[column block][16] True where the block belongs to a read (t < its block count); `blocks` by call index"""
    order = st["order"]
    ntile = len(order) // 16
    rT = np.where(order >= 0, np.asarray(blocks)[np.maximum(order, 0)], 0).reshape(ntile, 16)
    boff = st["tile_boff"]
    ncb = len(st["tb_end"]) // 16
    live = np.zeros((ncb, 16), bool)
    for t in range(ntile):
        Tt = int(rT[t].max())
        live[boff[t]:boff[t] + Tt] = np.arange(Tt)[:, None] < rT[t][None, :]
    return live


@pytest.mark.parametrize("kw", [dict(), dict(local_pen=150.0), dict(local_pen=200.0, use_slip=1, skip_pen=0.3, stay_pen=0.2, tempW=1.2, tempb=0.9),
                                dict(local_pen=120.0, skip_pen=0.3, stay_pen=0.2, tempW=1.2, tempb=0.9), dict(local_pen=100.0, skip_pen=0.25), "hmm"])
def test_whole_traceback_fused_equals_two_kernel_form(eng, models, hmm_model, kw):
    """The three forms of S1 + decoder -- k_ff_viterbi_teams (two teams of waves, scores updated in place: the default
    without slip), k_ff_viterbi (eight do-everything waves: "fv_single", and what runs with slip) and k_ff_lds + k_viterbi
    ("ff_separate") -- on the SAME launch group of 4200 mixed-length reads (263 tiles: decoded in pieces): not just the
    winning paths but every traceback byte of every state of every block, the end-state pointers, and the final scores
    of all 1024 k-mer states + start + end of every read must be identical -- with default penalties (path in the start
    state), with an expensive start state (path through the k-mer states), with slip, with a skip penalty and
    temperatures, and on HMM-like posteriors through the trunk-input hook."""
    name = "rgrgr_r94"
    base = [sig(400 + 37 * (i % 29), 9000 + i) for i in range(61)]
    reads = [base[(i * 7) % 61] for i in range(4200)]
    if kw == "hmm":
        name, kw = "hmm", dict()
        eng.set_trunk_input([synth.hmm_trunk((len(base[j]) + 4) // 5, 800 + j, plant_homopolymers=2) [0] for j in range(61)])
        reads = [base[i % 61] for i in range(4200)]              # read i <-> trunk i % 61: same block count
    p = eng.default_params(**kw)
    ln = np.array([len(x) for x in reads], np.uint32)
    off = np.concatenate([[0], np.cumsum(ln[:-1], dtype=np.uint64)]).astype(np.uint64)
    d = eng.upload(np.concatenate(reads))
    st = []
    calls = []
    try:
        eng.debug_option("dump_final", 1)
        for sep, single in ((0, 0), (1, 0), (0, 1)):
            eng.debug_option("ff_separate", sep)
            eng.debug_option("fv_single", single)
            eng.run_device(d, off, ln, name, p)
            calls.append(eng.collect(len(reads), p))
            st.append(_tb_state(eng))
    finally:
        eng.debug_option("ff_separate", 0)
        eng.debug_option("fv_single", 0)
        eng.debug_option("dump_final", 0)
        eng.set_trunk_input(None)
        eng.free(d)
    _same_decoder_state(st[0], st[2], calls[0], calls[2], ln)
    a, b = st[0], st[1]
    assert np.array_equal(a["order"], b["order"]) and np.array_equal(a["tile_boff"], b["tile_boff"])
    live = _live_mask(a, (ln + 4) // 5)
    ncb = live.shape[0]
    assert len(a["tb"]) == ncb * 1024 * 16
    ta, tb = a["tb"].reshape(ncb, 256, 16, 4), b["tb"].reshape(ncb, 256, 16, 4)
    m = live[:, None, :, None]
    assert np.array_equal(ta & m, tb & m), "traceback bytes differ"
    assert np.array_equal(a["tb_end"].reshape(ncb, 16)[live], b["tb_end"].reshape(ncb, 16)[live])
    real = a["order"] >= 0
    assert np.array_equal(a["final_state"][real], b["final_state"][real])
    assert np.array_equal(a["final_score"][real], b["final_score"][real])
    ntile = len(a["order"]) // 16
    fa, fb = a["final_scores"].reshape(ntile, 1024 * 16 + 32), b["final_scores"].reshape(ntile, 1024 * 16 + 32)
    rm = np.repeat(real.reshape(ntile, 1, 16), 256, axis=1)[..., None].repeat(4, axis=3).reshape(ntile, -1)
    assert np.array_equal(fa[:, :1024 * 16][rm], fb[:, :1024 * 16][rm]), "final scores of the k-mer states differ"
    r2 = np.concatenate([real.reshape(ntile, 16)] * 2, axis=1)
    assert np.array_equal(fa[:, 1024 * 16:][r2], fb[:, 1024 * 16:][r2]), "start / end state scores differ"
    key = lambda c: None if c is None else (c["bases"], c["score"], c["nblock"])
    assert [key(c) for c in calls[0]] == [key(c) for c in calls[1]]
    # what the traceback holds: the move mix of the live states
    codes = ta[np.broadcast_to(m, ta.shape)]
    frac = {nm: float(np.mean((codes >= lo) & (codes < hi))) for nm, lo, hi in (("stay", 0, 1), ("step", 1, 5), ("skip", 5, 21), ("slip", 21, 85), ("start", 85, 86))}
    nb = sum(len(c["bases"]) for c in calls[0] if c) / float(sum(c["nblock"] for c in calls[0] if c))
    print("traceback of %d column blocks identical; moves %s; %.2f bases per block" % (ncb, {k: round(v, 3) for k, v in frac.items()}, nb))
    if p.local_pen >= 100 or name == "hmm":
        assert nb > 0.3


@pytest.mark.parametrize("kw", [dict(), dict(use_slip=1, skip_pen=0.3, stay_pen=0.2), dict(tempW=1.2, tempb=0.8, local_pen=4.0, homopolymer=0)])
def test_trunk_input_hook_production_decoder_on_hmm_posteriors(eng, orc, hmm_model, kw):
    """scrappie_hip_set_trunk_input: the DEFAULT path -- S1 inside the decoder (k_ff_viterbi), pieces, homopolymer
    rows, host stitching -- on posteriors of a simulated k-mer path (~0.5 bases per block, planted homopolymers).
    (1) the posterior the engine reports for the injected trunk equals the oracle's S1 + S2 of the same
    activations within tolerance (peaked posteriors: max p ~0.6, unlike the flat ones random weights give);
    (2) every call equals, bit for bit, the oracle's decode (decode.c:123, homopolymer.c:175, decode.c:449) of the
    engine's own posterior; (3) calls of duplicates agree across tiles and pieces."""
    w, om = hmm_model
    Ts = [800, 640, 333, 801, 97, 12, 500]
    trunks = [synth.hmm_trunk(T, 4100 + i, plant_homopolymers=4)[0] for i, T in enumerate(Ts)]
    n = 4300                                                       # 269 tiles: decoded in pieces
    sigs = [sig(5 * Ts[i % 7] - (i % 7 == 3) * 3, 600 + i % 7) for i in range(n)]
    p = eng.default_params(want_pos=1, **kw)
    try:
        posts = []
        for k in range(7):
            eng.set_trunk_input([trunks[k]])
            post = eng.posterior(sigs[k], "hmm", min_prob=p.min_prob, tempW=p.tempW, tempb=p.tempb)
            want = orc.softmax_posterior(om, trunks[k], min_prob=p.min_prob, tempW=p.tempW, tempb=p.tempb)
            assert post.shape == want.shape == (Ts[k], 1025)
            pg, pw = np.exp(post.astype(np.float64)), np.exp(want.astype(np.float64))
            assert np.max(np.abs(pg - pw)) <= P_TOL, k
            big = pw > 1e-4
            assert np.max(np.abs(post[big] - want[big])) <= LOGP_TOL, k
            posts.append(post)
        assert np.mean([np.exp(q).max(axis=1).mean() for q in posts]) > 0.3          # peaked
        eng.set_trunk_input(trunks)
        calls = eng.basecall(sigs, "hmm", p)
    finally:
        eng.set_trunk_input(None)
    nb = nt = 0
    for k in range(7):
        post, c = posts[k], calls[k]
        wsc, wseq = orc.decode_transducer(post, p.stay_pen, p.skip_pen, p.local_pen, bool(p.use_slip))
        if p.homopolymer:
            rc, wseq = orc.homopolymer_path(post, wseq)
        wb, wpos = orc.overlapper(wseq, 1024)
        assert c["bases"] == wb and np.array_equal(c["pos"], wpos) and np.float32(c["score"]) == np.float32(wsc), k
        nb += len(wb); nt += Ts[k]
    assert nb > 0.3 * nt
    key = lambda c: (c["bases"], np.float32(c["score"]).tobytes(), c["nblock"])
    assert all(key(calls[i]) == key(calls[i % 7]) for i in range(n))
    # hook off again: the network's own (flat) posterior is back
    c0 = eng.basecall(sigs[:2], "hmm")
    assert all(len(c["bases"]) < 0.1 * c["nblock"] for c in c0 if c)


def test_handover_timeout_reruns_group_on_whole_tiles(models, tmp_path):
    """A timed-out state hand-over (sh_wait_flag gives up after SH_HANDOVER_TIMEOUT_S) must not lose the
    launch group: scrappie_hip_collect runs it again on whole tiles.  SH_FAKE_HANDOVER_TIMEOUT makes the
    first collect of a process behave as if the error word were set."""
    import subprocess
    import sys
    w, _ = models["rgrgr_r94"]
    mpath = str(tmp_path / "m.scrm")
    model.save_model(w, mpath)
    code = """
import sys, json, numpy as np
sys.path.insert(0, %r)
import scrappie_amd as sa
from scrappie_amd import synth
e = sa.Engine(0); e.load_model("rgrgr_r94", %r)
base = [synth.medmad_normalise(synth.synthetic_signal(300 + 11 * (i %% 23), 8000 + i)) for i in range(53)]
reads = [base[(i * 5) %% 53] for i in range(8800)]
out = []
for rep in range(2):
    out.append([None if c is None else (c["bases"], c["score"], c["nblock"]) for c in e.basecall(reads, "rgrgr_r94")])
print(json.dumps(out[0] == out[1]))
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), mpath)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SH_FAKE_HANDOVER_TIMEOUT="1"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "re-run on whole tiles" in r.stderr
    assert r.stdout.strip().splitlines()[-1] == "true"      # first pass (re-run) == second pass (normal)


def test_fused_s1_decoder_equals_two_kernel_form(models, tmp_path):
    """k_ff_viterbi (S1 inside the decoder: the posterior never written) against k_ff_lds + k_viterbi
    (SH_FF_SEPARATE=1, a second process): every call identical -- bases, score bits, block count -- on 4800
    mixed-length reads, i.e. more tiles than CUs, so tiles are decoded in pieces that hand over their
    state; default and slip / temperature / penalty parameters."""
    import subprocess
    import sys
    import json
    w, _ = models["rgrgr_r94"]
    mpath = str(tmp_path / "m.scrm")
    model.save_model(w, mpath)
    code = """
import sys, json, hashlib, numpy as np
sys.path.insert(0, %r)
import scrappie_amd as sa
from scrappie_amd import synth
e = sa.Engine(0); e.load_model("rgrgr_r94", %r)
base = [synth.medmad_normalise(synth.synthetic_signal(400 + 37 * (i %% 29), 9000 + i)) for i in range(61)]
reads = [base[(i * 7) %% 61] for i in range(4800)]
out = []
for kw in (dict(), dict(tempW=1.3, tempb=0.8, use_slip=1, stay_pen=0.3, skip_pen=0.2), dict(use_slip=1, local_pen=1.0, homopolymer=0),
           dict(local_pen=150.0), dict(local_pen=300.0, use_slip=1, skip_pen=0.25, stay_pen=0.1, tempW=0.9), dict(local_pen=100.0, skip_pen=0.6, homopolymer=0)):
    h = hashlib.sha256()
    nb = nt = 0
    for c in e.basecall(reads, "rgrgr_r94", e.default_params(**kw)):
        h.update(repr(None if c is None else (c["bases"], np.float32(c["score"]).tobytes().hex(), c["nblock"])).encode())
        if c is not None:
            nb += len(c["bases"]); nt += c["nblock"]
    out.append([h.hexdigest(), nb / max(nt, 1)])
print(json.dumps(out))
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), mpath)
    got = []
    for extra in ({}, {"SH_FF_SEPARATE": "1"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        got.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert got[0] == got[1]
    assert len(set(h for h, _ in got[0])) == 6            # the parameter sets do decode differently
    # with an expensive start state the calls are real ones: the path runs through the k-mer states
    assert all(bpb > 0.3 for _, bpb in got[0][3:]), got[0]
    assert all(bpb < 0.1 for _, bpb in got[0][:1]), got[0]


@pytest.mark.parametrize("name,kw", [("rgrgr_r94", dict()), ("rnnrf_r94", dict()), ("rgrgr_r94", dict(local_pen=150.0)),
                                     ("rgrgr_r94", dict(local_pen=200.0, use_slip=1, skip_pen=0.3))])
def test_batch_end_to_end_vs_oracle(eng, orc, models, name, kw):
    """The whole path against the oracle's whole path (its own network, its own decode).  The two posteriors
    differ in the last bits (tolerance tests above), so the Viterbi PATH may legitimately differ where two paths
    tie to within that noise ('bit-identity of the path').  Required: block count equal, score
    within 1e-3 relative, and the bases IDENTICAL unless the oracle itself, decoding the ENGINE's posterior,
    reproduces the engine's call -- i.e. unless the whole difference is explained by the posterior's last bits.
    With an expensive start state the calls are hundreds of bases long (not the ~5-base degenerate ones)."""
    w, om = models[name]
    lens = [1000, 1203, 777, 1500, 640, 901, 2000, 455]
    sigs = [sig(n, 300 + i) for i, n in enumerate(lens)]
    p = eng.default_params(**kw)
    calls = eng.basecall(sigs, name, p)
    same = explained = 0
    nb = nt = 0
    for x, c in zip(sigs, calls):
        o = _oracle_call(orc, om, x, **{("homopolymer_mean" if k == "homopolymer" else k): v for k, v in kw.items()})
        if o is None:
            assert c is None
            continue
        assert c is not None and c["nblock"] == o["nblock"]
        assert abs(c["score"] - o["score"]) <= 1e-3 * max(1.0, abs(o["score"])), (c["score"], o["score"])
        nb += len(c["bases"]); nt += c["nblock"]
        if c["bases"] == o["bases"]:
            same += 1
            continue
        # not identical: the engine's call must be THE call of its own posterior (oracle decode, bit-exact) ...
        post = eng.posterior(x, name, min_prob=p.min_prob)
        if name == "rnnrf_r94":
            wsc, path = orc.decode_crf(post)
            wb = orc.crfpath_to_basecall(path, post.shape[0])
        else:
            wsc, wseq = orc.decode_transducer(post, p.stay_pen, p.skip_pen, p.local_pen, bool(p.use_slip))
            if p.homopolymer:
                rc, wseq = orc.homopolymer_path(post, wseq)
            wb, _ = orc.overlapper(wseq, 1024)
        assert c["bases"] == wb, "call differs from the oracle's and is not the decode of the engine's own posterior"
        # ... and the two posteriors must be within tolerance of each other (a near tie decided by their last bits)
        want = orc.posterior(om, x, min_prob=p.min_prob)
        d = np.max(np.abs(post - want)) if name == "rnnrf_r94" else np.max(np.abs(np.exp(post.astype(np.float64)) - np.exp(want.astype(np.float64))))
        assert d <= (CRF_TOL if name == "rnnrf_r94" else P_TOL)
        explained += 1
    print("%s %s: %d of %d calls identical to the oracle's; %d differ at near ties (each equal to the oracle's decode of the "
          "engine's own posterior); %.2f bases per block" % (name, kw, same, len(lens), explained, nb / max(nt, 1)))
    assert same + explained == len(lens)
    if kw.get("local_pen", 2.0) >= 100:
        assert nb > 0.3 * nt
    else:
        assert same >= len(lens) - 1                 # short calls on flat posteriors: at most one near tie


def test_small_kmer_model_end_to_end(eng, orc, models):
    """3-mer transducer (NS = 65) through the same kernels"""
    w, om = models["small"]
    sigs = [sig(n, 700 + i) for i, n in enumerate((800, 555, 1000))]
    calls = eng.basecall(sigs, "small", eng.default_params(want_pos=1))
    for x, c in zip(sigs, calls):
        post = eng.posterior(x, "small")
        assert np.max(np.abs(np.exp(post) - np.exp(orc.posterior(om, x)))) <= P_TOL
        wsc, wseq = orc.decode_transducer(post)
        rc, wseq = orc.homopolymer_path(post, wseq)
        wb, wpos = orc.overlapper(wseq, 64)
        assert (c["bases"] if c else None) == wb


def test_ragged_and_degenerate_inputs(eng, models):
    """empty / too-short reads give no call and do not disturb their tile-mates"""
    ms = eng.min_samples("rgrgr_r94")
    good = [sig(1200 + 37 * i, 900 + i) for i in range(5)]
    alone = eng.basecall(good, "rgrgr_r94")
    mixed = [good[0], np.zeros(0, np.float32), good[1], sig(ms - 1, 1), good[2], sig(7, 2), good[3], good[4]]
    calls = eng.basecall(mixed, "rgrgr_r94")
    assert calls[1] is None and calls[3] is None and calls[5] is None
    for j, k in enumerate((0, 2, 4, 6, 7)):
        assert calls[k]["bases"] == alone[j]["bases"] and calls[k]["score"] == alone[j]["score"]
    assert eng.basecall([np.zeros(0, np.float32)], "rgrgr_r94") == [None]


def test_two_launch_groups_in_flight(eng, models):
    """The engine holds two launch groups: group B may be enqueued before group A
    is collected (host stitching of A overlaps B's kernels).  Results must equal
    the one-at-a-time results, collect() takes the OLDEST group, and a third
    enqueue without a collect is refused."""
    A = [sig(1200 + 37 * i, 7000 + i) for i in range(40)]
    B = [sig(900 + 53 * i, 7100 + i) for i in range(24)]
    refA, refB = eng.basecall(A, "rgrgr_r94"), eng.basecall(B, "rgrgr_r94")

    def up(reads):
        ln = np.array([len(x) for x in reads], np.uint32)
        off = np.concatenate([[0], np.cumsum(ln[:-1], dtype=np.uint64)]).astype(np.uint64)
        return eng.upload(np.concatenate(reads)), off, ln
    dA, offA, lnA = up(A)
    dB, offB, lnB = up(B)
    key = lambda c: (c["bases"], c["score"], c["nblock"], tuple(c["pos"]) if "pos" in c else None)
    try:
        for _ in range(3):
            eng.run_device(dA, offA, lnA, "rgrgr_r94")
            eng.run_device(dB, offB, lnB, "rgrgr_r94")
            with pytest.raises(RuntimeError):
                eng.run_device(dA, offA, lnA, "rgrgr_r94")
            with pytest.raises(RuntimeError):
                eng.collect(len(B))              # oldest group is A: size mismatch is refused
            gotA = eng.collect(len(A))
            eng.run_device(dA, offA, lnA, "rgrgr_r94")   # slot of A is free again
            gotB = eng.collect(len(B))
            gotA2 = eng.collect(len(A))
            assert [key(c) for c in gotA] == [key(c) for c in refA]
            assert [key(c) for c in gotB] == [key(c) for c in refB]
            assert [key(c) for c in gotA2] == [key(c) for c in refA]
        with pytest.raises(RuntimeError):
            eng.collect(len(A))                  # nothing in flight
    finally:
        eng.free(dA)
        eng.free(dB)


def test_lane_schedule_handover_is_invisible(eng, models):
    """More tiles than lanes (> 512 tiles = 8192 reads on 256 CUs): the recurrent
    kernel cuts tiles between lanes and hands the state over through HBM
    (sh_sched.h).  The calls must be bit-identical to the same reads run in
    batches small enough that nothing is cut."""
    n = 9100
    base = [sig(300 + 7 * (i % 41), 9000 + i) for i in range(97)]
    reads = [base[(i * 13) % 97] for i in range(n)]
    key = lambda c: None if c is None else (c["bases"], c["score"], c["nblock"])
    whole = [key(c) for c in eng.basecall(reads, "rgrgr_r94")]
    parts = []
    for lo in range(0, n, 1300):
        parts += [key(c) for c in eng.basecall(reads[lo:lo + 1300], "rgrgr_r94")]
    assert whole == parts
    ref = {i: key(c) for i, c in enumerate(eng.basecall(base, "rgrgr_r94"))}
    assert all(whole[i] == ref[(i * 13) % 97] for i in range(n))


def test_full_size_batch_properties(eng, models):
    """BASELINE config 2 shape (4000-sample reads) at a size the oracle cannot
    check read by read: results must be (a) deterministic, (b) independent of
    batch composition / tile neighbours / launch-group size, (c) equal for
    duplicate reads.  n = 10 000 is the benchmark's launch group exactly: 625 tiles, so the
    recurrence's lane cuts, the decoder's pieces and the fused projection all run."""
    n = 10000
    base = [sig(4000, 5000 + i) for i in range(64)]
    sigs = [base[i % 64] for i in range(n)]
    flat = np.concatenate(sigs)
    off = np.arange(n, dtype=np.uint64) * 4000
    ln = np.full(n, 4000, np.uint32)
    d = eng.upload(flat)
    try:
        eng.run_device(d, off, ln, "rgrgr_r94")
        a = eng.collect(n)
        eng.run_device(d, off, ln, "rgrgr_r94")
        b = eng.collect(n)
    finally:
        eng.free(d)
    key = lambda c: (c["bases"], c["score"], c["nblock"])
    assert [key(c) for c in a] == [key(c) for c in b]
    for i in range(n):
        assert key(a[i]) == key(a[i % 64])
    assert all(c["nblock"] == 800 for c in a)
    solo = eng.basecall(base[:5], "rgrgr_r94")
    assert [key(c) for c in solo] == [key(c) for c in a[:5]]
    eng.set_max_launch_reads(48)          # forces several launch groups
    try:
        split = eng.basecall(base[:20] + base[:20] + base[:20], "rgrgr_r94")
    finally:
        eng.set_max_launch_reads(16384)
    assert [key(c) for c in split] == [key(c) for c in a[:20]] * 3


def test_config4_one_gpu_share_at_stated_size(eng, models):
    """BASELINE config 4 is 1 M reads of 4000 samples over 8 GPUs: one GPU's share, 125 000 reads (5e8 samples),
    in ONE host-to-host call (8 launch groups of <= 16384 reads, two in flight).  The oracle cannot check that
    read by read; every call must equal the call of the same signal in a 256-read batch (batch independence),
    whatever launch group and tile it landed in."""
    n, distinct = 125000, 256
    base = [sig(4000, 9000 + i) for i in range(distinct)]
    want = eng.basecall(base, "rgrgr_r94")
    key = lambda c: (c["bases"], np.float32(c["score"]).tobytes(), c["nblock"])
    wk = [key(c) for c in want]
    assert all(c["nblock"] == 800 for c in want)
    order = (np.arange(n, dtype=np.int64) * 7919) % distinct
    got = eng.basecall([base[j] for j in order], "rgrgr_r94")
    assert len(got) == n
    bad = [i for i in range(n) if key(got[i]) != wk[order[i]]]
    assert not bad, (len(bad), bad[:5])


def test_very_long_read_in_a_large_batch(eng, models):
    """A read of 70 000 blocks (350 000 samples: more than the 65535 blocks per tile the two-tiles-per-workgroup
    form of k_gru_proj can count) inside a batch with more tiles than CUs: the group must fall back to one
    tile per workgroup and give the long read the call it gets alone."""
    long_read = sig(350000, 31337)
    short = [sig(300 + 7 * (i % 23), 8100 + i) for i in range(53)]
    key = lambda c: None if c is None else (c["bases"], np.float32(c["score"]).tobytes(), c["nblock"])
    alone = eng.basecall([long_read], "rgrgr_r94")[0]
    assert alone is not None and alone["nblock"] == 70000
    batch = eng.basecall([short[(i * 5) % 53] for i in range(4300)] + [long_read], "rgrgr_r94")
    assert key(batch[-1]) == key(alone)
    ref = eng.basecall(short, "rgrgr_r94")
    assert all(key(batch[i]) == key(ref[(i * 5) % 53]) for i in range(4300))


def test_scrappy_surface_basecall_raw(eng, orc, models):
    """python/scrappy/__init__.py:403: trim -> scale -> calc_post(min_prob 1e-6) -> decode"""
    w, om = models["rgrgr_r94"]
    raw = synth.synthetic_signal(3000, 77, raw_units=True)
    seq, score, pos, start, end, bp = sa.basecall_raw(raw, "rgrgr_r94")
    p = orc.lib().orc_default_params()
    p.min_prob = 1e-6
    p.homopolymer_mean = 0
    o = orc.basecall_raw(om, raw, p)
    assert (start, end) == (o["start"], o["end"])
    assert abs(score - o["score"]) <= 1e-3 * max(1.0, abs(o["score"]))
    assert sa.get_model_stride("rgrgr_r94") == 5
    seq2, score2, pos2, s2, e2, bp2 = sa.basecall_raw(raw, "rnnrf_r94", with_base_probs=True)
    # posterior_crf's normaliser starts its log-sum at 0.0f, i.e. adds an extra
    # e^0 to every column total (decode.c:969,998; quirk Q16), so columns do not
    # sum to one; parity is against the oracle (bit-exact vs compiled reference).
    rt = sa.RawTable(raw).trim().scale()
    trans = sa.calc_post(rt, "rnnrf_r94").data(as_numpy=True, sloika=False)
    L = orc.lib()
    want = orc.mat_to_numpy(L.orc_posterior_crf(orc.NpMat(trans).ptr), L.orc_free_mat)
    assert bp2.shape == want.shape and np.array_equal(bp2.view(np.uint32), want.view(np.uint32))


def test_handover_can_be_disabled(models):
    """SCRAPPIE_HIP_HANDOVER=0: whole tiles per lane / per decoder workgroup (no inter-workgroup
    waits at all); the calls are the same as with the cut schedules."""
    w, _ = models["rgrgr_r94"]
    base = [sig(300 + 11 * (i % 23), 8000 + i) for i in range(53)]
    reads = [base[(i * 5) % 53] for i in range(8800)]
    key = lambda c: None if c is None else (c["bases"], c["score"], c["nblock"])
    res = []
    for flag in ("1", "0"):
        os.environ["SCRAPPIE_HIP_HANDOVER"] = flag
        try:
            e = sa.Engine(0)
            e.load_model("rgrgr_r94", w)
            res.append([key(c) for c in e.basecall(reads, "rgrgr_r94")])
            e.close()
        finally:
            os.environ.pop("SCRAPPIE_HIP_HANDOVER", None)
    assert res[0] == res[1]


def test_one_kernel_layer_equals_separate_kernels_bitwise(eng, models, tmp_path):
    """A recurrent layer as one kernel (k_gru_proj: projection team + recurrence team, gate inputs in LDS)
    performs the same split products in the same order as the projection kernel followed by the
    recurrence kernel: posteriors and calls of a process run with SH_GRU_SEPARATE=1 are bit-identical
    to this process's."""
    import subprocess
    import sys
    w, _ = models["rgrgr_r94"]
    mpath = str(tmp_path / "m.scrm")
    model.save_model(w, mpath)
    script = tmp_path / "separate.py"
    script.write_text(
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import scrappie_amd as sa\n"
        "from scrappie_amd import synth\n"
        "e = sa.Engine(0); e.load_model('m', %r)\n"
        "x = [synth.medmad_normalise(synth.synthetic_signal(n, 700 + i)) for i, n in enumerate((2000, 1203, 4000))]\n"
        "np.save(%r, e.posterior(x[0], 'm'))\n"
        "calls = e.basecall(x, 'm')\n"
        "open(%r, 'w').write('\\n'.join('%%s %%r' %% (c['bases'], c['score']) for c in calls))\n"
        % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), mpath, str(tmp_path / "post.npy"), str(tmp_path / "calls.txt")))
    env = dict(os.environ, SH_GRU_SEPARATE="1")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    x = [sig(n, 700 + i) for i, n in enumerate((2000, 1203, 4000))]
    post = eng.posterior(x[0], "rgrgr_r94")
    assert np.array_equal(post.view(np.uint32), np.load(str(tmp_path / "post.npy")).view(np.uint32))
    calls = eng.basecall(x, "rgrgr_r94")
    assert "\n".join("%s %r" % (c["bases"], c["score"]) for c in calls) == open(str(tmp_path / "calls.txt")).read()
    # ... and the exact-fp32 MFMA recurrence (SH_GRU_F32=1, the kernel the split products replaced) agrees
    # with the split products to well within the posterior tolerance, end to end
    env = dict(os.environ, SH_GRU_SEPARATE="1", SH_GRU_F32="1")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    exact = np.load(str(tmp_path / "post.npy"))
    assert np.max(np.abs(np.exp(post) - np.exp(exact))) <= P_TOL / 4


def test_barrier_free_layer_equals_barrier_form(eng, models, tmp_path):
    """k_gru_free (SH_GRU_FREE=1, a second process: no s_barrier in the step loop -- projection team free running on
    a ring of three blocks, all synchronisation through LDS counters; measured 4 % slower, profiles/r3_gru_free_stamps.txt)
    performs k_gru_proj's arithmetic: posterior bits and calls identical, also across lane cuts (9100 reads) and for
    the residual variant (rnnrf)."""
    import json
    import subprocess
    import sys
    code = """
import sys, json, hashlib, numpy as np
sys.path.insert(0, %r)
import scrappie_amd as sa
from scrappie_amd import synth, model
e = sa.Engine(0)
out = []
for name in ("rgrgr_r94", "rnnrf_r94"):
    e.load_model(name, model.synthetic_model(name, seed=11))
    x = synth.medmad_normalise(synth.synthetic_signal(2003, 700))
    out.append(hashlib.sha256(e.posterior(x, name).tobytes()).hexdigest())
    base = [synth.medmad_normalise(synth.synthetic_signal(300 + 7 * (i %% 41), 9000 + i)) for i in range(97)]
    h = hashlib.sha256()
    for c in e.basecall([base[(i * 13) %% 97] for i in range(9100)], name, e.default_params(local_pen=120.0)):
        h.update(repr((c["bases"], np.float32(c["score"]).tobytes().hex())).encode())
    out.append(h.hexdigest())
print(json.dumps(out))
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),)
    got = []
    for extra in ({}, {"SH_GRU_FREE": "1", "SCRAPPIE_HIP_LIB": sa.EXP_LIB_PATH}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        got.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert got[0] == got[1] and len(set(got[0])) == 4


@pytest.mark.parametrize("kw", [
    dict(tempW=1.3, tempb=0.8, use_slip=1, stay_pen=0.3, skip_pen=0.2),       # S1 with the division, slip move
    dict(tempW=0.7, tempb=1.0, use_slip=0, local_pen=1.0, homopolymer=0),     # input scaling only, no homopolymer pass
    dict(min_prob=1e-3, use_slip=1, skip_pen=1.0),
    dict(local_pen=150.0),                                                     # real paths (see REAL_PATH_KW): k-mer scores cross the
    dict(local_pen=100.0, use_slip=1, skip_pen=0.4, stay_pen=0.1),             # piece hand-over and decide the calls
])
def test_large_batch_equals_small_batches_with_options(eng, models, kw):
    """The large-batch kernels (LDS-resident S1 incl. its tempb != 1 variant, lane cuts, decoder
    pieces, slip move) against the small-batch kernels on the same reads: identical calls, and the
    decode of a read equals the oracle's decode of the engine's own posterior for these options."""
    import oracle
    p = eng.default_params(want_pos=1, **kw)
    base = [sig(300 + 11 * (i % 19), 9500 + i) for i in range(41)]
    n = 9300
    reads = [base[(i * 3) % 41] for i in range(n)]
    key = lambda c: None if c is None else (c["bases"], c["score"], c["nblock"], tuple(c["pos"]))
    small = [key(c) for c in eng.basecall(base, "rgrgr_r94", p)]
    big = [key(c) for c in eng.basecall(reads, "rgrgr_r94", p)]
    assert all(big[i] == small[(i * 3) % 41] for i in range(n))
    if p.local_pen >= 100:
        assert sum(len(c[0]) for c in small) > 0.3 * sum(c[2] for c in small)
    for x, c in list(zip(base, small))[:4]:
        post = eng.posterior(x, "rgrgr_r94", min_prob=p.min_prob, tempW=p.tempW, tempb=p.tempb)
        sc, seq = oracle.decode_transducer(post, p.stay_pen, p.skip_pen, p.local_pen, bool(p.use_slip))
        if p.homopolymer:
            rc, seq = oracle.homopolymer_path(post, seq)
        bases, pos = oracle.overlapper(seq, 1024)
        assert c[0] == bases and np.float32(c[1]) == np.float32(sc)


def test_one_or_two_tiles_per_workgroup_is_a_schedule_choice(eng, models):
    """A launch lasts its longest lane's steps, so with mixed lengths whose longest tile sets the lane capacity the
    recurrent layers run one tile per workgroup (lanes = CUs, tiles cut and handed over through HBM as in the two-tile
    form; profiles/r3_mixed_rate_*.txt), with equal reads two.  Which one runs must not show in any call: the same
    reads through both forms (debug option "gru_tiles") give identical bases, scores and positions, and the long
    reads' calls equal the oracle's decode of the engine's own posterior."""
    import oracle
    p = eng.default_params(want_pos=1, local_pen=120.0)
    long_reads = [sig(15000 + 37 * i, 7700 + i) for i in range(16)]
    base = [sig(300 + 11 * (i % 23), 7800 + i) for i in range(47)]
    reads = long_reads + [base[(i * 5) % 47] for i in range(4300)]            # 270 tiles on 256 CUs, one of 3000+ blocks
    key = lambda c: None if c is None else (c["bases"], np.float32(c["score"]).tobytes(), c["nblock"], tuple(c["pos"]))
    got = {}
    try:
        for mode in (0, 1, 2):
            eng.debug_option("gru_tiles", mode)
            got[mode] = [key(c) for c in eng.basecall(reads, "rgrgr_r94", p)]
            tiles = int(eng.debug_fetch("gru_tiles", np.int32)[0])
            assert tiles == (mode if mode else 1)                              # the longest tile sets both capacities: one
        eng.debug_option("gru_tiles", 0)
        same = [sig(500, 7900 + i) for i in range(29)]
        eng.basecall([same[i % 29] for i in range(9000)], "rgrgr_r94", p)       # equal reads, two tiles per CU and more: two
        assert int(eng.debug_fetch("gru_tiles", np.int32)[0]) == 2
    finally:
        eng.debug_option("gru_tiles", 0)
    assert got[0] == got[1] == got[2]
    small = [key(c) for c in eng.basecall(base, "rgrgr_r94", p)]
    assert all(got[0][16 + i] == small[(i * 5) % 47] for i in range(4300))
    for x, c in list(zip(long_reads, got[0]))[:2]:
        post = eng.posterior(x, "rgrgr_r94")
        sc, seq = oracle.decode_transducer(post, p.stay_pen, p.skip_pen, p.local_pen, False)
        rc, seq = oracle.homopolymer_path(post, seq)
        bases, pos = oracle.overlapper(seq, 1024)
        assert c[0] == bases and c[1] == np.float32(sc).tobytes() and len(bases) > 0.3 * c[2]


def test_mixed_lengths_many_launch_groups_in_flight(eng, models):
    """basecall_batch cuts its input into launch groups bounded in reads and in column blocks
    (scrappie_hip_plan_groups) and keeps two in flight, uploads and downloads on their own streams:
    the calls must not depend on the cut (config 3: mixed-length reads)."""
    rng = np.random.default_rng(21)
    lens = rng.integers(200, 6001, size=150)
    lens[7] = 0; lens[33] = 12                      # empty and too-short reads keep their place
    sigs = [synth.synthetic_signal(int(n), 900 + i) if n else np.zeros(0, np.float32) for i, n in enumerate(lens)]
    key = lambda c: None if c is None else (c["bases"], c["score"], c["nblock"])
    whole = eng.basecall(sigs, "rgrgr_r10")
    assert whole[7] is None and whole[33] is None and sum(c is not None for c in whole) == 148
    try:
        eng.set_max_launch_blocks(2600)              # a handful of reads per group
        cut = eng.basecall(sigs, "rgrgr_r10")
        eng.set_max_launch_blocks(0)
        eng.set_max_launch_reads(16)
        cut2 = eng.basecall(sigs, "rgrgr_r10")
    finally:
        eng.set_max_launch_blocks(0)
        eng.set_max_launch_reads(16384)
    assert [key(c) for c in cut] == [key(c) for c in whole]
    assert [key(c) for c in cut2] == [key(c) for c in whole]
    eng.set_max_launch_blocks(100)
    try:
        with pytest.raises(RuntimeError, match="too long"):
            eng.basecall(sigs[:4], "rgrgr_r10")
        eng.set_max_launch_blocks(0)
        again = eng.basecall(sigs[:4], "rgrgr_r10")  # the engine is usable after the refusal
    finally:
        eng.set_max_launch_blocks(0)
    assert [key(c) for c in again] == [key(c) for c in whole[:4]]


@pytest.mark.parametrize("name", ["rgrgr_r10", "rnnrf_r94"])
def test_config3_mixed_lengths_at_stated_size(eng, orc, models, name):
    """BASELINE config 3 at the size SURVEY 8(d) states: reads N ~ U{1000..40000} (incl. N % 5 != 0 and == 0, i.e.
    Q1 hit and miss for both window lengths, the shortest legal read and reads with T < 4) through rgrgr_r10
    (transducer decode) and rnnrf_r94 (globalnorm + decode_crf) in ONE call.  A sample of reads is checked read
    by read against the oracle (posterior within tolerance, integer path bit-exact on the engine's own
    posterior); every read is checked by batch independence: the call it gets in the big mixed batch is the call
    it gets in a different, smaller company."""
    w, om = models[name]
    rng = np.random.default_rng(33)
    n = 2000
    lens = rng.integers(1000, 40001, size=n)
    min_n = eng.min_samples(name)
    lens[:6] = [min_n, min_n + 1, 17, 1000, 40000, 39998]          # shortest legal, T < 4, both ends of the range
    long_sig = synth.medmad_normalise(synth.synthetic_signal(40000 + 64 * 13, 77))
    sigs = [long_sig[(i % 64) * 13:(i % 64) * 13 + int(L)] for i, L in enumerate(lens)]
    sigs[2] = sigs[2][:3]                                             # too short: no call, keeps its place
    key = lambda c: None if c is None else (c["bases"], c["score"], c["nblock"])
    params = eng.default_params(want_pos=1)
    big = eng.basecall(sigs, name, params)
    assert big[2] is None and all(c is not None for i, c in enumerate(big) if i != 2)
    assert [c["nblock"] for c in big[:2]] == [(min_n + 4) // 5, (min_n + 5) // 5]
    # (1) read by read against the oracle
    sample = [0, 1, 3, 4, 5] + [int(i) for i in rng.choice(np.arange(6, n), size=7, replace=False)]
    worst = 0.0
    for i in sample:
        x, c = sigs[i], big[i]
        post = eng.posterior(x, name, min_prob=1e-5)
        want = orc.posterior(om, x, min_prob=1e-5)
        assert post.shape == want.shape
        if name == "rnnrf_r94":
            d = float(np.max(np.abs(post - want)))
            assert d <= CRF_TOL, (i, len(x), d)
            wsc, path = orc.decode_crf(post)
            wb = orc.crfpath_to_basecall(path, post.shape[0])
            assert c["bases"] == wb
            assert abs(c["score"] - wsc) <= 2e-3 * max(1.0, abs(wsc))     # (k_crf subtracts logZ / T before its Viterbi; same path)
        else:
            d = float(np.max(np.abs(np.exp(post.astype(np.float64)) - np.exp(want.astype(np.float64)))))
            assert d <= P_TOL, (i, len(x), d)
            wsc, wseq = orc.decode_transducer(post)
            rc, wseq = orc.homopolymer_path(post, wseq)
            wb, wpos = orc.overlapper(wseq, 1024)
            assert (c["bases"] if c else None) == wb and np.float32(c["score"]) == np.float32(wsc)
            if wb is not None:
                assert np.array_equal(c["pos"], wpos)
        worst = max(worst, d)
    print("config 3 %s: %d reads, %.1f M samples; worst posterior difference over %d sampled reads %.3g" %
          (name, n, sum(len(x) for x in sigs) / 1e6, len(sample), worst))
    # (2) batch independence for every read: other company, other launch-group cut
    perm = rng.permutation(n)
    try:
        eng.set_max_launch_reads(304)
        small = eng.basecall([sigs[j] for j in perm], name, params)
    finally:
        eng.set_max_launch_reads(16384)
    assert [key(small[k]) for k in np.argsort(perm)] == [key(c) for c in big]


def test_several_engines_dynamic_hand_out(eng, models, tmp_path):
    """scrappie_hip_basecall_batch_multi: engines (here two on device 0; in production one per GPU) take launch
    groups from an atomic cursor over the reads sorted by length.  The calls equal the single-engine ones, read
    for read, whichever engine ran them; the C command line does the same with --devices."""
    w, _ = models["rgrgr_r94"]
    rng = np.random.default_rng(5)
    lens = rng.integers(300, 4001, size=9000)
    base = [sig(int(L), 7000 + i) for i, L in enumerate(lens[:97])]
    sigs = [base[i % 97][:int(lens[i])] if lens[i] <= len(base[i % 97]) else base[i % 97] for i in range(9000)]
    sigs[11] = np.zeros(0, np.float32)
    key = lambda c: None if c is None else (c["bases"], c["score"], c["nblock"])
    one = [key(c) for c in eng.basecall(sigs, "rgrgr_r94")]
    e2 = [sa.Engine(0), sa.Engine(0)]
    try:
        for e in e2:
            e.load_model("rgrgr_r94", w)
        two = [key(c) for c in sa.basecall_multi(e2, sigs, "rgrgr_r94")]
        assert two == one
        assert [key(c) for c in sa.basecall_multi(e2, sigs[:40], "rgrgr_r94")] == one[:40]     # fewer groups than engines
        assert sa.basecall_multi(e2, [], "rgrgr_r94") == []
    finally:
        for e in e2:
            e.close()


@pytest.mark.parametrize("size,nfilter,what", [
    (64, None, "k_gru_proj<4>: one-kernel layers at S = 64"),
    (96, 64, "input narrower than the state in layer 1: projection and recurrence as two kernels there"),
    (128, None, "S = 128: one tile per workgroup (k_gru), register-stationary projection"),
])
def test_other_layer_sizes(eng, orc, size, nfilter, what):
    """Every kernel family the dispatch can pick for a recurrent layer, against the oracle: posterior of
    ragged reads (incl. a Q1-hit length) and a many-read batch equal to the single-read results."""
    w = model.synthetic_model("rgrgr_r94", seed=31 + size, size=size, nfilter=nfilter, nstate=65)
    name = "size%d_%s" % (size, nfilter)
    eng.load_model(name, w)
    om = orc.OracleModel(w)
    sigs = [sig(n, 1300 + i) for i, n in enumerate((900, 643, 1201))]
    for x in sigs:
        post = eng.posterior(x, name)
        assert np.max(np.abs(np.exp(post) - np.exp(orc.posterior(om, x)))) <= P_TOL, what
    key = lambda c: None if c is None else (c["bases"], c["score"], c["nblock"])
    solo = [key(c) for c in eng.basecall(sigs, name)]
    many = eng.basecall([sigs[i % 3] for i in range(40)], name)          # several tiles
    assert [key(c) for c in many] == [solo[i % 3] for i in range(40)], what


def test_unsupported_layer_size_fails_loudly(eng, models):
    """State widths other than 32 / 64 / 96 / 128 are refused with an error, never approximated."""
    w = model.synthetic_model("rgrgr_r94", seed=5, size=48, nstate=65)
    eng.load_model("size48", w)
    with pytest.raises(RuntimeError, match="unsupported"):
        eng.posterior(sig(900, 1), "size48")
    assert eng.basecall([sig(900, 2)], "rgrgr_r94")[0] is not None      # the engine is still usable


# ------------------------------------------------------------------ operand range of the split products
def test_weight_outside_split_range_runs_exact_fp32_layer(eng, orc):
    """|w| >= 255 cannot be held by the fp16 pieces (256 w would be inf).  A recurrent layer with such a weight must
    run on the exact-fp32 kernels (k_affine<.., F32> + k_gru_lanes), chosen at model load, and still match the
    oracle at the fp32 tolerances; the layers around it keep the split products."""
    w = model.synthetic_model("rgrgr_r94", seed=11)
    w["gru2_iW"] = w["gru2_iW"].copy(); w["gru2_iW"][7, 3] = 300.0
    w["gru3_sW"] = w["gru3_sW"].copy(); w["gru3_sW"][100, 11] = -420.0
    w["gru3_sW2"] = w["gru3_sW2"].copy(); w["gru3_sW2"][5, 80] = 256.0
    eng.load_model("bigw", w)
    om = orc.OracleModel(w)
    for N in (1500, 1203):
        x = sig(N, 77 + N)
        for upto in (3, 4, 5):
            got, want = eng.trunk(x, "bigw", upto), orc.trunk(om, x, upto)
            assert np.max(np.abs(got - want)) <= ACT_TOL, (N, upto, float(np.max(np.abs(got - want))))
        got, want = eng.posterior(x, "bigw"), orc.posterior(om, x)
        assert np.max(np.abs(np.exp(got.astype(np.float64)) - np.exp(want.astype(np.float64)))) <= P_TOL
    # many reads (several tiles, lane schedule) through the fallback layers == the single-read results
    sigs = [sig(900 + 31 * i, 400 + i) for i in range(5)]
    key = lambda c: None if c is None else (c["bases"], c["score"], c["nblock"])
    solo = [key(c) for c in eng.basecall(sigs, "bigw")]
    assert [key(c) for c in eng.basecall([sigs[i % 5] for i in range(80)], "bigw")] == [solo[i % 5] for i in range(80)]


def test_weight_outside_split_range_is_refused_where_no_fp32_kernel_exists(eng):
    """... and in a layer without an exact-fp32 kernel (the output layer, the joining layers, the LSTM) the model
    is refused at load with a message -- never silently turned into inf."""
    w = model.synthetic_model("rgrgr_r94", seed=11)
    w["ff_W"] = w["ff_W"].copy(); w["ff_W"][1000, 5] = 300.0
    with pytest.raises(RuntimeError, match="outside the split products' range"):
        eng.load_model("bigff", w)
    w = model.synthetic_model("nanonet_events", seed=17)
    w["lstm1_sW"] = w["lstm1_sW"].copy(); w["lstm1_sW"][3, 3] = -1e4
    with pytest.raises(RuntimeError, match="outside the split products' range"):
        eng.load_model("bigl", w)
    w = model.synthetic_model("rgrgr_r94", seed=11)
    w["gru0_sW"] = w["gru0_sW"].copy(); w["gru0_sW"][0, 0] = np.inf
    with pytest.raises(RuntimeError, match="non-finite"):
        eng.load_model("infw", w)
    assert eng.basecall([sig(900, 2)], "rgrgr_r94")[0] is not None      # the engine is still usable


def test_unnormalised_signal_in_and_out_of_operand_range(eng, orc, models):
    """layers.c:159-246 takes any signal; the reference's callers normalise it, the per-read nanonet_*_posterior
    does not require it.  (1) A pA-scale signal (90 +- 12, conv outputs of a few hundred) is INSIDE the split
    products' range (|activation| < 1000): posterior within tolerance of the oracle's on the same raw input.
    (2) A signal whose first-layer activations leave the range is reported: the per-read surface returns NULL
    with scrappie_hip_last_error() set, a batch gives that read no call and leaves its tile-mates' calls
    untouched -- never a silent clamp, never inf/NaN."""
    w, om = models["rgrgr_r94"]
    raw = synth.synthetic_signal(2000, 91, raw_units=True)           # ~90 +- 12, not normalised
    got = eng.posterior(raw, "rgrgr_r94")
    want = orc.posterior(om, raw)
    assert np.max(np.abs(eng.trunk(raw, "rgrgr_r94", 0))) > 100.0     # the activations really are large
    assert np.max(np.abs(np.exp(got.astype(np.float64)) - np.exp(want.astype(np.float64)))) <= P_TOL
    huge = (raw * 1000.0).astype(np.float32)
    with pytest.raises(RuntimeError, match="outside the supported range"):
        eng.posterior(huge, "rgrgr_r94")
    with pytest.raises(RuntimeError, match="outside the supported range"):
        eng.trunk(huge, "rgrgr_r94", 2)
    nanny = sig(1500, 5).copy(); nanny[700] = np.nan
    good = [sig(1200 + 37 * i, 900 + i) for i in range(5)]
    alone = eng.basecall(good, "rgrgr_r94")
    mixed = [good[0], huge, good[1], good[2], nanny, good[3], good[4]]
    calls = eng.basecall(mixed, "rgrgr_r94")
    assert calls[1] is None and calls[4] is None
    assert "outside the supported range" in sa.last_error() and "2 read" in sa.last_error()
    for j, k in enumerate((0, 2, 3, 5, 6)):
        assert calls[k]["bases"] == alone[j]["bases"] and calls[k]["score"] == alone[j]["score"]
    # events: the same check on the feature columns
    f3 = sa.event_features(synth.synthetic_events(300, 61))
    f3[100, 3] = 5000.0
    with pytest.raises(RuntimeError, match="outside the supported range"):
        eng.posterior(f3.ravel(), "nanonet_events")


def test_multi_engine_failure_leaves_nothing_behind(eng, models):
    """scrappie_hip_basecall_batch_multi with a failure injected on one engine after some launch groups were
    already collected: the call returns -1 with the engine's message, out[] holds NO strings (a failed call returns
    nothing; nothing leaks, nothing for the caller to guess), and the engines serve the next call."""
    w, _ = models["rgrgr_r94"]
    base = [sig(300 + 11 * (i % 23), 8000 + i) for i in range(53)]
    sigs = [base[(i * 5) % 53] for i in range(20000)]                 # >= 5 launch groups of >= 4096
    e2 = [sa.Engine(0), sa.Engine(0)]
    try:
        for e in e2:
            e.load_model("rgrgr_r94", w)
        e2[1].debug_option("fail_run", 2)                               # engine 1: its second launch group is refused
        n = len(sigs)
        keep = [np.ascontiguousarray(x, dtype=np.float32) for x in sigs]
        rts = (sa._RawTable * n)()
        for i, x in enumerate(keep):
            rts[i] = sa._RawTable(None, len(x), 0, len(x), x.ctypes.data_as(C.POINTER(C.c_float)))
        calls = (sa._Call * n)()
        C.memset(calls, 0xff, C.sizeof(calls))                          # garbage in: the call must not trust it
        hs = (C.c_void_p * 2)(*[e._h for e in e2])
        ms = (C.c_int * 2)(*[e._models["rgrgr_r94"] for e in e2])
        rc = sa.lib().scrappie_hip_basecall_batch_multi(hs, ms, 2, rts, n, C.byref(e2[0].default_params()), calls)
        assert rc != 0 and "injected failure" in sa.last_error()
        raw = np.frombuffer(calls, dtype=np.uint64).reshape(n, C.sizeof(sa._Call) // 8)
        assert not raw[:, 2].any() and not raw[:, 4].any() and not raw[:, 3].any()      # basecall, pos, length: all NULL / 0
        key = lambda c: None if c is None else (c["bases"], c["score"], c["nblock"])
        again = [key(c) for c in sa.basecall_multi(e2, sigs[:9000], "rgrgr_r94")]
        assert again == [key(c) for c in eng.basecall(sigs[:9000], "rgrgr_r94")]
        # single engine, several launch groups, failure in the third: same contract
        e2[0].set_max_launch_reads(64)
        e2[0].debug_option("fail_run", 3)
        with pytest.raises(RuntimeError, match="injected failure"):
            e2[0].basecall(sigs[:400], "rgrgr_r94")
        e2[0].set_max_launch_reads(16384)
        assert [key(c) for c in e2[0].basecall(sigs[:50], "rgrgr_r94")] == again[:50]
    finally:
        for e in e2:
            e.close()


# ------------------------------------------------------------------ D2 + D3 on the device (k_stitch)
def _side_rows(post, klen):
    """the five posterior rows homopolymer_path reads (homopolymer.c:200,209): homopolymer k-mers of A, C, G, T; stay"""
    rep = [sum(b * 4 ** i for i in range(klen)) for b in range(4)]
    return np.ascontiguousarray(np.concatenate([post[:, rep], post[:, -1:]], axis=1), dtype=np.float32)


def test_device_stitch_bit_exact_vs_reference_fixture(eng, golden):
    """k_stitch (homopolymer correction + k-mer stitching on the device, what the batched path runs) on the
    compiled reference's own Viterbi paths: bases after homopolymer.c:175 + decode.c:449, bases and pos[] of
    decode.c:449 alone, and decode.c:895 for the CRF paths -- all identical to the compiled reference's."""
    g = golden["ref_decode"]
    nredo = 0
    for (T, seed, klen, stay, skip, local, slip, hp) in g["transducer_cases"]:
        T, seed, klen, hp = int(T), int(seed), int(klen), int(hp)
        post = synth.fixture_posterior(T, seed, klen, hp)
        path = g["seq_%d" % seed]
        bases, pos, redo = eng.debug_stitch(path, None, 4 ** klen + 1)
        assert (bases or "") == str(g["bases_%d" % seed]) and redo == 0, seed
        if bases:
            assert np.array_equal(pos, g["pos_%d" % seed]), seed
        bases, pos, redo = eng.debug_stitch(path, _side_rows(post, klen), 4 ** klen + 1)
        nredo += redo
        if not redo:
            assert (bases or "") == str(g["hp_bases_%d" % seed]), seed
    assert nredo <= 1
    for T, seed in g["crf_cases"]:
        bases, pos, redo = eng.debug_stitch(g["crf_path_%d" % int(seed)], None, 25, crf=True)
        assert bases == str(g["crf_bases_%d" % int(seed)])
    assert eng.debug_stitch(np.full(9, -1, np.int32), None, 1025)[0] is None          # all stays: no call (decode.c:456-461)


def test_device_stitch_equals_host_code_on_many_paths(eng, orc):
    """... and against the host C (sh_host.c, itself bit-exact against the compiled reference) on 300 decoded paths
    of HMM-like posteriors with planted homopolymer runs (3-, 4- and 5-mers, with and without the slip move)."""
    L = sa.lib()
    nrun_changed = nredo = 0
    for i in range(300):
        klen = (5, 5, 4, 3)[i % 4]
        T = 40 + (i * 37) % 700
        post, _ = synth.simulated_posterior(T, 6000 + i, klen=klen, plant_homopolymers=1 + i % 7)
        sc, seq = orc.decode_transducer(post, 0.0, 0.1 * (i % 3), 2.0, bool(i & 1) and klen > 3)
        want_seq = seq.copy()
        pm = sa.ScrappyMatrix.from_numpy(post, sloika=False)
        assert L.homopolymer_path(pm.data(), want_seq.ctypes.data_as(ip), 1) == 0
        nrun_changed += int(not np.array_equal(want_seq, seq))
        wpos = np.zeros(T + 1, np.int32)
        wb = sa._take_string(L.overlapper(want_seq.ctypes.data_as(ip), T + 1, 4 ** klen, wpos.ctypes.data_as(ip)))
        bases, pos, redo = eng.debug_stitch(seq, _side_rows(post, klen), 4 ** klen + 1)
        nredo += redo
        if redo:
            continue
        assert bases == wb, i
        if wb:
            assert np.array_equal(pos, wpos), i
    assert nrun_changed > 30 and nredo <= 3, (nrun_changed, nredo)


def test_device_stitch_equals_host_stitch_end_to_end(eng, hmm_model, tmp_path):
    """The batched path with the device's stitching (default: inside k_walk_stitch_out; SH_SPLIT_TAIL=1: the three-kernel form with
    k_stitch) against the same path stitched on host threads (SH_HOST_STITCH=1, a
    second process: paths and side rows over PCIe, sh_host.c) on 4300 reads of HMM-like posteriors: every call
    identical, pos[] included; and the host fallback for reads the device will not decide (forced for every read)."""
    import subprocess
    import sys
    import json
    w, _ = hmm_model
    mpath = str(tmp_path / "hmm.scrm")
    model.save_model(w, mpath)
    code = """
import sys, json, hashlib, numpy as np
sys.path.insert(0, %r)
import scrappie_amd as sa
from scrappie_amd import synth
e = sa.Engine(0); e.load_model("hmm", %r)
Ts = [800, 640, 333, 801, 97, 12, 500]
e.set_trunk_input([synth.hmm_trunk(T, 4100 + i, plant_homopolymers=4)[0] for i, T in enumerate(Ts)])
sigs = [synth.medmad_normalise(synth.synthetic_signal(5 * Ts[i %% 7], 600 + i %% 7)) for i in range(4300)]
out = []
for kw in (dict(want_pos=1), dict(homopolymer=0), dict(use_slip=1, skip_pen=0.2)):
    h = hashlib.sha256()
    nb = 0
    for c in e.basecall(sigs, "hmm", e.default_params(**kw)):
        h.update(repr(None if c is None else (c["bases"], np.float32(c["score"]).tobytes().hex(), c["nblock"], c.get("pos", np.zeros(0)).tobytes().hex())).encode())
        nb += len(c["bases"]) if c else 0
    out.append([h.hexdigest(), nb])
print(json.dumps(out))
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), mpath)
    got = []
    # (the default is ONE kernel behind the decoder, k_walk_stitch_out; SH_SPLIT_TAIL=1: k_backtrace -> k_stitch -> k_results_out)
    for extra in ({}, {"SH_SPLIT_TAIL": "1"}, {"SH_HOST_STITCH": "1"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        got.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert got[0] == got[1] == got[2]
    assert got[0][0][1] > 0.3 * 4300 * 450                      # realistic calls
    # the fallback: every read re-stitched by the host code from the path and side rows left on the device
    Ts = [800, 640, 333, 801, 97, 12, 500]
    sigs = [sig(5 * Ts[i % 7], 600 + i % 7) for i in range(300)]
    key = lambda c: (c["bases"], c["score"], c["nblock"], c["pos"].tobytes())
    try:
        eng.set_trunk_input([synth.hmm_trunk(T, 4100 + i, plant_homopolymers=4)[0] for i, T in enumerate(Ts)])
        p = eng.default_params(want_pos=1)
        a = [key(c) for c in eng.basecall(sigs, "hmm", p)]
        n0 = int(eng.debug_fetch("n_redo", np.uint64)[0])
        eng.debug_option("redo_all", 1)
        b = [key(c) for c in eng.basecall(sigs, "hmm", p)]
        assert int(eng.debug_fetch("n_redo", np.uint64)[0]) == n0 + 300
    finally:
        eng.debug_option("redo_all", 0)
        eng.set_trunk_input(None)
    assert a == b


def test_chain_bound_reads_run_beside_the_rest(models):
    """A call with a long tail of read lengths: scrappie_hip_basecall_batch hands the reads whose own serial chain would outlast
    the rest of the call (scrappie_hip_plan_tail) to a helper engine on the same device and runs both at once.  The calls are
    those of the unsplit call (a read's call never depends on its batch), the split did happen, and a failure or a later model
    load do not upset the pair."""
    w, _ = models["rgrgr_r94"]
    e = sa.Engine(0)
    try:
        e.load_model("rgrgr_r94", w)
        long_reads = [sig(120000 + 5000 * i, 700 + i) for i in range(3)]
        base = [sig(1500 + 13 * (i % 40), 6000 + i) for i in range(60)]
        reads = []
        for i in range(3000):
            reads.append(base[(i * 7) % 60])
            if i in (5, 1700, 2999):
                reads.append(long_reads[len([r for r in reads if len(r) > 100000])])
        lens = np.array([len(r) for r in reads], np.uint32)
        assert sa.plan_tail(lens, 5).sum() == 3
        key = lambda c: None if c is None else (c["bases"], c["score"], c["nblock"])
        p = e.default_params(local_pen=150.0)
        e.debug_option("tail", 0)
        whole = [key(c) for c in e.basecall(reads, "rgrgr_r94", p)]
        e.debug_option("tail", 1)
        n0 = int(e.debug_fetch("n_tail_calls", np.uint64)[0])
        split = [key(c) for c in e.basecall(reads, "rgrgr_r94", p)]
        assert int(e.debug_fetch("n_tail_calls", np.uint64)[0]) == n0 + 1 and int(e.debug_fetch("n_tail_reads", np.uint64)[0]) >= 3
        assert split == whole and all(k is not None and k[2] == (len(r) + 4) // 5 for k, r in zip(split, reads))
        # a model loaded after the helper exists reaches it too (same index on both engines)
        w2 = model.synthetic_model("rgrgr_r10", seed=5)
        e.load_model("rgrgr_r10", w2)
        a = [key(c) for c in e.basecall(reads[:1800], "rgrgr_r10", p)]
        e.debug_option("tail", 0)
        b = [key(c) for c in e.basecall(reads[:1800], "rgrgr_r10", p)]
        assert a == b
    finally:
        e.close()


def test_deferred_chain_bound_reads_across_calls(models):
    """scrappie_hip_basecall_batch_deferred: a stream of calls whose long reads are collected later.  Every call -- the ones returned at
    once and the deferred ones -- equals the unsplit call's; tickets queued while the helper is busy are served as one launch
    group; a ticket can be polled; an uncollected ticket does not leak or hang at engine destruction."""
    w, _ = models["rgrgr_r94"]
    e = sa.Engine(0)
    try:
        e.load_model("rgrgr_r94", w)
        base = [sig(1500 + 13 * (i % 40), 6000 + i) for i in range(60)]
        key = lambda c: None if c is None else (c["bases"], c["score"], c["nblock"])
        p = e.default_params(local_pen=150.0)
        batches = []
        for k in range(4):
            reads = [base[(i * 7 + k) % 60] for i in range(2500)]
            reads.insert(100 + k, sig(150000 + 7000 * k, 800 + k))
            reads.insert(2000, sig(110000, 900 + k))
            batches.append(reads)
        e.debug_option("tail", 0)
        want = [[key(c) for c in e.basecall(r, "rgrgr_r94", p)] for r in batches]
        e.debug_option("tail", 1)
        got, tickets = [], []
        for r in batches:
            calls, tk, deferred = e.basecall_deferred(r, "rgrgr_r94", p)
            assert tk > 0 and deferred.sum() == 2 and all((c is None) == bool(d) for c, d in zip(calls, deferred))
            got.append([key(c) for c in calls]); tickets.append((tk, np.flatnonzero(deferred)))
        polled = e.collect_deferred(tickets[-1][0], wait=False)          # the last one is most likely still running
        for k, (tk, idx) in enumerate(tickets):
            late = polled if (k == len(tickets) - 1 and polled is not None) else e.collect_deferred(tk)
            assert len(late) == len(idx)
            for i, c in zip(idx, late):
                got[k][i] = key(c)
        assert got == want
        ng = int(e.debug_fetch("n_tail_groups", np.uint64)[0])
        assert 1 <= ng <= len(batches)
        print("deferred: %d tickets served by %d helper launch groups" % (len(tickets), ng))
        # equal reads: nothing deferred
        calls, tk, deferred = e.basecall_deferred(base, "rgrgr_r94", p)
        assert tk == 0 and not deferred.any() and all(c is not None for c in calls)
        # a ticket nobody collects
        calls, tk, deferred = e.basecall_deferred(batches[0], "rgrgr_r94", p)
        assert tk > 0
    finally:
        e.close()


def test_failure_on_the_helper_engine_returns_nothing(models):
    """the helper engine refuses its launch group (injected): the call fails as a whole -- no calls, no strings left behind, the error
    text says why -- and the pair of engines is usable afterwards; the same for a deferred ticket, whose failure surfaces at collect"""
    w, _ = models["rgrgr_r94"]
    e = sa.Engine(0)
    try:
        e.load_model("rgrgr_r94", w)
        base = [sig(1500 + 13 * (i % 40), 6000 + i) for i in range(60)]
        reads = [base[(i * 7) % 60] for i in range(2000)] + [sig(130000, 31), sig(140000, 32)]
        key = lambda c: None if c is None else (c["bases"], c["score"], c["nblock"])
        e.debug_option("tail", 0)
        want = [key(c) for c in e.basecall(reads, "rgrgr_r94")]
        e.debug_option("tail", 1)
        e.debug_option("fail_tail", 1)
        with pytest.raises(RuntimeError):
            e.basecall(reads, "rgrgr_r94")
        assert [key(c) for c in e.basecall(reads, "rgrgr_r94")] == want
        e.debug_option("fail_tail", 1)
        calls, tk, deferred = e.basecall_deferred(reads, "rgrgr_r94")
        assert tk > 0 and deferred.sum() == 2
        with pytest.raises(RuntimeError):
            e.collect_deferred(tk)
        calls, tk, deferred = e.basecall_deferred(reads, "rgrgr_r94")
        late = e.collect_deferred(tk)
        got = [key(c) for c in calls]
        for i, c in zip(np.flatnonzero(deferred), late):
            got[i] = key(c)
        assert got == want
    finally:
        e.close()


@pytest.mark.gpu
def test_per_read_calls_from_many_threads_share_launch_groups(models):
    """The reference's per-read network functions (networks.h:22, python/pyscrap.h:11-23) called from many host threads at once, as its
    OpenMP loop over reads does (scrappie_raw.c:355,387): the calls are coalesced into launch groups, and every caller gets, bit for bit,
    the matrix it gets alone; a bad read fails alone; transducer and CRF models, log and probability outputs, mixed lengths."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    L = sa.lib()
    L.scrappie_hip_coalescer_stats.argtypes = [C.POINTER(C.c_ulonglong)]
    st0 = (C.c_ulonglong * 3)()
    L.scrappie_hip_coalescer_stats(st0)
    rng = np.random.default_rng(3)
    jobs = []
    for i in range(96):
        name = ("rgrgr_r94", "rgrgr_r10", "rnnrf_r94")[i % 3]
        n = int(rng.integers(300, 6000))
        log = True if name == "rnnrf_r94" else bool(i % 2)
        jobs.append((name, sig(n, 7000 + i), 1e-5 if i % 4 else 1e-3, log))
    jobs.append(("rgrgr_r94", np.zeros(12, np.float32), 1e-5, True))             # below the model's minimum: fails, alone
    jobs.append(("rgrgr_r94", np.full(2000, 5.0e6, np.float32), 1e-5, True))     # outside the operand range: fails, alone

    def one(j):
        name, x, mp, log = j
        try:
            return sa.calc_post(sa.RawTable(x), name, min_prob=mp, log=log).data(as_numpy=True)
        except RuntimeError as err:
            return str(err)
    with ThreadPoolExecutor(32) as pool:
        together = list(pool.map(one, jobs))
    st1 = (C.c_ulonglong * 3)()
    L.scrappie_hip_coalescer_stats(st1)
    assert st1[1] - st0[1] == len(jobs) and st1[0] - st0[0] < len(jobs) // 2 and st1[2] >= 4, list(st1)
    alone = [one(j) for j in jobs]
    for j, a, b in zip(jobs, together, alone):
        if isinstance(b, str):
            assert isinstance(a, str) and ("minimum" in a or "range" in a), a
        else:
            assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), j[0]
    assert isinstance(together[-1], str) and isinstance(together[-2], str)


@pytest.mark.gpu
def test_per_read_decode_from_many_threads_is_the_single_read_decode(models, orc):
    """decode_transducer (decode.h:13) from many host threads at once runs as one launch, one workgroup per call: path and score of every call are the
    ones it gets alone, and the oracle's (decode.c:123-365 restated) -- posteriors of different lengths, two parameter sets, 3-mer and 5-mer models."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    L = sa.lib()
    L.scrappie_hip_decode_coalescer_stats.argtypes = [C.POINTER(C.c_ulonglong)]
    st0 = (C.c_ulonglong * 3)()
    L.scrappie_hip_decode_coalescer_stats(st0)
    rng = np.random.default_rng(9)
    posts = []
    for i in range(40):
        n = int(rng.integers(400, 5000))
        posts.append(sa.calc_post(sa.RawTable(sig(n, 8100 + i)), "rgrgr_r94", min_prob=1e-5, log=True))
    jobs = [(p, dict(local_pen=150.0) if k % 2 else dict()) for k, p in enumerate(posts)]

    def one(j):
        p, kw = j
        return sa._decode_post(p, **kw)
    with ThreadPoolExecutor(20) as pool:
        together = list(pool.map(one, jobs))
    st1 = (C.c_ulonglong * 3)()
    L.scrappie_hip_decode_coalescer_stats(st1)
    assert st1[1] - st0[1] == len(jobs) and st1[0] - st0[0] < len(jobs) // 2 and st1[2] >= 4, list(st1)
    alone = [one(j) for j in jobs]
    nlong = 0
    for (p, kw), a, b in zip(jobs, together, alone):
        assert a[0] == b[0] and np.float32(a[1]) == np.float32(b[1]) and np.array_equal(a[2], b[2])
        lp = p.data(as_numpy=True, sloika=False)
        wsc, wseq = orc.decode_transducer(lp, 0.0, 0.0, kw.get("local_pen", 2.0), False)
        wb, wpos = orc.overlapper(wseq, 1024)
        assert a[0] == (wb or "") or (a[0] is None and wb is None) or a[0] == wb
        assert np.float32(a[1]) == np.float32(wsc)
        nlong += len(a[0] or "") > 100
    assert nlong >= 10


@pytest.mark.gpu
def test_per_read_crf_decode_from_many_threads(models, orc):
    """decode_crf (decode.h:22) from many host threads: one launch, a thread per call; path, score and bases of every call are the ones it gets alone and
    the oracle's (decode.c:836-918 restated)."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    L = sa.lib()
    L.scrappie_hip_crf_coalescer_stats.argtypes = [C.POINTER(C.c_ulonglong)]
    st0 = (C.c_ulonglong * 3)()
    L.scrappie_hip_crf_coalescer_stats(st0)
    rng = np.random.default_rng(21)
    posts = [sa.calc_post(sa.RawTable(sig(int(rng.integers(300, 6000)), 8300 + i)), "rnnrf_r94", log=True) for i in range(48)]
    with ThreadPoolExecutor(24) as pool:
        together = list(pool.map(sa._decode_post_crf, posts))
    st1 = (C.c_ulonglong * 3)()
    L.scrappie_hip_crf_coalescer_stats(st1)
    assert st1[1] - st0[1] == len(posts) and st1[0] - st0[0] < len(posts) // 2 and st1[2] >= 4, list(st1)
    alone = [sa._decode_post_crf(p) for p in posts]
    for p, a, b in zip(posts, together, alone):
        assert a[0] == b[0] and np.float32(a[1]) == np.float32(b[1])
        tr = p.data(as_numpy=True, sloika=False)
        wsc, wpath = orc.decode_crf(tr)
        assert np.float32(a[1]) == np.float32(wsc)
        assert a[0] == (orc.crfpath_to_basecall(wpath, len(wpath) - 1) or "")
    assert sum(len(a[0]) > 100 for a in together) >= 24


@pytest.mark.gpu
def test_per_read_events_posterior_from_many_threads(eng, models):
    """nanonet_posterior (networks.c:146) from many host threads: coalesced like the raw models' functions, every matrix the one the call gets alone
    (= the explicit engine's posterior of the same features)."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    L = sa.lib()
    L.nanonet_posterior.restype = C.POINTER(sa._Mat)
    L.nanonet_posterior.argtypes = [sa._EventTable, C.c_float, C.c_float, C.c_float, C.c_bool]
    evs = [np.ascontiguousarray(synth.synthetic_events(int(n), 500 + i)) for i, n in enumerate(np.random.default_rng(4).integers(40, 900, 40))]

    def one(ev):
        et = sa._EventTable(len(ev), 0, len(ev), C.cast(ev.ctypes.data, C.POINTER(sa._Event)))
        pm = L.nanonet_posterior(et, 1e-5, 1.0, 1.0, True)
        assert pm, sa.last_error()
        return sa.ScrappyMatrix(pm).data(as_numpy=True, sloika=False)
    with ThreadPoolExecutor(20) as pool:
        together = list(pool.map(one, evs))
    for ev, got in zip(evs, together):
        want = eng.posterior(sa.event_features(ev).ravel(), "nanonet_events")
        assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))


# ---------------------------------------------------------------------------
# P0 on the device (k_p0): trim_and_segment_raw + medmad_normalise_array for a batch, bit-identical to the host functions
# ---------------------------------------------------------------------------
def _p0_signal(n, seed, kind):
    rng = np.random.default_rng(seed)
    sig = synth.synthetic_signal(max(n, 2), seed, raw_units=True)[:n].copy()
    if kind == 1:          # quiet stretches at both ends (what the segmentation is for)
        a, b = min(n, 250), min(n, 130)
        sig[:a] = sig[:a] * np.float32(0.02) + np.float32(90)
        sig[n - b:] = sig[n - b:] * np.float32(0.02) + np.float32(90)
    elif kind == 2:        # DAC-quantised: many equal samples (ties in every order statistic)
        sig = np.round(sig).astype(np.float32)
    elif kind == 3:        # constant chunks: chunk MADs of exactly zero, a zero threshold
        sig[: n // 2] = np.float32(80.0)
    elif kind == 4:        # negative and positive values, zeros of both signs
        sig = (sig - np.float32(90)).astype(np.float32)
        sig[rng.integers(0, n, size=max(1, n // 50))] = np.float32(0.0)
        sig[rng.integers(0, n, size=max(1, n // 50))] = np.float32(-0.0)
    return sig.astype(np.float32)


def _host_p0(sig, st, en, ts, te, chunk, perc):
    rt = sa.RawTable(sig, st, en)
    if chunk:
        rt.trim(ts, te, chunk, perc)
    else:                   # no segmentation: the fixed trims alone (scrappie_common.c:12-16)
        s = st + ts if (len(sig) - st) > ts else len(sig)
        e = en - te if en > te else 0
        rt._rt.start, rt._rt.end = (s, e) if s < e else (0, 0)
    rt.scale()
    return rt.start, rt.end, rt.data(as_numpy=True)


@pytest.mark.gpu
def test_device_signal_prep_matches_reference_fixtures(golden):
    """k_p0 against outputs of the reference's own trim_raw_by_mad / medmad_normalise_array (oracle/_ref/libref_pure.so,
    tests/golden/ref_signal_prep.npz): windows equal, normalised samples bit-identical."""
    g = golden["ref_signal_prep"]
    prep = sa.Prep(0)
    sigs, chunks = [], []
    for n, seed in g["cases"]:
        n, seed = int(n), int(seed)
        sig = synth.synthetic_signal(n, seed, raw_units=True)
        if seed % 2 == 0:
            sig[:250] = sig[:250] * 0.02 + 90
            sig[-130:] = sig[-130:] * 0.02 + 90
        sigs.append(sig.astype(np.float32)); chunks.append(100 if n >= 200 else 10)
    for perc in (0.0, 0.25):
        for chunk in sorted(set(chunks)):
            sel = [i for i, c in enumerate(chunks) if c == chunk]
            _, off, ln, st, en = prep.run([sigs[i] for i in sel], 0, 0, chunk, perc)
            for k, i in enumerate(sel):
                seed = int(g["cases"][i][1])
                want = [int(v) for v in g["trim_%d_%g" % (seed, perc)]]
                got = [int(st[k]), int(en[k])] if ln[k] else None
                assert got == want or (got is None and want[0] >= want[1]), (seed, perc, got, want)
    _, off, ln, st, en = prep.run(sigs, 0, 0, 0, 0.0)         # no segmentation: the whole read is normalised
    for i, sig in enumerate(sigs):
        seed = int(g["cases"][i][1])
        assert (int(st[i]), int(en[i])) == (0, len(sig))
        got = prep.fetch(off[i], ln[i])
        assert np.array_equal(got.view(np.uint32), g["norm_%d" % seed].view(np.uint32)), seed
    prep.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ts,te,chunk,perc", [(200, 10, 100, 0.0), (200, 10, 100, 0.3), (0, 0, 10, 0.5), (50, 70, 1000, 0.1),
                                              (200, 10, 1500, 0.0), (3, 5, 1, 0.0), (200, 10, 64, 1.0), (10, 10, 37, 0.77),
                                              (200, 10, 100, 0.37), (0, 0, 23, 0.61)])     # (round 6: percentiles whose interpolation weight is neither 0 nor 0.5 -- util.c:121's float product)
def test_device_signal_prep_equals_host_functions(ts, te, chunk, perc, stage_capacity=None):
    """A ragged batch (1 ... 100 000 samples; quiet ends, quantised values, constant stretches, signed zeros; windows that come
    out empty; entry windows that do not start at 0) through k_p0 and, read by read, through trim_and_segment_raw +
    medmad_normalise_array of sh_host.c (themselves bit-exact against the compiled reference, test_host_cpu.py)."""
    lens = [1, 2, 3, 63, 64, 65, 100, 199, 200, 211, 257, 300, 999, 1000, 1001, 2047, 2048, 3790, 4000, 4000, 4001, 4096,
            8191, 8192, 8193, 9000, 20011, 100000]
    sigs, wins = [], []
    for k, n in enumerate(lens):
        sigs.append(_p0_signal(n, 100 + k, k % 5))
        wins.append((0, n) if k % 7 else (min(n, 17), max(min(n, 17), n - 5)))
    prep = sa.Prep(0)
    _, off, ln, st, en = prep.run(sigs, ts, te, chunk, perc, windows=wins, stage_capacity=stage_capacity)
    if stage_capacity is not None:
        assert 0 < prep.n_staged < len(sigs)          # some reads in the pinned buffer, the others gathered behind them
    nlive = 0
    for i, sig in enumerate(sigs):
        hs, he, hx = _host_p0(sig, wins[i][0], wins[i][1], ts, te, chunk, perc)
        if he <= hs:
            assert ln[i] == 0, (lens[i], hs, he, int(st[i]), int(en[i]))
            continue
        nlive += 1
        assert (int(st[i]), int(en[i]), int(ln[i])) == (hs, he, he - hs), (lens[i], (hs, he), (int(st[i]), int(en[i])))
        got = prep.fetch(off[i], ln[i])
        assert np.array_equal(got.view(np.uint32), hx.view(np.uint32)), (lens[i], int(np.sum(got.view(np.uint32) != hx.view(np.uint32))))
    # (chunks of one sample: every MAD is 0; percentile 1: the threshold is the largest MAD -- nothing is above it, nothing is left)
    assert nlive >= (0 if chunk == 1 or perc == 1.0 else 4 if chunk >= 1000 else 1)
    prep.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cap", [30000, 150000])
def test_device_signal_prep_from_the_staging_buffer(cap):
    """The loader's path: reads placed in the slot's pinned staging buffer through scrappie_hip_prep_alloc (until its capacity is
    used up; the rest stay in ordinary memory and are gathered behind them, which may move the buffer) -- same windows, same
    samples."""
    test_device_signal_prep_equals_host_functions(200, 10, 100, 0.0, stage_capacity=cap)


@pytest.mark.gpu
def test_device_prepared_reads_with_a_long_tail_are_deferred(models):
    """raw reads -> k_p0 -> scrappie_hip_basecall_device_deferred: the chain-bound reads of a batch are copied back into memory their
    ticket owns and run on the helper engine while the NEXT batch (prepared into the same slot: the device buffer is reused at
    once) goes through; every call equals host preparation + the unsplit scrappie_hip_basecall_batch."""
    w, _ = models["rgrgr_r94"]
    e = sa.Engine(0)
    prep = sa.Prep(0)
    try:
        e.load_model("rgrgr_r94", w)
        p = e.default_params(local_pen=150.0)
        key = lambda c: None if c is None else (c["bases"], c["score"], c["nblock"])
        base = [_p0_signal(1700 + 13 * (i % 40), 6000 + i, i % 3) for i in range(50)]
        batches = []
        for k in range(3):
            reads = [base[(i * 7 + k) % 50] for i in range(2500)]
            reads.insert(100 + k, _p0_signal(150000 + 7000 * k, 800 + k, 1))
            reads.insert(2000, _p0_signal(110000, 900 + k, 0))
            batches.append(reads)
        e.debug_option("tail", 0)
        want = []
        for r in batches:
            host = []
            for x in r:
                hs, he, hx = _host_p0(x, 0, len(x), 200, 10, 100, 0.0)
                host.append(hx if he > hs else np.zeros(0, np.float32))
            want.append([key(c) for c in e.basecall(host, "rgrgr_r94", p)])
        e.debug_option("tail", 1)
        got, tickets = [], []
        for r in batches:
            d, off, ln, st, en = prep.run(r, slot=0)              # (the same slot every time)
            calls, tk, deferred = e.basecall_device_deferred(d, off, ln, "rgrgr_r94", p)
            assert tk > 0 and deferred.sum() == 2 and all((c is None) == bool(f) for c, f in zip(calls, deferred))
            got.append([key(c) for c in calls]); tickets.append((tk, np.flatnonzero(deferred)))
        for k, (tk, idx) in enumerate(tickets):
            late = e.collect_deferred(tk)
            assert len(late) == len(idx)
            for i, c in zip(idx, late):
                got[k][i] = key(c)
        assert got == want
    finally:
        prep.close()
        e.close()


@pytest.mark.gpu
def test_streaming_calls_equal_ordinary_calls(models):
    """scrappie_hip_basecall_device_stream: every call returns with its last launch group in flight; the next call (or the flush)
    delivers it into the same out[].  Four batches of mixed lengths, several launch groups each (max_launch_reads lowered): every call
    equals scrappie_hip_basecall_device's; entries of the group in flight are blank until delivered; an ordinary call behind a
    streaming one delivers the carried group first."""
    w, _ = models["rgrgr_r94"]
    e = sa.Engine(0)
    L = sa.lib()
    u64, u32 = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
    L.scrappie_hip_basecall_device_stream.argtypes = [C.c_void_p, C.c_int, C.c_void_p, u64, u32, C.c_size_t, C.POINTER(sa.Params), C.POINTER(sa._Call)]
    L.scrappie_hip_stream_flush.argtypes = [C.c_void_p]
    L.scrappie_hip_stream_pending.argtypes = [C.c_void_p]
    try:
        e.load_model("rgrgr_r94", w)
        e.set_max_launch_reads(256)
        p = e.default_params(local_pen=150.0)
        rng = np.random.default_rng(11)
        key = lambda c: None if not c.basecall else (C.string_at(c.basecall), np.float32(c.score).tobytes(), int(c.nblock))
        batches = []
        for k in range(4):
            lens = rng.integers(300, 3000, size=700 + 50 * k)
            sigs = [sig(int(n), 9000 + 100 * k + i) for i, n in enumerate(lens)]
            flat = np.concatenate(sigs).astype(np.float32)
            off = np.zeros(len(sigs), np.uint64); off[1:] = np.cumsum(lens[:-1]).astype(np.uint64)
            batches.append((e.upload(flat), off, lens.astype(np.uint32)))
        want = []
        for d, off, ln in batches:
            calls = (sa._Call * len(ln))()
            assert L.scrappie_hip_basecall_device(e._h, e._models["rgrgr_r94"], d, off.ctypes.data_as(u64), ln.ctypes.data_as(u32), len(ln), C.byref(p), calls) == 0
            want.append([key(c) for c in calls]); L.scrappie_hip_free_calls(calls, len(ln))
        outs = [(sa._Call * len(ln))() for _, _, ln in batches]
        for k, (d, off, ln) in enumerate(batches[:3]):
            assert L.scrappie_hip_basecall_device_stream(e._h, e._models["rgrgr_r94"], d, off.ctypes.data_as(u64), ln.ctypes.data_as(u32), len(ln), C.byref(p), outs[k]) == 0, sa.last_error()
            assert L.scrappie_hip_stream_pending(e._h) == 1
            nblank = sum(1 for c in outs[k] if not c.basecall)
            assert 0 < nblank <= 256 + 8                      # the launch group still in flight (+ reads that give no call)
            if k > 0:
                assert [key(c) for c in outs[k - 1]] == want[k - 1]          # delivered behind this call's first launch
        # an ordinary call behind the stream delivers the carried group first
        d, off, ln = batches[3]
        assert L.scrappie_hip_basecall_device(e._h, e._models["rgrgr_r94"], d, off.ctypes.data_as(u64), ln.ctypes.data_as(u32), len(ln), C.byref(p), outs[3]) == 0
        assert L.scrappie_hip_stream_pending(e._h) == 0
        assert [key(c) for c in outs[2]] == want[2] and [key(c) for c in outs[3]] == want[3]
        # flush
        for c, (_, _, ln) in zip(outs, batches):
            L.scrappie_hip_free_calls(c, len(ln))
        d, off, ln = batches[0]
        assert L.scrappie_hip_basecall_device_stream(e._h, e._models["rgrgr_r94"], d, off.ctypes.data_as(u64), ln.ctypes.data_as(u32), len(ln), C.byref(p), outs[0]) == 0
        assert L.scrappie_hip_stream_flush(e._h) == 0 and L.scrappie_hip_stream_pending(e._h) == 0
        assert [key(c) for c in outs[0]] == want[0]
        L.scrappie_hip_free_calls(outs[0], len(ln))
        for d, _, _ in batches:
            e.free(d)
    finally:
        e.close()


@pytest.mark.gpu
def test_streaming_calls_that_defer_reads_equal_ordinary_calls(models):
    """scrappie_hip_basecall_device_deferred_stream on batches WITH chain-bound reads: those go to the helper engine (ticket), the
    others run as a streaming call -- its last launch group stays in flight and is delivered, into the places of the caller's out[]
    the subset's reads have there, behind the next call's first launch or by the flush.  Every entry equals the unsplit call's."""
    w, _ = models["rgrgr_r94"]
    e = sa.Engine(0)
    L = sa.lib()
    u64, u32 = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
    fn = L.scrappie_hip_basecall_device_deferred_stream
    fn.restype = C.c_long
    fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, u64, u32, C.c_size_t, C.POINTER(sa.Params), C.POINTER(sa._Call), C.POINTER(C.c_ubyte)]
    L.scrappie_hip_stream_flush.argtypes = [C.c_void_p]
    L.scrappie_hip_stream_pending.argtypes = [C.c_void_p]
    L.scrappie_hip_deferred_collect.restype = C.c_long
    L.scrappie_hip_deferred_collect.argtypes = [C.c_void_p, C.c_long, C.POINTER(sa._Call), C.c_size_t, C.c_int]
    try:
        e.load_model("rgrgr_r94", w)
        e.set_max_launch_reads(512)
        p = e.default_params(local_pen=150.0)
        rng = np.random.default_rng(23)
        key = lambda c: None if not c.basecall else (C.string_at(c.basecall), np.float32(c.score).tobytes(), int(c.nblock))
        batches = []
        for k in range(3):
            lens = rng.integers(300, 2500, size=1400 + 30 * k)
            lens[37 + k] = 150000 + 9000 * k                  # chain-bound: longer than everything else in the call takes
            lens[900] = 120000
            sigs = [sig(int(n), 7000 + 100 * k + i) for i, n in enumerate(lens)]
            flat = np.concatenate(sigs).astype(np.float32)
            off = np.zeros(len(sigs), np.uint64); off[1:] = np.cumsum(lens[:-1]).astype(np.uint64)
            batches.append((e.upload(flat), off, lens.astype(np.uint32)))
        e.debug_option("tail", 0)
        want = []
        for d, off, ln in batches:
            calls = (sa._Call * len(ln))()
            assert L.scrappie_hip_basecall_device(e._h, e._models["rgrgr_r94"], d, off.ctypes.data_as(u64), ln.ctypes.data_as(u32), len(ln), C.byref(p), calls) == 0
            want.append([key(c) for c in calls]); L.scrappie_hip_free_calls(calls, len(ln))
        e.debug_option("tail", 1)
        outs = [(sa._Call * len(ln))() for _, _, ln in batches]
        flags = [(C.c_ubyte * len(ln))() for _, _, ln in batches]
        tickets = []
        for k, (d, off, ln) in enumerate(batches):
            tk = fn(e._h, e._models["rgrgr_r94"], d, off.ctypes.data_as(u64), ln.ctypes.data_as(u32), len(ln), C.byref(p), outs[k], flags[k])
            assert tk > 0, sa.last_error()
            assert sum(flags[k]) == 2 and flags[k][37 + k] == 1 and flags[k][900] == 1
            assert L.scrappie_hip_stream_pending(e._h) == 1
            nblank = sum(1 for c in outs[k] if not c.basecall)
            assert 2 < nblank <= 512 + 2 + 8                  # the deferred reads + the launch group still in flight
            tickets.append(tk)
            if k > 0:                                         # delivered behind this call's first launch; the deferred entries still blank
                assert [key(c) for i, c in enumerate(outs[k - 1]) if not flags[k - 1][i]] == [x for i, x in enumerate(want[k - 1]) if not flags[k - 1][i]]
        assert L.scrappie_hip_stream_flush(e._h) == 0 and L.scrappie_hip_stream_pending(e._h) == 0
        for k, tk in enumerate(tickets):
            got = [key(c) for c in outs[k]]
            idx = [i for i in range(len(got)) if flags[k][i]]
            assert all(got[i] is None for i in idx)
            late = (sa._Call * 2)()
            assert L.scrappie_hip_deferred_collect(e._h, tk, late, 2, 1) == 2, sa.last_error()
            for i, c in zip(idx, late):
                got[i] = key(c)
            L.scrappie_hip_free_calls(late, 2)
            assert got == want[k]
        for c, (_, _, ln) in zip(outs, batches):
            L.scrappie_hip_free_calls(c, len(ln))
        for d, _, _ in batches:
            e.free(d)
    finally:
        e.close()


@pytest.mark.gpu
def test_device_signal_prep_feeds_the_engine():
    """raw reads -> k_p0 -> scrappie_hip_basecall_device == host preparation -> scrappie_hip_basecall_batch, call for call;
    the second slot is prepared while the first is still in use."""
    w = model.synthetic_model("rgrgr_r94", seed=1)
    eng = sa.Engine(0)
    eng.load_model("rgrgr_r94", w)
    prep = sa.Prep(0)
    L = sa.lib()
    p = eng.default_params(want_pos=1)
    for slot, base in ((0, 0), (1, 40)):
        raws = [_p0_signal(n, base + i, i % 3) for i, n in enumerate([4000, 1203, 755, 4000, 260, 230, 5000, 12000])]
        d, off, ln, st, en = prep.run(raws, slot=slot)
        host = []
        for r in raws:
            hs, he, hx = _host_p0(r, 0, len(r), 200, 10, 100, 0.0)
            host.append(hx if he > hs else np.zeros(0, np.float32))
        want = eng.basecall([h for h in host], "rgrgr_r94", p)
        calls = (sa._Call * len(raws))()
        rc = L.scrappie_hip_basecall_device(eng._h, eng._models["rgrgr_r94"], d, off.ctypes.data_as(C.POINTER(C.c_uint64)),
                                            ln.ctypes.data_as(C.POINTER(C.c_uint32)), len(raws), C.byref(p), calls)
        assert rc == 0, sa.last_error()
        got = sa.Engine._unpack(calls, len(raws), 1)
        assert sum(1 for c in got if c) >= 5
        for a, b in zip(got, want):
            assert (a is None) == (b is None)
            if a:
                assert a["bases"] == b["bases"] and np.float32(a["score"]) == np.float32(b["score"]) and np.array_equal(a["pos"], b["pos"])
    prep.close()
    eng.close()
