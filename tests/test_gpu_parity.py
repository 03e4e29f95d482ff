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


def test_rnnrf_transitions(eng, orc, models):
    w, om = models["rnnrf_r94"]
    for N in (2000, 1333):
        x = sig(N, 41)
        got = eng.posterior(x, "rnnrf_r94")
        want = orc.posterior(om, x)
        assert got.shape == want.shape == ((N + 4) // 5, 25)
        print("rnnrf N=%d: max |HIP - oracle| = %.3g" % (N, np.max(np.abs(got - want))))
        assert np.max(np.abs(got - want)) <= CRF_TOL


# ------------------------------------------------------------------ decode (integer path)
def test_decode_transducer_bit_exact_vs_reference_fixture(golden):
    """GPU Viterbi on the SAME posterior as the compiled reference decode.c:
    path and score must be identical (incl. slip, penalties, 3/4/5-mers)."""
    g = golden["ref_decode"]
    for (T, seed, klen, stay, skip, local, slip, hp) in g["transducer_cases"]:
        T, seed, klen, hp = int(T), int(seed), int(klen), int(hp)
        post, _ = synth.simulated_posterior(T, seed, klen=klen, plant_homopolymers=hp)
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


def test_batch_integer_path_exact_given_gpu_posterior(eng, orc, models):
    """Engine.basecall (device Viterbi + host homopolymer/stitching) must equal
    the oracle's decode applied to the engine's own posterior, bit for bit."""
    w, om = models["rgrgr_r94"]
    lens = [1500, 1203, 777, 2000, 1500, 901, 350, 1777, 1500, 640, 1234, 1999, 1500, 1001, 1502, 888, 1640, 455]
    sigs = [sig(n, 100 + i) for i, n in enumerate(lens)]
    for hp in (1, 0):
        params = eng.default_params(homopolymer=hp, want_pos=1)
        calls = eng.basecall(sigs, "rgrgr_r94", params)
        for x, c in zip(sigs, calls):
            post = eng.posterior(x, "rgrgr_r94", min_prob=1e-5)
            wsc, wseq = orc.decode_transducer(post)
            if hp:
                rc, wseq = orc.homopolymer_path(post, wseq)
            wb, wpos = orc.overlapper(wseq, 1024)
            if wb is None:
                assert c is None
                continue
            assert c["bases"] == wb
            assert np.array_equal(c["pos"], wpos)
            assert np.float32(c["score"]) == np.float32(wsc)
            assert c["nblock"] == post.shape[0]


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
for kw in (dict(), dict(tempW=1.3, tempb=0.8, use_slip=1, stay_pen=0.3, skip_pen=0.2), dict(use_slip=1, local_pen=1.0, homopolymer=0)):
    h = hashlib.sha256()
    for c in e.basecall(reads, "rgrgr_r94", e.default_params(**kw)):
        h.update(repr(None if c is None else (c["bases"], np.float32(c["score"]).tobytes().hex(), c["nblock"])).encode())
    out.append(h.hexdigest())
print(json.dumps(out))
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), mpath)
    got = []
    for extra in ({}, {"SH_FF_SEPARATE": "1"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        got.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert got[0] == got[1]
    assert len(set(got[0])) == 3            # the three parameter sets do decode differently


def test_batch_end_to_end_vs_oracle(eng, orc, models):
    """GPU posterior vs CPU posterior may differ in the last bits, so paths may
    legitimately differ at near ties (SURVEY section 7 'bit-identity of the path');
    require score agreement to 1e-3 relative and near-total base identity."""
    for name in ("rgrgr_r94", "rnnrf_r94"):
        w, om = models[name]
        lens = [1000, 1203, 777, 1500, 640, 901]
        sigs = [sig(n, 300 + i) for i, n in enumerate(lens)]
        calls = eng.basecall(sigs, name)
        same = 0
        for x, c in zip(sigs, calls):
            o = _oracle_call(orc, om, x)
            if o is None:
                assert c is None
                continue
            assert c is not None and c["nblock"] == o["nblock"]
            assert abs(c["score"] - o["score"]) <= 1e-3 * max(1.0, abs(o["score"]))
            same += (c["bases"] == o["bases"])
        assert same >= len(lens) - 1, (name, same)


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


@pytest.mark.parametrize("kw", [
    dict(tempW=1.3, tempb=0.8, use_slip=1, stay_pen=0.3, skip_pen=0.2),       # S1 with the division, slip move
    dict(tempW=0.7, tempb=1.0, use_slip=0, local_pen=1.0, homopolymer=0),     # input scaling only, no homopolymer pass
    dict(min_prob=1e-3, use_slip=1, skip_pen=1.0),
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
    for x, c in list(zip(base, small))[:4]:
        post = eng.posterior(x, "rgrgr_r94", min_prob=p.min_prob, tempW=p.tempW, tempb=p.tempb)
        sc, seq = oracle.decode_transducer(post, p.stay_pen, p.skip_pen, p.local_pen, bool(p.use_slip))
        if p.homopolymer:
            rc, seq = oracle.homopolymer_path(post, seq)
        bases, pos = oracle.overlapper(seq, 1024)
        assert c[0] == bases and np.float32(c[1]) == np.float32(sc)


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
