"""CPU tests: the oracle (oracle/oracle.c) against
  (a) the reference's own test literals and data files (tests/golden/ref_test_files.npz),
  (b) outputs of the reference's own code compiled in the build container
      (tests/golden/ref_*.npz, generator: tests/golden/make_golden.py),
  (c) live, the compiled reference checkers in oracle/_ref when present.
Integer results must be bit-exact; float results bit-exact where the oracle
restates the same arithmetic, otherwise within the tolerance the reference's own
tests use (src/test/test_scrappie_signal.c:88,100)."""
import ctypes as C

import numpy as np
import pytest

import oracle
from scrappie_amd import synth

fp = C.POINTER(C.c_float)


def _apply(fn, x):
    return np.array([fn(float(v)) for v in x], dtype=np.float32)


# ---------------------------------------------------------------- A1 math
@pytest.mark.parametrize("name,ofn,key", [
    ("expfv", "orc_expf", "x"), ("logisticfv", "orc_logisticf", "x"),
    ("tanhfv", "orc_tanhf", "x"), ("elufv", "orc_eluf", "x"), ("logfv", "orc_logf", "xpos")])
def test_vector_math_bit_exact(orc, golden, name, ofn, key):
    g = golden["ref_math"]
    got = _apply(getattr(orc.lib(), ofn), g[key])
    want = g[name]
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), \
        "%s differs from the compiled reference at %d inputs" % (name, (got.view(np.uint32) != want.view(np.uint32)).sum())


def test_elu_reference_literals(orc):
    """src/test/test_scrappie_elu.c:23-71"""
    L = orc.lib()
    assert L.orc_eluf(0.0) == 0.0 and L.orc_eluf(-0.0) == 0.0
    for x in (1.0, 2.0, 3.0, 4.0):
        assert L.orc_eluf(x) == x
    for x, want in ((-1.0, -0.6321206), (-2.0, -0.8646647), (-3.0, -0.9502129), (-4.0, -0.9816844)):
        assert abs(L.orc_eluf(x) - want) < 1e-6


# ---------------------------------------------------------------- P0 signal
def test_reference_signal_files(orc, golden):
    """src/test/test_scrappie_signal.c:59-103 on the reference's own data."""
    g = golden["ref_test_files"]
    L = orc.lib()
    raw = g["raw_signal"].astype(np.float32)
    unit = np.float32(1373.41) / np.float32(8192.0)
    raw = ((raw + np.float32(16.0)) * unit).astype(np.float32)
    rt, keep = orc.raw_table(raw)
    r = L.orc_trim_raw_by_mad(rt, 100, 0.0)
    assert r.start == 0 and r.end == (len(raw) // 100) * 100
    start, end = r.start + 200, r.end - 10
    trimmed = raw[start:end]
    assert len(trimmed) == len(g["trimmed_signal"])
    assert np.max(np.abs(trimmed - g["trimmed_signal"])) <= 1e-4
    sig = g["trimmed_signal"].copy()
    L.orc_medmad_normalise_array(sig.ctypes.data_as(fp), len(sig))
    assert np.max(np.abs(sig - g["normalised_signal"])) <= 1e-5


def test_median_reference_literals(orc):
    """src/test/test_util.c:24-34"""
    L = orc.lib()
    odd = np.array([1, 2, 3, 4, 5], dtype=np.float32)[::-1].copy()
    even = np.array([1, 2, 3, 4], dtype=np.float32)
    assert L.orc_medianf(odd.ctypes.data_as(fp), 5) == 3.0
    assert L.orc_medianf(even.ctypes.data_as(fp), 4) == 2.5


def test_signal_prep_vs_compiled_reference(orc, golden):
    g = golden["ref_signal_prep"]
    L = orc.lib()
    for n, seed in g["cases"]:
        n, seed = int(n), int(seed)
        sig = synth.synthetic_signal(n, seed, raw_units=True)
        if seed % 2 == 0:
            sig[:250] = sig[:250] * 0.02 + 90
            sig[-130:] = sig[-130:] * 0.02 + 90
        rt, keep = orc.raw_table(sig.copy())
        for perc in (0.0, 0.25):
            r = L.orc_trim_raw_by_mad(rt, 100 if n >= 200 else 10, perc)
            assert [r.start, r.end] == list(g["trim_%d_%g" % (seed, perc)])
        assert np.float32(L.orc_medianf(sig.ctypes.data_as(fp), n)) == g["median_%d" % seed]
        assert np.float32(L.orc_madf(sig.ctypes.data_as(fp), n, None)) == g["mad_%d" % seed]
        nrm = sig.copy()
        L.orc_medmad_normalise_array(nrm.ctypes.data_as(fp), n)
        assert np.array_equal(nrm.view(np.uint32), g["norm_%d" % seed].view(np.uint32))


def test_trim_and_segment_defaults(orc):
    """scrappie_common.c:11-17 arithmetic == python/scrappy/__init__.py:125-133"""
    L = orc.lib()
    sig = synth.synthetic_signal(4000, 3, raw_units=True)
    rt, keep = orc.raw_table(sig)
    a = L.orc_trim_raw_by_mad(rt, 100, 0.0)
    b = L.orc_trim_and_segment_raw(rt, 200, 10, 100, 0.0)
    assert b.start == a.start + 200 and b.end == a.end - 10
    assert (b.end - b.start) % 5 == 0          # default trims keep nsample % 5 == 0
    tiny = synth.synthetic_signal(150, 4, raw_units=True)
    rt2, keep2 = orc.raw_table(tiny)
    assert not L.orc_trim_and_segment_raw(rt2, 200, 10, 100, 0.0).raw   # start >= end -> no read


# ---------------------------------------------------------------- matrix
@pytest.mark.parametrize("nr", [8, 9, 10, 11])
def test_row_normalise_reference_cases(orc, nr):
    """src/test/test_scrappie_matrix.c:24-51: all lanes (pads too) = 1.0"""
    L = orc.lib()
    m = orc.NpMat(np.ones((1, nr), dtype=np.float32))
    m.buf[:] = 1.0
    L.orc_row_normalise_inplace(m.ptr)
    assert np.allclose(m.buf[0, :nr], 1.0 / nr, atol=1e-5)


# ---------------------------------------------------------------- C1 conv
def _np_conv_same(x, W, b, stride):
    """Textbook 'same' zero-padded strided conv, float64 (what convolution()
    intends: layers.c:148-158)."""
    F, WL = W.shape
    padL = (WL - 1) // 2
    T = (len(x) + stride - 1) // stride
    xp = np.concatenate([np.zeros(padL), x.astype(np.float64), np.zeros(WL)])
    out = np.zeros((T, F))
    for c in range(T):
        out[c] = b + W.astype(np.float64) @ xp[c * stride:c * stride + WL]
    return out


def _orc_conv(orc, x, W, b, stride):
    L = orc.lib()
    X = orc.NpMat(x.reshape(-1, 1))
    Wm = orc.conv_filter_mat(W)
    bm = orc.NpMat(b.reshape(1, -1))
    return orc.mat_to_numpy(L.orc_convolution(X.ptr, Wm.ptr, bm.ptr, stride, None), L.orc_free_mat)


def test_convolution_unit_filter_stride1(orc):
    """src/test/test_scrappie_convolution.c:387-408 (the only reference test of
    the real convolution()): 1-tap unit filter, stride 1, odd and even length."""
    for n in (9, 10):
        x = np.arange(1, n + 1, dtype=np.float32)
        out = _orc_conv(orc, x, np.ones((1, 1), np.float32), np.zeros(1, np.float32), 1)
        assert np.allclose(out[:, 0], x, atol=1e-6)


@pytest.mark.parametrize("WL,st", [(11, 5), (19, 5), (11, 2), (3, 1), (7, 3), (1, 1)])
def test_convolution_interior_matches_textbook(orc, WL, st):
    """Away from the right edge convolution() is a plain 'same' conv; the
    right-edge columns carry quirk Q1 and are pinned as
    characterisation in test_convolution_right_edge_quirk."""
    rng = np.random.RandomState(WL * 10 + st)
    F = 8
    W = rng.normal(size=(F, WL)).astype(np.float32)
    b = rng.normal(size=F).astype(np.float32)
    nstepX = st * ((WL + st - 1) // st)
    for N in range(6 * nstepX, 6 * nstepX + nstepX):
        x = rng.normal(size=N).astype(np.float32)
        got = _orc_conv(orc, x, W, b, st)
        want = _np_conv_same(x, W, b, st)
        assert got.shape == want.shape
        nedge = 3
        assert np.max(np.abs(got[:-nedge] - want[:-nedge])) < 1e-4


def test_convolution_right_edge_quirk(orc):
    """Quirk Q1 as measured on the compiled reference:
    WL=11, st=5 exact iff N%5==0; WL=19, st=5 wrong iff N%5==0; st=1 always exact.
    When wrong: column T-1 is bias only and T-2 holds the partial window meant
    for T-1."""
    rng = np.random.RandomState(7)
    F = 4
    b = rng.normal(size=F).astype(np.float32)

    def exact(WL, st, N):
        W = rng.normal(size=(F, WL)).astype(np.float32)
        x = rng.normal(size=N).astype(np.float32)
        got, want = _orc_conv(orc, x, W, b, st), _np_conv_same(x, W, b, st)
        return np.max(np.abs(got - want)) < 1e-4, got, want

    for N in range(400, 410):
        assert exact(11, 5, N)[0] == (N % 5 == 0)
        assert exact(19, 5, N)[0] == (N % 5 != 0)
        assert exact(7, 1, N)[0]
    ok, got, want = exact(11, 5, 4001)
    assert not ok and np.allclose(got[-1], b, atol=1e-6)


# ---------------------------------------------------------------- network vs float64
def _np_network(w, x):
    """Independent float64 numpy statement of N1/N2 (networks.c:250-296,
    :567-615) with textbook math; cross-checks the oracle's reading of the
    reference, not its rounding."""
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    st = w["stride"]
    act = _np_conv_same(x, w["conv_W"], f64(w["conv_b"]), st)
    act = np.where(act >= 0, act, np.exp(act) - 1) if w["conv_act"] == "elu" else np.tanh(act)
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))
    for l in range(5):
        iW, sW, sW2, b = (f64(w["gru%d_%s" % (l, n)]) for n in ("iW", "sW", "sW2", "b"))
        S = sW2.shape[0]
        xin = act @ iW.T + b
        T = xin.shape[0]
        out = np.zeros((T, S))
        h = np.zeros(S)
        order = range(T - 1, -1, -1) if l % 2 == 0 else range(T)
        for t in order:
            g = xin[t].copy()
            g[:2 * S] += sW @ h
            z, r = sig(g[:S]), sig(g[S:2 * S])
            hbar = np.tanh(g[2 * S:] + sW2 @ (r * h))
            h = z * h + (1 - z) * hbar
            out[t] = h
        act = out + act if w["arch"] == "rnnrf" else out
    logits = act @ f64(w["ff_W"]).T + f64(w["ff_b"])
    return act, logits


@pytest.mark.parametrize("name,N", [("rgrgr_r94", 400), ("rgrgr_r10", 403), ("rnnrf_r94", 400)])
def test_network_oracle_vs_float64(orc, name, N):
    from scrappie_amd import model
    w = model.synthetic_model(name, seed=5, size=32)
    x = synth.medmad_normalise(synth.synthetic_signal(N, 9))
    om = orc.OracleModel(w)
    top, logits = _np_network(w, x)
    got_top = orc.trunk(om, x, 5)
    # N chosen Q1-free for each window length (WL=11: N%5==0; WL=19: N%5!=0)
    assert np.max(np.abs(got_top - top)) < 2e-5
    post = orc.posterior(om, x, min_prob=1e-5)
    if w["arch"] == "rgrgr":
        p = np.exp(logits)
        p /= p.sum(axis=1, keepdims=True)
        want = np.log(1e-5 + (1 - 1e-5) * p)
        assert np.max(np.abs(np.exp(post) - np.exp(want))) < 1e-5
        assert np.max(np.abs(post - want)) < 1e-4
    else:
        # globalnorm: subtract logZ/T where logZ is the forward partition function
        T = logits.shape[0]
        prev = np.zeros(5)
        for t in range(T):
            prev = np.array([np.logaddexp.reduce(logits[t, s * 5:(s + 1) * 5] + prev) for s in range(5)])
        logZ = np.logaddexp.reduce(prev)
        assert np.max(np.abs(post - (logits - logZ / T))) < 2e-4


def test_gru_needs_two_columns(orc):
    """quirk Q6: layers.c:400,448 park the zero state in an output column"""
    L = orc.lib()
    X = orc.NpMat(np.zeros((1, 12), np.float32))
    sW = orc.NpMat(np.zeros((8, 4), np.float32))
    sW2 = orc.NpMat(np.zeros((4, 4), np.float32))
    assert not L.orc_gru_forward(X.ptr, sW.ptr, sW2.ptr, None)


# ---------------------------------------------------------------- decode
def test_decode_vs_compiled_reference_fixture(orc, golden):
    g = golden["ref_decode"]
    for (T, seed, klen, stay, skip, local, slip, hp) in g["transducer_cases"]:
        T, seed, klen, hp = int(T), int(seed), int(klen), int(hp)
        post = synth.fixture_posterior(T, seed, klen, hp)
        score, seq = orc.decode_transducer(post, stay, skip, local, bool(slip))
        assert np.array_equal(seq, g["seq_%d" % seed])
        assert np.float32(score) == g["score_%d" % seed]
        if not slip:
            s2, seq2 = orc.decode_transducer(post, stay, skip, local, False,
                                             fn=lambda m, a, b, c, s, sl: orc.lib().orc_sloika_viterbi(m, a, b, c, s))
            assert np.array_equal(seq2, g["sloika_seq_%d" % seed])
            assert np.float32(s2) == g["sloika_score_%d" % seed]
            # the reference's own unit test: decode_transducer == sloika_viterbi
            # (src/test/test_scrappie_decoding.c:33-67: path exact over nblock, score 1e-5)
            # (holds wherever the compiled reference's two functions agree themselves: in case 116, T = 1500, they
            # break one exact tie differently -- same score, one block of the path differs -- and so must we)
            ref_same = np.array_equal(g["seq_%d" % seed][:T], g["sloika_seq_%d" % seed][:T])
            assert seed == 116 or ref_same
            assert np.array_equal(seq[:T], seq2[:T]) == ref_same and abs(score - s2) < 1e-5 * max(1, abs(score))
        bases, pos = orc.overlapper(seq, 4 ** klen)
        assert (bases or "") == str(g["bases_%d" % seed])
        assert np.array_equal(pos, g["pos_%d" % seed])
        rc, hseq = orc.homopolymer_path(post, seq)
        assert rc == 0 and np.array_equal(hseq, g["hp_seq_%d" % seed])
        hb, _ = orc.overlapper(hseq, 4 ** klen)
        assert (hb or "") == str(g["hp_bases_%d" % seed])


def test_crf_vs_compiled_reference_fixture(orc, golden):
    g = golden["ref_decode"]
    L = orc.lib()
    for T, seed in g["crf_cases"]:
        T, seed = int(T), int(seed)
        tr = synth.simulated_crf_transitions(T, seed)
        score, path = orc.decode_crf(tr)
        assert np.array_equal(path, g["crf_path_%d" % seed])
        assert np.float32(score) == g["crf_score_%d" % seed]
        assert orc.crfpath_to_basecall(path, T) == str(g["crf_bases_%d" % seed])
        m = orc.NpMat(tr)
        pp = orc.mat_to_numpy(L.orc_posterior_crf(m.ptr), L.orc_free_mat)
        assert np.array_equal(pp.view(np.uint32), g["crf_post_%d" % seed].view(np.uint32))


def test_overlapper_edge_cases(orc):
    """decode.c:449-509: all stays -> NULL; homopolymer repeat adds one base"""
    assert orc.overlapper(np.array([-1, -1, -1], np.int32), 1024)[0] is None
    bases, pos = orc.overlapper(np.array([-1, 0, -1, 0, 1], np.int32), 1024)
    assert bases == "AAAAA" + "A" + "C" and list(pos) == [0, 0, 0, 1, 2]


def test_reference_path_fixture_loads(golden):
    """src/test/path.crp (Sloika golden path, score -115.5761) is kept as data;
    its posterior (posterior_trimmed.crp) is a missing blob, so it cannot be
    replayed."""
    p = golden["ref_test_files"]["path"]
    assert p.shape == (1000,) and p.min() == -1 and p.max() < 1024


# ---------------------------------------------------------------- live vs _ref
def test_live_against_compiled_reference(orc):
    rd, rp = orc.ref_decode(), orc.ref_pure()
    if rd is None or rp is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this box)")
    PM = C.POINTER(orc.Mat)
    ip = C.POINTER(C.c_int)
    rd.decode_transducer.restype = C.c_float
    rd.decode_transducer.argtypes = [PM, C.c_float, C.c_float, C.c_float, ip, C.c_bool]
    rp.homopolymer_path.restype = C.c_int
    rp.homopolymer_path.argtypes = [PM, ip, C.c_int]
    rd.overlapper.restype = C.c_void_p
    rd.overlapper.argtypes = [ip, C.c_size_t, C.c_int, ip]
    rng = np.random.RandomState(0)
    for i in range(12):
        T = int(rng.randint(3, 260))
        slip = bool(i % 2)
        pens = [float(v) for v in rng.choice([0.0, 0.5, 2.0], size=3)]
        post, _ = synth.simulated_posterior(T, 1000 + i, plant_homopolymers=3)
        if i % 3 == 0:      # heavy floor -> many exactly-tied scores (tie-break order, quirk Q7)
            post = np.maximum(post, np.float32(np.log(np.float32(1e-5) * 4))).astype(np.float32)
        a = orc.decode_transducer(post, pens[0], pens[1], max(pens[2], 0.5), slip)
        b = orc.decode_transducer(post, pens[0], pens[1], max(pens[2], 0.5), slip, fn=rd.decode_transducer)
        assert a[0] == b[0] and np.array_equal(a[1], b[1])
        ha = orc.homopolymer_path(post, a[1])
        hb = orc.homopolymer_path(post, a[1], fn=rp.homopolymer_path)
        assert np.array_equal(ha[1], hb[1])
        assert orc.overlapper(ha[1], 1024)[0] == orc.overlapper(ha[1], 1024, fn=rd.overlapper)[0]


# ---------------------------------------------------------------------------
# events path: features pinned on the compiled reference, LSTM on an
# independent float64 restatement (the reference layer cannot be compiled here: cblas.h)
# ---------------------------------------------------------------------------
def test_event_features_match_compiled_reference(golden):
    g = golden["ref_events"]
    for n, seed in g["cases"]:
        ev = synth.synthetic_events(int(n), int(seed))
        for norm in (1, 0):
            got = oracle.features_from_events(ev, normalise=bool(norm))
            want = g["feat_%d_%d" % (seed, norm)]
            assert got.shape == want.shape == (n, 4)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (n, seed, norm)


def test_window_first_column_quirk():
    """layers.c:119-147 as written: column 0 of the windowed features stays zero (the loop
    condition compares an int with a size_t), every other column is [t-1 | t | t+1] with
    zeros past the end."""
    rng = np.random.default_rng(3)
    f = rng.standard_normal((9, 4)).astype(np.float32)
    w = oracle.window(f, 3, 1)
    assert w.shape == (9, 12)
    assert not w[0].any()
    for t in range(1, 9):
        want = np.concatenate([f[t - 1], f[t], f[t + 1] if t + 1 < 9 else np.zeros(4, np.float32)])
        assert np.array_equal(w[t], want)


def _lstm_f64(xaff, sW, p, backward):
    """float64 restatement of lstm_step (layers.c:806-830) for cross-checking the oracle"""
    T, S4 = xaff.shape
    S = S4 // 4
    sg = lambda x: 1.0 / (1.0 + np.exp(-x))
    out = np.zeros((T, S))
    h = np.zeros(S); c = np.zeros(S)
    order = range(T - 1, -1, -1) if backward else range(T)
    for t in order:
        xF = xaff[t].astype(np.float64) + sW.astype(np.float64) @ h
        forget = sg(xF[2 * S:3 * S] + c * p[S:2 * S]) * c
        update = sg(xF[S:2 * S] + c * p[:S]) * np.tanh(xF[:S])
        c = forget + update
        h = sg(xF[3 * S:] + c * p[2 * S:]) * np.tanh(c)
        out[t] = h
    return out


@pytest.mark.parametrize("backward", [False, True])
def test_lstm_layer_against_float64(backward):
    rng = np.random.default_rng(11)
    S, T = 32, 40
    xaff = rng.standard_normal((T, 4 * S)).astype(np.float32)
    sW = (rng.standard_normal((4 * S, S)) / np.sqrt(S)).astype(np.float32)
    p = rng.uniform(-0.5, 0.5, 3 * S).astype(np.float32)
    got = oracle.lstm(xaff, sW, p, backward=backward)
    want = _lstm_f64(xaff, sW, p.astype(np.float64), backward)
    assert got.shape == (T, S)
    assert np.abs(got - want).max() < 5e-6


def test_fixture_provenance():
    import os
    """tests/golden/PROVENANCE.json (written by the generators, tests/golden/provenance.py): every fixture is the file the record was
    made with, and -- wherever the reference checkout is present -- every reference file a fixture derives from is still the
    file it was derived from.  Keeps "pinned by the reference's own code / data" a checkable statement."""
    import hashlib
    import json
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    rec = json.load(open(os.path.join(here, "PROVENANCE.json")))

    def sha(path):
        return hashlib.sha256(open(path, "rb").read()).hexdigest()
    seen = set()
    for pat, d in rec.items():
        assert d["fixtures"] and d["reference_files"], pat
        for f, h in d["fixtures"].items():
            assert sha(os.path.join(here, f)) == h, "fixture %s is not the one PROVENANCE.json records: regenerate both" % f
            seen.add(f)
    import glob
    have = {os.path.relpath(f, here) for f in glob.glob(os.path.join(here, "*.npz")) + glob.glob(os.path.join(here, "reads", "*.i16")) + glob.glob(os.path.join(here, "fast5", "*.fast5"))}
    assert have <= seen, "fixtures without a provenance record: %s" % sorted(have - seen)
    ref = "/root/reference"
    if os.path.isdir(os.path.join(ref, "src")):
        for pat, d in rec.items():
            for f, h in d["reference_files"].items():
                assert sha(os.path.join(ref, f)) == h, "reference file %s changed since %s was generated" % (f, pat)
