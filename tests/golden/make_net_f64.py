#!/usr/bin/env python3
"""Independent float64 statements of the reference's raw / events networks at BASELINE dims,
written from /root/reference/src/{layers.c,networks.c,scrappie_matrix.c,util.h} alone (NOT from
oracle/oracle.c), and the fixtures tests/golden/net_f64_*.npz they produce.

Why: layers.c / scrappie_matrix.c cannot be compiled in this image (no cblas.h) and networks.c
needs model headers that are missing from the checkout, so the network rows of SURVEY.md section 8
cannot be pinned against compiled reference code.  This second, independently written statement in
a different language and precision is what stands between "oracle.c misreads layers.c" and green
tests: the CPU suite checks oracle.c against these fixtures and the -m gpu suite checks the HIP
posterior against them DIRECTLY (tests/test_net_f64.py).

Graphs (reference file:line):
  N1  rgrgr_r94  networks.c:250-296   conv -> elu  -> 5 x {affine, GRU B,F,B,F,B} -> softmax -> robustlog
      rgrgr_r10  networks.c:348-394   same with tanh after the convolution, window 19
  N2  rnnrf_r94  networks.c:567-615   same trunk + residual_inplace per layer -> globalnorm
  N3  raw_r94    networks.c:196-247   conv -> tanh -> 2 x {gru_forward, gru_backward -> feedforward2_tanh} -> softmax
  EV  events     networks.c:146-194   window(features, 3, 1) -> 2 x {lstm_forward, lstm_backward -> feedforward2_tanh} -> softmax

Everything is float64 with exact exp / log / tanh; the reference's *semantics* are kept where they
differ from the textbook (each one cited at its line):
  * convolution edges incl. the right-edge quirk (layers.c:172-241), index for index;
  * exp input clamped to +-88.3762626647949 (sse_mathfun.h:211-222), no max-subtraction (layers.c:350);
  * temperatures applied as divisions, tempW/tempb on the softmax INPUT (layers.c:345, scrappie_matrix.c:560);
  * robustlog = log(min_prob + (1 - min_prob) p) (layers.c:85-91: the code, not the doc comment);
  * window() leaves output column 0 zero (layers.c:131-133: int/size_t comparison);
  * the events network's input features come from the reference's OWN nnfeatures.c, compiled as shipped
    (oracle/_ref/libref_features.so): its studentisation uses the 12-bit rsqrtps estimate, which is
    arithmetic, not rounding, so the float64 statement starts from its float32 output.

Fixture content (data only): the input signal / events / features, the expected output on a
subset of columns in full (first 8, last 8, 32 spread) plus, for EVERY column, arg max, max and
sum of squares of the posterior, and a hash of the weights the run used (scrappie_amd.model.
synthetic_model is seeded numpy.RandomState, bit-stable across numpy versions).

Runs only in the build container:  python tests/golden/make_net_f64.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

EXP_HI = 88.3762626647949       # sse_mathfun.h:215
EXP_LO = -88.3762626647949      # sse_mathfun.h:216


def f64(a):
    return np.asarray(a, dtype=np.float64)


def ref_exp(x):
    """expfv semantics (sse_mathfun.h:221-222: input clamped), exact exp otherwise"""
    return np.exp(np.clip(x, EXP_LO, EXP_HI))


def logistic(x):          # util.h:180-183
    return 1.0 / (1.0 + ref_exp(-x))


def tanh_ref(x):          # util.h:185-188: 2 logistic(2x) - 1
    y = logistic(x + x)
    return (y + y) - 1.0


def elu(x):               # util.h:190-198: x >= 0 ? x : exp(x) - 1
    return np.where(x >= 0, x, ref_exp(x) - 1.0)


def convolution(x, W, b, st):
    """layers.c:159-246 for one input feature.  x[N]; W[F][WL] (tap w = row 4w of the reference's
    filter matrix, misc/parse_rgrgr.py:79); returns C[T][F].  Variable names follow nothing but the
    arithmetic: every index below is derived from the cited lines."""
    N = len(x)
    F, WL = W.shape
    padL = (WL - 1) // 2                                  # :172
    padR = WL // 2                                        # :173
    T = (N + st - 1) // st                                # :174
    C = np.tile(b, (T, 1))                                # :185-187
    # left edge (:190-196): output column w/st sees taps padL-w .. WL-1 on x[0 ..]
    for w in range(0, padL, st):
        k0 = padL - w
        C[w // st] += W[:, k0:] @ x[:WL - k0]
    c0 = (padL + st - 1) // st                            # :199 columns done on the left
    shift = c0 * st - padL                                # :203
    nstepC = (WL + st - 1) // st                          # :206
    nstepX = st * nstepC                                  # :207
    # body (:209-224): one strided GEMM per w; column j of it is a FULL window starting at
    # x[shift + w + j nstepX], written to output column c0 + w/st + j nstepC
    for w in range(0, WL, st):
        ncol = (N - shift - w) // nstepX                  # :218
        col0 = c0 + w // st                               # :219
        for j in range(ncol):
            s = shift + w + j * nstepX
            C[col0 + j * nstepC] += W @ x[s:s + WL]
    # right edge (:227-241)
    maxcol = (N - shift) // nstepX                        # :227
    rem = (N - shift) % nstepX                            # :228
    colR = c0 + nstepC * (maxcol - 1) + rem // st + 1     # :229-231
    xR = N - WL + 1                                       # :232
    startR = st - (padL + N - WL) % st - 1                # :234
    for w in range(startR, padR, st):
        nt = WL - (w + 1)                                 # :236 taps 0 .. WL-2-w
        C[colR + w // st] += W[:, :nt] @ x[xR + w:xR + w + nt]
    return C


def gru(xaff, sW, sW2, backward):
    """gru_forward / gru_backward (layers.c:373-470) over gru_step (layers.c:472-527).
    xaff[T][3S] = iW x + b with gate blocks [z | r | candidate] (:511-513);
    sW[2S][S] rows 0..S-1 feed z, S..2S-1 feed r (:505); sW2[S][S] acts on r*h (:514-518)."""
    T = xaff.shape[0]
    S = sW2.shape[0]
    out = np.zeros((T, S))
    h = np.zeros(S)                                       # :399-400 / :447-448: zero initial state
    for t in (range(T - 1, -1, -1) if backward else range(T)):
        g = xaff[t].copy()                                # :503
        g[:2 * S] += sW @ h                               # :505
        zr = logistic(g[:2 * S])                          # :507-509
        z, r = zr[:S], zr[S:]
        hb = tanh_ref(g[2 * S:] + sW2 @ (r * h))          # :514-521
        h = z * h + (1.0 - z) * hb                        # :524-526
        out[t] = h
    return out


def lstm(xaff, sW, p, backward):
    """lstm_forward / lstm_backward (layers.c:673-772) over lstm_step (layers.c:774-832).
    xaff[T][4S] blocks [input | update | forget | output]; peepholes p[3S] = [update | forget | output]."""
    T = xaff.shape[0]
    S = sW.shape[1]
    out = np.zeros((T, S))
    o = np.zeros(S)
    c = np.zeros(S)
    for t in (range(T - 1, -1, -1) if backward else range(T)):
        g = xaff[t] + sW @ o                                              # :807-810
        forget = logistic(g[2 * S:3 * S] + c * p[S:2 * S]) * c            # :816-818
        update = logistic(g[S:2 * S] + c * p[:S]) * tanh_ref(g[:S])       # :820-822
        c = forget + update                                               # :823
        o = logistic(g[3 * S:] + c * p[2 * S:]) * tanh_ref(c)             # :825-830
        out[t] = o
    return out


def window3(feat):
    """window(features, 3, 1) as layers.c:119-146 actually behaves: `w1 <= icol + wh` compares an
    int with a size_t, so for column 0 (w1 = -1) the loop body never runs and the column stays zero;
    for column c >= 1 the loop visits input columns c-1 .. c+2, the fourth lands in the next output
    column's first rows and is overwritten when that column is processed.  Net effect:
    out[c] = [in[c-1] | in[c] | in[c+1] (zero past the end)], out[0] = 0."""
    n, nf = feat.shape
    out = np.zeros((n, 3 * nf))
    for c in range(1, n):
        out[c, :nf] = feat[c - 1]
        out[c, nf:2 * nf] = feat[c]
        if c + 1 < n:
            out[c, 2 * nf:] = feat[c + 1]
    return out


def softmax_temperature(X, W, b, tempW, tempb):
    """softmax_with_temperature (layers.c:340-357); row_normalise_inplace (scrappie_matrix.c:385-407)
    over the nr real rows"""
    X = X / (tempW / tempb)                               # :345
    Cm = X @ W.T + b                                      # :347
    Cm = Cm / tempb                                       # :350
    E = ref_exp(Cm)                                       # :351
    return E / E.sum(axis=1, keepdims=True)               # :352


def robustlog(P, min_prob):                               # layers.c:79-94
    return np.log(min_prob + (1.0 - min_prob) * P)


def globalnorm(X, W, b):
    """globalnorm (layers.c:874-889) over crf_partition_function (layers.c:835-871)"""
    Cm = X @ W.T + b
    T = Cm.shape[0]
    n = int(round(np.sqrt(Cm.shape[1])))
    prev = np.zeros(n)                                    # calloc: :839
    for t in range(T):
        prev = np.array([np.logaddexp.reduce(Cm[t, s * n:(s + 1) * n] + prev) for s in range(n)])   # :853-858
    logZ = np.logaddexp.reduce(prev)                      # :861-864
    return Cm - logZ / T                                  # :878-885


def net_rgrgr(w, x, min_prob=1e-5, tempW=1.0, tempb=1.0):
    """N1 / N2; returns (trunk top [T][S], output)"""
    act = convolution(f64(x), f64(w["conv_W"]), f64(w["conv_b"]), int(w["stride"]))
    act = elu(act) if w["conv_act"] == "elu" else tanh_ref(act)      # networks.c:260 / :358
    for l in range(5):
        iW, sW, sW2, b = (f64(w["gru%d_%s" % (l, k)]) for k in ("iW", "sW", "sW2", "b"))
        out = gru(act @ iW.T + b, sW, sW2, backward=(l % 2 == 0))   # B1 F2 B3 F4 B5
        act = out + act if w["arch"] == "rnnrf" else out             # networks.c:583-607
    if w["arch"] == "rnnrf":
        return act, globalnorm(act, f64(w["ff_W"]), f64(w["ff_b"]))
    P = softmax_temperature(act, f64(w["ff_W"]), f64(w["ff_b"]), tempW, tempb)
    return act, robustlog(P, min_prob)


def net_raw(w, x, min_prob=1e-5):
    """N3 networks.c:196-247: gru0..3 = F1, B1, F2, B2; ff1 / ff2 = feedforward2_tanh (layers.c:359-371)"""
    act = tanh_ref(convolution(f64(x), f64(w["conv_W"]), f64(w["conv_b"]), int(w["stride"])))
    for lvl in range(2):
        hs = []
        for d in range(2):
            l = 2 * lvl + d
            iW, sW, sW2, b = (f64(w["gru%d_%s" % (l, k)]) for k in ("iW", "sW", "sW2", "b"))
            hs.append(gru(act @ iW.T + b, sW, sW2, backward=(d == 1)))
        k = "ff%d" % (lvl + 1)
        act = tanh_ref(hs[0] @ f64(w[k + "_Wf"]).T + hs[1] @ f64(w[k + "_Wb"]).T + f64(w[k + "_b"]))
    P = softmax_temperature(act, f64(w["ff_W"]), f64(w["ff_b"]), 1.0, 1.0)
    return act, robustlog(P, min_prob)


def net_events(w, feat, min_prob=1e-5):
    """networks.c:146-194 from the studentised features [nevent][4] on"""
    act = window3(f64(feat))
    for lvl in range(2):
        hs = []
        for d in range(2):
            l = 2 * lvl + d
            iW, sW, b, p = (f64(w["lstm%d_%s" % (l, k)]) for k in ("iW", "sW", "b", "p"))
            hs.append(lstm(act @ iW.T + b, sW, p, backward=(d == 1)))
        k = "ff%d" % (lvl + 1)
        act = tanh_ref(hs[0] @ f64(w[k + "_Wf"]).T + hs[1] @ f64(w[k + "_Wb"]).T + f64(w[k + "_b"]))
    P = softmax_temperature(act, f64(w["ff_W"]), f64(w["ff_b"]), 1.0, 1.0)
    return act, robustlog(P, min_prob)


# ---------------------------------------------------------------------------------------------
def weights_hash(w):
    from scrappie_amd import model
    h = hashlib.sha256()
    for nm in model.matrix_names(w):
        h.update(np.ascontiguousarray(w[nm], dtype=np.float32).tobytes())
    return h.hexdigest()


def sample_columns(T):
    cols = set(range(min(8, T))) | set(range(max(T - 8, 0), T)) | set(int(v) for v in np.linspace(0, T - 1, 32))
    return np.array(sorted(cols), dtype=np.int32)


def pack(out, top, extra):
    """out[T][NS] float64 log-posterior (or transitions) -> fixture fields"""
    d = dict(extra)
    T = out.shape[0]
    d["T"] = np.int32(T)
    if out.shape[1] <= 32:                       # rnnrf: the whole thing is small
        d["out"] = out.astype(np.float32)
    else:
        cols = sample_columns(T)
        d["cols"] = cols
        d["out_cols"] = out[cols].astype(np.float32)
        p = np.exp(out)
        d["argmax"] = np.argmax(out, axis=1).astype(np.int16)
        d["max"] = out.max(axis=1).astype(np.float32)
        d["sumsq_p"] = (p * p).sum(axis=1).astype(np.float32)
    d["top_cols"] = top[sample_columns(T)].astype(np.float32)     # trunk output on the same columns
    return d


# (fixture name, model, seed, input length): for each graph one Q1-free and one Q1-hit length
# (window 11, stride 5: hit iff N % 5 != 0; window 19: hit iff N % 5 == 0 -- SURVEY section 8 C1)
RAW_CASES = [
    ("rgrgr_r94_4000", "rgrgr_r94", 4000), ("rgrgr_r94_3998", "rgrgr_r94", 3998),
    ("rgrgr_r10_4000", "rgrgr_r10", 4000), ("rgrgr_r10_3998", "rgrgr_r10", 3998),
    ("rnnrf_r94_4000", "rnnrf_r94", 4000), ("rnnrf_r94_3997", "rnnrf_r94", 3997),
    ("raw_r94_4000", "raw_r94", 4000), ("raw_r94_3999", "raw_r94", 3999),
]
EVENT_CASES = [("events_800", 800), ("events_803", 803)]
MODEL_SEED = 1          # the seed bench.py and the other tests use


def main():
    from scrappie_amd import model, synth
    import oracle
    for i, (fx, name, N) in enumerate(RAW_CASES):
        w = model.synthetic_model(name, seed=MODEL_SEED)
        x = synth.medmad_normalise(synth.synthetic_signal(N, 100 + i))
        extra = {"x": x, "model": name, "model_seed": np.int32(MODEL_SEED), "weights_sha256": weights_hash(w),
                 "min_prob": np.float32(1e-5)}
        if w["arch"] == "raw":
            top, out = net_raw(w, x)
        else:
            top, out = net_rgrgr(w, x)
        d = pack(out, top, extra)
        if fx == "rgrgr_r94_4000":      # temperatures: only the output layer changes
            _, out_t = net_rgrgr(w, x, tempW=1.5, tempb=0.8)
            d["temp"] = np.array([1.5, 0.8], dtype=np.float32)
            d["out_temp_cols"] = out_t[d["cols"]].astype(np.float32)
        np.savez_compressed(os.path.join(HERE, "net_f64_%s.npz" % fx), **d)
        print("%-18s T=%d  max logp %.4f  mean max p %.4f" % (fx, out.shape[0], out.max(), np.exp(out.max(axis=1)).mean()))
    # events: features from the reference's own nnfeatures.c (compiled as shipped)
    import ctypes as C
    rf = oracle.ref_features()
    PM = C.POINTER(oracle.Mat)
    rf.nanonet_features_from_events.restype = PM
    rf.nanonet_features_from_events.argtypes = [oracle.EventTable, C.c_bool]
    rf.free_scrappie_matrix.restype = PM
    rf.free_scrappie_matrix.argtypes = [PM]
    for i, (fx, n) in enumerate(EVENT_CASES):
        w = model.synthetic_model("nanonet_events", seed=MODEL_SEED)
        ev = synth.synthetic_events(n, 200 + i)
        feat = oracle.features_from_events(ev, True, fn=rf.nanonet_features_from_events, free=rf.free_scrappie_matrix)
        assert feat.shape == (n, 4) and feat.dtype == np.float32
        top, out = net_events(w, feat)
        d = pack(out, top, {"events": ev, "features": feat, "feature3": window3(feat).astype(np.float32),
                            "model": "nanonet_events", "model_seed": np.int32(MODEL_SEED),
                            "weights_sha256": weights_hash(w), "min_prob": np.float32(1e-5)})
        np.savez_compressed(os.path.join(HERE, "net_f64_%s.npz" % fx), **d)
        print("%-18s T=%d  max logp %.4f" % (fx, out.shape[0], out.max()))
    sys.path.insert(0, HERE)
    import provenance
    provenance.write()          # sha256 of every fixture and of the reference files it derives from


if __name__ == "__main__":
    main()
