"""GPU tests added in round 6 (run with -m gpu on an MI355X), all through the C ABI:

  * the REFERENCE'S OWN literal vectors for the network rows, pushed through the KERNELS -- until now only the oracle saw them
    (tests/test_oracle_golden.py): src/test/test_scrappie_convolution.c:387-408 (convolution() with a unit filter), test_scrappie_elu.c:23-71
    (elufv), test_scrappie_matrix.c:24-51 (row_normalise_inplace on all-ones rows of 8 ... 11 states);
  * the worst cases of the split products: one recurrent layer on operands chosen to hurt -- dense 24-bit mantissas, operands at
    the range limits, cancelling sums, activations whose low piece is an fp16 subnormal -- against float64 and against the exact-fp32 layer;
  * concurrent small scrappie_hip_basecall_batch calls share launch groups and return what they return alone;
  * the helper engine is sized from the memory that is free when it is made.
"""
import ctypes as C
import threading

import numpy as np
import pytest

import scrappie_amd as sa
from scrappie_amd import model, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = sa.Engine(0)
    yield e
    e.close()


def sig(n, seed):
    return synth.medmad_normalise(synth.synthetic_signal(n, seed))


def _unit_filter_model(act_model, S=32, nstate=65):
    """F = S one-tap filters of weight 1, bias 0, stride 1: convolution() is the identity on every filter row."""
    w = model.synthetic_model(act_model, seed=3, size=S, winlen=1, stride=1, nstate=nstate)
    w["conv_W"] = np.ones((S, 1), np.float32)
    w["conv_b"] = np.zeros(S, np.float32)
    return w


# ------------------------------------------------------------------ item 4: the reference's literals through the kernels
def test_reference_convolution_literal_through_kernel(eng):
    """src/test/test_scrappie_convolution.c:387-408 (test_scrappie_convolution_f1s1): convolution(x, one-tap filter {1.0}, bias 0, stride 1)
    == x to 1e-5 for xrange_odd = 0 .. 10 and xrange_even = 0 .. 9 (:16-22).  Here through k_conv_act (layers.c:159-246 + elu: the
    identity on x >= 0), every one of the 32 filter rows."""
    eng.load_model("unit_elu", _unit_filter_model("rgrgr_r94"))
    for n in (11, 10):
        x = np.arange(n, dtype=np.float32)
        got = eng.trunk(x, "unit_elu", 0)
        assert got.shape == (n, 32)
        assert np.max(np.abs(got - x[:, None])) <= 1e-5, (n, got[:, 0])
        assert np.array_equal(got, np.repeat(x[:, None], 32, axis=1))        # (in fact exact: one product by 1.0, one addition of 0.0)


def test_reference_elu_literals_through_kernel(eng):
    """src/test/test_scrappie_elu.c:23-71: elufv(0) == elufv(-0) == 0 (compared with ==), elufv(1 .. 4) == 1 .. 4 exactly,
    elufv(-1 .. -4) = -0.6321206, -0.8646647, -0.9502129, -0.9816844 to 1e-6 -- through the convolution kernel's activation
    (unit filter, zero bias: the convolution hands its input to elu unchanged)."""
    if "unit_elu" not in eng._models:
        eng.load_model("unit_elu", _unit_filter_model("rgrgr_r94"))
    x = np.array([0.0, -0.0, 1.0, 2.0, 3.0, 4.0, -1.0, -2.0, -3.0, -4.0, 1.0, -2.0, 3.0, -4.0], np.float32)
    got = eng.trunk(x, "unit_elu", 0)[:, 0]
    assert got[0] == 0.0 and got[1] == 0.0
    assert np.array_equal(got[2:6], np.array([1, 2, 3, 4], np.float32))
    want = np.array([-0.6321206, -0.8646647, -0.9502129, -0.9816844], np.float32)
    assert np.max(np.abs(got[6:10] - want)) < 1e-6, got[6:10]
    assert got[10] == 1.0 and got[12] == 3.0 and abs(got[11] - want[1]) < 1e-6 and abs(got[13] - want[3]) < 1e-6      # test_mixed_elu


@pytest.mark.parametrize("nr", [8, 9, 10, 11])
def test_reference_row_normalise_literals_through_kernel(eng, nr):
    """src/test/test_scrappie_matrix.c:24-51: row_normalise_inplace of an all-ones column of nr = 8, 9, 10, 11 rows (the padding of the last
    vector is masked out of the sum) gives 1 / nr to 1e-5.  Here an output layer of nr states with zero weights and biases -- every
    exp is 1, whatever the trunk says -- through S1 (k_ff_lds: 16-row tiles masked at nr) and the posterior surface; and the log form
    log(min_prob + (1 - min_prob) / nr) of S2 (layers.c:79-94)."""
    w = model.synthetic_model("rgrgr_r94", seed=5, size=32, nstate=nr)
    w["ff_W"] = np.zeros((nr, 32), np.float32)
    w["ff_b"] = np.zeros(nr, np.float32)
    name = "ones%d" % nr
    eng.load_model(name, w)
    x = sig(600, 70 + nr)
    p = eng.posterior(x, name, log=False)
    assert p.shape == (120, nr)
    assert np.max(np.abs(p - 1.0 / nr)) <= 1e-5
    lp = eng.posterior(x, name, min_prob=1e-5, log=True)
    assert np.max(np.abs(lp - np.log(1e-5 + (1 - 1e-5) / nr))) <= 1e-5
    # ... and a model of that many states is not decodable (decode.c:132-138): refused, loudly
    with pytest.raises(RuntimeError, match="state count"):
        eng.basecall([x], name)


# ------------------------------------------------------------------ item 3: worst cases of the split products
def _dense_mantissa(rng, shape, lo_exp, hi_exp):
    """floats with all 24 significant bits set to random values, the LAST one forced to 1 (nothing for a 22-bit cut to lose for free),
    magnitudes 2^lo_exp .. 2^hi_exp, random signs"""
    man = rng.integers(1 << 23, 1 << 24, size=shape, dtype=np.int64) | 1
    ex = rng.integers(lo_exp, hi_exp, size=shape)
    sgn = rng.choice([-1.0, 1.0], size=shape)
    return (sgn * np.ldexp(man.astype(np.float64), ex - 23)).astype(np.float32)


def _gru_layer_f64(x, iW, b, sW, sW2, backward):
    """one recurrent layer in float64 (layers.c:248, :373-527: gate order z, r, candidate; sW columns j < S feed z, j >= S feed r)"""
    T, S = x.shape[0], sW2.shape[0]
    xa = x.astype(np.float64) @ iW.astype(np.float64).T + b.astype(np.float64)
    W1, W2 = sW.astype(np.float64), sW2.astype(np.float64)
    h = np.zeros(S)
    out = np.zeros((T, S))
    for t in (range(T - 1, -1, -1) if backward else range(T)):
        zr = xa[t, :2 * S] + W1 @ h
        with np.errstate(over="ignore"):             # (saturated gates: exp overflows to inf, 1 / inf = 0 -- as intended)
            z, r = 1 / (1 + np.exp(-zr[:S])), 1 / (1 + np.exp(-zr[S:]))
        c = np.tanh(xa[t, 2 * S:] + W2 @ (r * h))
        h = z * h + (1 - z) * c
        out[t] = h
    return out


WORST = {
    # (a) every operand with a dense 24-bit mantissa (the cut keeps 22 bits: this is where it shows); magnitudes that keep the gates alive:
    #     input weights 2^-6 .. 2^-3, layer inputs 2^-3 .. 2 -- pre-activations of a few units
    "dense_mantissas": dict(w=lambda r, sh: _dense_mantissa(r, sh, -6, -2), x=lambda r, sh: _dense_mantissa(r, sh, -3, 2)),
    # (b) operands at the limits of the range the engine accepts (|w| < SH_W_LIMIT = 255, |x| < SH_ACT_LIMIT = 1000): pieces at the top of fp16's range
    #     (256 w -> 65 254, 64 x -> 63 936; fp16's largest finite value is 65 504).  Both at once saturates every gate (sums of 1e6) -- a finiteness
    #     check; one at a time keeps the gates alive
    "range_limits": dict(w=lambda r, sh: (r.choice([-1, 1], size=sh) * r.uniform(200.0, 254.9, size=sh)).astype(np.float32),
                         x=lambda r, sh: (r.choice([-1, 1], size=sh) * r.uniform(900.0, 999.0, size=sh)).astype(np.float32)),
    "largest_weights": dict(w="big_iw", x=lambda r, sh: _dense_mantissa(r, sh, -11, -8)),
    "largest_inputs": dict(w="small_iw", x=lambda r, sh: (r.choice([-1, 1], size=sh) * r.uniform(900.0, 999.0, size=sh)).astype(np.float32)),
    # (c) sums that cancel to ~1e-4 of their terms: every input weight row is (v, -v) pairs up to a 1e-4 perturbation, inputs pairwise equal
    "cancelling_sums": dict(w="cancel", x="pairs"),
    # (d) activations so small that the low piece (x * 64 - fp16(x * 64)) is an fp16 subnormal or zero: |x| < 2e-3
    "subnormal_low_piece": dict(w=lambda r, sh: _dense_mantissa(r, sh, -3, 2), x=lambda r, sh: _dense_mantissa(r, sh, -22, -9)),
}
# how much worse than the exact-fp32 layer (8 fp32 MFMAs per k step, fp32 FMA chains) the split-product layer (3 f16 MFMAs on 22-bit operand
# cuts) may be against float64, per case: k x the exact layer's own error + an absolute floor of 2e-7 (the fp32 rounding of the output itself).
# Measured on MI355X (profiles/r6_split_worst_case.txt): the ratios of the maxima are 0.75 ... 2.3.
SPLIT_K = 4.0


@pytest.mark.parametrize("case", list(WORST))
def test_split_products_worst_case(eng, case):
    """The contractions of a recurrent layer run as three f16 partial products of two-piece fp16 cuts (22 of 24 mantissa bits) of both
    operands, accumulated in fp32 (sh_kernels.h: split_pair / split_dot), where the reference calls cblas_sgemm / cblas_sgemv on fp32
    (scrappie_matrix.c:346, layers.c:505,517).  One layer (scrappie_hip_trunk, upto = 1: convolution as the identity, then projection +
    recurrence, backward as networks.c:262) on adversarial operands, against a float64 statement of the layer AND against the same layer
    on the exact-fp32 kernels (debug option force_f32_layers: v_mfma_f32_16x16x4_f32 throughout): the split products' error against
    float64 must stay within SPLIT_K x the exact-fp32 layer's."""
    S, T = 96, 64
    rng = np.random.default_rng(600 + list(WORST).index(case))
    spec = WORST[case]
    w = model.synthetic_model("rgrgr_r94", seed=9, size=S, winlen=1, stride=1, nstate=65)
    w["conv_W"] = np.ones((S, 1), np.float32)                     # the layer's input column t is x[t] on every filter row ...
    w["conv_b"] = np.zeros(S, np.float32)
    if spec["x"] == "pairs":
        xs = np.abs(_dense_mantissa(rng, (T,), -2, 3))
    else:
        xs = np.abs(spec["x"](rng, (T,)))                        # (>= 0: elu is the identity there)
    # ... so per-unit variety comes from the input weights: column f of iW meets x[t] for every f
    if spec["w"] == "cancel":
        half = _dense_mantissa(rng, (3 * S, S // 2), -4, 1)
        iW = np.empty((3 * S, S), np.float32)
        iW[:, 0::2] = half
        iW[:, 1::2] = -half * (1 + 1e-4 * rng.standard_normal(half.shape)).astype(np.float32)
        sW, sW2 = _dense_mantissa(rng, (2 * S, S), -6, -1), _dense_mantissa(rng, (S, S), -6, -1)
    elif spec["w"] == "big_iw":          # input weights at the limit, inputs of ~1e-3: products of ~0.5, sums of a few units
        iW = (rng.choice([-1, 1], size=(3 * S, S)) * rng.uniform(200.0, 254.9, size=(3 * S, S))).astype(np.float32)
        sW, sW2 = _dense_mantissa(rng, (2 * S, S), -6, -1), _dense_mantissa(rng, (S, S), -6, -1)
    elif spec["w"] == "small_iw":        # inputs at the limit, input weights of ~2e-4
        iW = _dense_mantissa(rng, (3 * S, S), -14, -11)
        sW, sW2 = _dense_mantissa(rng, (2 * S, S), -6, -1), _dense_mantissa(rng, (S, S), -6, -1)
    else:
        iW = spec["w"](rng, (3 * S, S))
        if case == "range_limits":
            sW, sW2 = spec["w"](rng, (2 * S, S)), spec["w"](rng, (S, S))
        else:
            # recurrent weights of 2^-6 .. 2^-2 everywhere else: with |w| of a few units a 96-unit recurrence is CHAOTIC -- two correct arithmetics
            # part ways within a few steps (measured: both layer forms 2.0 away from float64), and an error bound says nothing
            sW, sW2 = _dense_mantissa(rng, (2 * S, S), -6, -1), _dense_mantissa(rng, (S, S), -6, -1)
    w["gru0_iW"], w["gru0_sW"], w["gru0_sW2"] = iW.astype(np.float32), sW.astype(np.float32), sW2.astype(np.float32)
    w["gru0_b"] = _dense_mantissa(rng, (3 * S,), -4, 0)
    split_name, exact_name = "worst_" + case, "worst_" + case + "_f32"
    eng.load_model(split_name, w)
    eng.debug_option("force_f32_layers", 1)
    eng.load_model(exact_name, w)
    eng.debug_option("force_f32_layers", 0)
    x_in = np.repeat(xs[:, None], S, axis=1).astype(np.float32)  # what layer 1 sees: [T][F]
    conv = eng.trunk(xs.astype(np.float32), split_name, 0)
    assert np.array_equal(conv, x_in)                             # the identity convolution really is one
    want = _gru_layer_f64(x_in, w["gru0_iW"], w["gru0_b"], w["gru0_sW"], w["gru0_sW2"], backward=True)
    got_split = eng.trunk(xs.astype(np.float32), split_name, 1).astype(np.float64)
    got_exact = eng.trunk(xs.astype(np.float32), exact_name, 1).astype(np.float64)
    e_split, e_exact = np.abs(got_split - want), np.abs(got_exact - want)
    print("split products, %s: max |err| vs float64 %.3g (rms %.3g); exact-fp32 layer %.3g (rms %.3g); ratio of maxima %.2f" %
          (case, e_split.max(), np.sqrt((e_split ** 2).mean()), e_exact.max(), np.sqrt((e_exact ** 2).mean()), e_split.max() / max(e_exact.max(), 1e-30)))
    assert np.all(np.isfinite(got_split))
    assert e_split.max() <= SPLIT_K * e_exact.max() + 2e-7, (case, e_split.max(), e_exact.max())
    assert np.sqrt((e_split ** 2).mean()) <= SPLIT_K * np.sqrt((e_exact ** 2).mean()) + 5e-8


# ------------------------------------------------------------------ item 6: small batch calls from many threads
def test_small_batch_calls_from_many_threads_share_launch_groups(eng):
    """BASELINE config 2 as written ('batch=64'): the reference's loop shape (#pragma omp parallel for schedule(dynamic),
    scrappie_raw.c:355,387) with a body that hands 64 reads to scrappie_hip_basecall_batch.  32 threads x 3 calls of 64 (ragged) reads on
    one engine: every call returns exactly what the same reads return in ONE call of them all, and the calls did share engine calls."""
    w = model.synthetic_model("rgrgr_r94", seed=11)
    eng.load_model("r94_b64", w)
    L = sa.lib()
    nthr, per, ncall = 32, 64, 3
    sigs = [sig(900 + 37 * (i % 23), 5000 + i) for i in range(nthr * ncall * per)]
    key = lambda c: None if c is None else (c["bases"], c["score"], c["nblock"])
    whole = [key(c) for c in eng.basecall(sigs, "r94_b64")]
    st0 = (C.c_ulonglong * 3)()
    L.scrappie_hip_batch_coalescer_stats(st0)
    got = [None] * len(sigs)
    errs = []

    def body(t):
        try:
            for k in range(ncall):
                lo = (t * ncall + k) * per
                got[lo:lo + per] = [key(c) for c in eng.basecall(sigs[lo:lo + per], "r94_b64")]
        except Exception as ex:                      # noqa: BLE001
            errs.append(repr(ex))
    th = [threading.Thread(target=body, args=(t,)) for t in range(nthr)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs[:3]
    assert got == whole
    st1 = (C.c_ulonglong * 3)()
    L.scrappie_hip_batch_coalescer_stats(st1)
    engine_calls, caller_calls, most = st1[0] - st0[0], st1[1] - st0[1], st1[2]
    assert caller_calls == nthr * ncall and engine_calls < caller_calls / 4 and most >= 8, (engine_calls, caller_calls, most)


def test_a_failing_small_batch_call_does_not_take_its_neighbours_down():
    """A call that cannot run (a read longer than a launch group may be: scrappie_hip_set_max_launch_blocks) fails with a message.  When it has joined other
    threads' calls in the queue the shared engine call fails as a whole; the members are then run one by one, so the others still get exactly their calls."""
    w = model.synthetic_model("rgrgr_r94", seed=11)
    e = sa.Engine(0)
    e.load_model("m", w)
    e.set_max_launch_blocks(4000)                             # a tile of 16 reads x 250 blocks: reads of up to 1250 samples fit, 30 000 samples do not
    good = [[sig(1000, 9000 + 8 * t + i) for i in range(8)] for t in range(12)]
    key = lambda c: None if c is None else (c["bases"], c["score"], c["nblock"])
    alone = [[key(c) for c in e.basecall(g, "m")] for g in good]
    bad = [sig(30000, 9999)]
    with pytest.raises(RuntimeError, match="too long"):
        e.basecall(bad, "m")
    res, errs = [None] * 13, [None] * 13

    def body(t):
        try:
            res[t] = [key(c) for c in e.basecall(good[t] if t < 12 else bad, "m")]
        except RuntimeError as ex:
            errs[t] = str(ex)
    for rep in range(3):
        th = [threading.Thread(target=body, args=(t,)) for t in range(13)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert res[:12] == alone and errs[:12] == [None] * 12, (rep, errs)
        assert errs[12] is not None and "too long" in errs[12]
    e.close()


# ------------------------------------------------------------------ the helper engine takes what is free
def test_helper_engine_is_sized_from_free_memory():
    """A call with chain-bound reads makes the helper engine; its arena is min(0.30, 0.85 x what is free THEN) of the device -- with next
    to nothing free (debug option tail_free_frac, in 1/1000) the call is simply not split and runs whole on the main engine, with the same
    calls either way."""
    w = model.synthetic_model("rgrgr_r94", seed=11)
    lens = [1000] * 300 + [150000]
    sigs = [sig(n, 8100 + i) for i, n in enumerate(lens)]
    key = lambda c: None if c is None else (c["bases"], c["score"], c["nblock"])
    res = []
    for free in (0, 100, 5):                         # 0: no hook (whatever is free); 100: 10 % free; 5: 0.5 % -- below the 3 % a helper needs
        e = sa.Engine(0)
        e.load_model("m", w)
        if free:
            e.debug_option("tail_free_frac", free)
        res.append([key(c) for c in e.basecall(sigs, "m")])
        e.close()
    assert res[0] == res[1] == res[2]
