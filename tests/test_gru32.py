"""k_gru_proj32 (sh_gru32.h): the recurrent layers of S = 96 on tiles of 32 reads -- pairs of the launch group's 16-read
tiles, eight specialised waves, v_mfma_f32_32x32x16_f16 -- selected per engine with debug_option("gru32", 1).

Its MFMA shape and product order differ from the 16-read kernels', so its bits do (a model runs ALL its tiles through one
form); what must hold: the oracle / float64 tolerances (tests/test_net_f64.py runs every fixture through both forms), and
the batch properties every form has: a read's call does not depend on its batch, its tile neighbours, the tile its tile is
paired with, or where the lane schedule cuts a pair (state handed over through HBM)."""
import os

import numpy as np
import pytest

import scrappie_amd as sa
from scrappie_amd import model, synth

pytestmark = pytest.mark.gpu
ACT_TOL, P_TOL, CRF_TOL = 2e-5, 1e-5, 1.2e-5

# k_gru_proj32 is not the default form (measured at par with k_gru_proj, DESIGN_NOTES.md section 5) and lives in the experiments build only
# (libscrappie_hip_exp.so).  In an ordinary session this module is ONE test that runs itself -- and the float64-fixture tests for
# the 32-read form -- in a process that loads that library; there the tests below are the tests.
EXP = os.path.abspath(os.environ.get("SCRAPPIE_HIP_LIB", "")) == os.path.abspath(sa.EXP_LIB_PATH)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(EXP, reason="this IS the experiments-library session")
def test_tiles32_suite_under_the_experiments_library():
    import subprocess
    import sys
    assert os.path.exists(sa.EXP_LIB_PATH), "build it: make -C scrappie_amd/csrc ../libscrappie_hip_exp.so"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gru32.py"), os.path.join(ROOT, "tests", "test_net_f64.py"),
                        "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"], cwd=ROOT, env=dict(os.environ, SCRAPPIE_HIP_LIB=sa.EXP_LIB_PATH),
                       capture_output=True, text=True, timeout=1500)
    tail = r.stdout[-1500:]
    assert r.returncode == 0, tail + r.stderr[-1500:]
    assert " passed" in tail and "tiles32" not in r.stdout.split("short test summary")[-1]
    print(tail.strip().splitlines()[-1])


need_exp = pytest.mark.skipif(not EXP, reason="needs the experiments library: run through test_tiles32_suite_under_the_experiments_library")


def sig(n, seed):
    return synth.medmad_normalise(synth.synthetic_signal(n, seed))


@pytest.fixture(scope="module")
def eng32():
    e = sa.Engine(0)
    e.debug_option("gru32", 1)
    yield e
    e.close()


@pytest.fixture(scope="module")
def models32(eng32, orc):
    out = {}
    for name in ("rgrgr_r94", "rnnrf_r94", "raw_r94"):
        w = model.synthetic_model(name, seed=11, size=96)
        eng32.load_model(name, w)
        out[name] = (w, orc.OracleModel(w))
    return out


key = lambda c: None if c is None else (c["bases"], c["score"], c["nblock"])


@need_exp
@pytest.mark.parametrize("name,upto", [("rgrgr_r94", 1), ("rgrgr_r94", 2), ("rgrgr_r94", 5), ("rnnrf_r94", 5), ("raw_r94", 1), ("raw_r94", 2)])
def test_tiles32_trunk_vs_oracle(eng32, orc, models32, name, upto):
    """layer by layer against the oracle (layers.c:373-527; residual networks.c:583; both directions)"""
    w, om = models32[name]
    for N in (1500, 1333):
        x = sig(N, 21 + N)
        got, want = eng32.trunk(x, name, upto), orc.trunk(om, x, upto)
        assert got.shape == want.shape
        d = float(np.max(np.abs(got - want)))
        print("tiles32 %s upto %d N %d: max |d| %.3g" % (name, upto, N, d))
        assert d <= ACT_TOL


@need_exp
def test_tiles32_differs_from_tiles16_only_in_the_last_bits(eng32, models32):
    """the two forms round differently (documented) -- and by no more than rounding"""
    w, _ = models32["rgrgr_r94"]
    e16 = sa.Engine(0)
    try:
        e16.debug_option("gru32", 0)
        e16.load_model("rgrgr_r94", w)
        x = sig(2000, 77)
        a, b = eng32.trunk(x, "rgrgr_r94", 5), e16.trunk(x, "rgrgr_r94", 5)
        d = float(np.max(np.abs(a - b)))
        print("tiles32 vs tiles16 trunk: max |d| %.3g" % d)
        assert d <= 5e-6
        pa, pb = eng32.posterior(x, "rgrgr_r94"), e16.posterior(x, "rgrgr_r94")
        assert np.max(np.abs(np.exp(pa.astype(np.float64)) - np.exp(pb.astype(np.float64)))) <= P_TOL / 4
    finally:
        e16.close()


@need_exp
def test_tiles32_batch_independence_pairing_and_ragged_reads(eng32, orc, models32):
    """mixed lengths, an odd number of tiles (the last tile is stepped alone), reads too short to call, duplicates in different
    tiles and different halves of a pair: every call equals the same signal's call in a tiny batch -- and that one equals
    the oracle's decode of the engine's own posterior"""
    w, om = models32["rgrgr_r94"]
    base = [sig(300 + 37 * (i % 29), 4000 + i) for i in range(61)] + [sig(12, 1), np.zeros(0, np.float32)]
    n = 16 * 45 + 5                                   # 46 tiles, the last one partly filled: 23 pairs
    reads = [base[(i * 17) % len(base)] for i in range(n)]
    p = eng32.default_params(local_pen=150.0)
    whole = [key(c) for c in eng32.basecall(reads, "rgrgr_r94", p)]
    ref = [key(c) for c in eng32.basecall(base, "rgrgr_r94", p)]
    assert all(whole[i] == ref[(i * 17) % len(base)] for i in range(n))
    one = [key(eng32.basecall([r], "rgrgr_r94", p)[0]) for r in base[:6]]          # a pair with an empty second tile
    assert one == ref[:6]
    assert ref[-1] is None and ref[-2] is None
    nb = 0
    for r, k in zip(base[:4], ref[:4]):
        post = eng32.posterior(r, "rgrgr_r94")
        sc, seq = orc.decode_transducer(post, local_pen=150.0)
        rc, seq = orc.homopolymer_path(post, seq)
        wb, wpos = orc.overlapper(seq, 1024)
        assert k[0] == wb
        nb += len(wb)
    assert nb > 0.3 * sum((len(r) + 4) // 5 for r in base[:4])


@need_exp
def test_tiles32_cut_pairs_hand_their_state_over(eng32, models32):
    """more pairs than workgroups (9100 reads = 285 pairs on 256 CUs): the lane schedule cuts pairs and the state crosses HBM;
    the calls are those of batches small enough that nothing is cut, for the transducer and the residual (rnnrf) stacks"""
    n = 9100
    base = [sig(300 + 7 * (i % 41), 9000 + i) for i in range(97)]
    reads = [base[(i * 13) % 97] for i in range(n)]
    for name in ("rgrgr_r94", "rnnrf_r94"):
        whole = [key(c) for c in eng32.basecall(reads, name)]
        ref = [key(c) for c in eng32.basecall(base, name)]
        assert all(whole[i] == ref[(i * 13) % 97] for i in range(n)), name


@need_exp
def test_tiles32_whole_pairs_when_handover_is_disabled(models32):
    w, _ = models32["rgrgr_r94"]
    base = [sig(300 + 11 * (i % 23), 8000 + i) for i in range(53)]
    reads = [base[(i * 5) % 53] for i in range(8800)]
    res = []
    for flag in ("1", "0"):
        os.environ["SCRAPPIE_HIP_HANDOVER"] = flag
        try:
            e = sa.Engine(0)
            e.debug_option("gru32", 1)
            e.load_model("rgrgr_r94", w)
            res.append([key(c) for c in e.basecall(reads, "rgrgr_r94")])
            e.close()
        finally:
            os.environ.pop("SCRAPPIE_HIP_HANDOVER", None)
    assert res[0] == res[1]


@need_exp
def test_tiles32_full_size_launch_group_and_long_read(eng32, models32):
    """BASELINE config 2's launch group (10 000 x 4000 samples: 313 pairs, cut between lanes) and a 60 000-sample read inside a
    batch: deterministic, independent of the batch"""
    n = 10000
    base = [sig(4000, 5000 + i) for i in range(64)]
    sigs = [base[i % 64] for i in range(n)]
    a = [key(c) for c in eng32.basecall(sigs, "rgrgr_r94")]
    assert all(a[i] == a[i % 64] for i in range(n)) and all(k[2] == 800 for k in a)
    assert [key(c) for c in eng32.basecall(base[:5], "rgrgr_r94")] == a[:5]
    long_read = sig(60000, 99)
    mixed = [long_read] + base[:40] + [sig(777, 5)]
    b = [key(c) for c in eng32.basecall(mixed, "rgrgr_r94")]
    assert b[0] == key(eng32.basecall([long_read], "rgrgr_r94")[0]) and b[1:41] == a[:40]


@need_exp
def test_two_tiles_in_opposite_phases_equal_one_tile_bit_for_bit(eng32, models32):
    """k_gru_proj32x2 (sh_gru32x2.h, debug_option gru32 = 2): two 32-read tiles per workgroup half a step apart, the chain waves' activations
    of one tile issued between the dependent MFMAs of the other, x_c handed over under an LDS count instead of a barrier.  Per tile
    it performs k_gru_proj32's operations in k_gru_proj32's order: trunk outputs and calls identical bit for bit -- single reads, 9100
    mixed reads (both slots busy, pairs cut between lanes), the residual stack, the benchmark's launch group, a long read in a batch."""
    e2 = sa.Engine(0)
    try:
        e2.debug_option("gru32", 2)
        for name in ("rgrgr_r94", "rnnrf_r94"):
            e2.load_model(name, models32[name][0])
        x = sig(2003, 5)
        for name in ("rgrgr_r94", "rnnrf_r94"):
            assert np.array_equal(eng32.trunk(x, name, 5), e2.trunk(x, name, 5)), name
        base = [sig(300 + 7 * (i % 41), 9000 + i) for i in range(97)]
        reads = [base[(i * 13) % 97] for i in range(9100)]
        p = eng32.default_params(local_pen=150.0)
        for name in ("rgrgr_r94", "rnnrf_r94"):
            a = [key(c) for c in eng32.basecall(reads, name, p)]
            b = [key(c) for c in e2.basecall(reads, name, p)]
            assert a == b, name
        big = [sig(4000, 5000 + i % 64) for i in range(10000)]
        assert [key(c) for c in eng32.basecall(big, "rgrgr_r94")] == [key(c) for c in e2.basecall(big, "rgrgr_r94")]
        mixed = [sig(60000, 99)] + base[:40] + [sig(777, 5), sig(12, 1)]
        assert [key(c) for c in eng32.basecall(mixed, "rgrgr_r94")] == [key(c) for c in e2.basecall(mixed, "rgrgr_r94")]
    finally:
        e2.close()
