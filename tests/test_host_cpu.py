"""CPU tests of the product's host side: the C ABI library loads and exports
every symbol include/scrappie_hip.h declares, the host-resident C functions
(signal prep, stitching, homopolymer, CRF posterior, record formatting) agree
bit-for-bit with the oracle / compiled-reference fixtures, and the GPU entry
points fail loudly without a GPU.  No GPU compute is attempted here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import scrappie_amd as sa
from scrappie_amd import model, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fp = C.POINTER(C.c_float)
ip = C.POINTER(C.c_int)


@pytest.fixture(scope="module")
def L():
    if not os.path.exists(sa.LIB_PATH):
        sa.build()
    return sa.lib()


def test_library_exports_every_declared_symbol(L):
    hdr = open(os.path.join(ROOT, "include", "scrappie_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    hdr = re.sub(r"typedef[^;{]*;", "", hdr)          # function-pointer typedefs are not symbols
    names = set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;", hdr))
    assert len(names) >= 45
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, "declared but not exported: %s" % missing


def test_no_cpu_fallback_without_gpu(L):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert L.scrappie_hip_engine_create(0) is None
    assert b"HIP device" in L.scrappie_hip_last_error()
    with pytest.raises(RuntimeError):
        sa.Engine(0)
    rt = sa.RawTable(np.zeros(4000, np.float32))
    with pytest.raises(RuntimeError):
        sa.calc_post(rt, "rgrgr_r94")


def test_medmad_and_trim_match_reference_fixtures(L, golden):
    g = golden["ref_signal_prep"]
    for n, seed in g["cases"]:
        n, seed = int(n), int(seed)
        sig = synth.synthetic_signal(n, seed, raw_units=True)
        if seed % 2 == 0:
            sig[:250] = sig[:250] * 0.02 + 90
            sig[-130:] = sig[-130:] * 0.02 + 90
        rt = sa.RawTable(sig)
        for perc in (0.0, 0.25):
            r = L.trim_raw_by_mad(rt.data(), 100 if n >= 200 else 10, perc)
            assert [r.start, r.end] == list(g["trim_%d_%g" % (seed, perc)])
        nrm = sig.copy()
        L.medmad_normalise_array(nrm.ctypes.data_as(fp), n)
        assert np.array_equal(nrm.view(np.uint32), g["norm_%d" % seed].view(np.uint32))


def test_reference_signal_files_through_scrappy_surface(golden):
    """RawTable.trim().scale() on the reference's own raw_signal.crp
    (src/test/test_scrappie_signal.c:59-103)."""
    g = golden["ref_test_files"]
    raw = g["raw_signal"].astype(np.float32)
    raw = ((raw + np.float32(16.0)) * (np.float32(1373.41) / np.float32(8192.0))).astype(np.float32)
    rt = sa.RawTable(raw).trim()
    assert (rt.start, rt.end) == (200, (len(raw) // 100) * 100 - 10)
    assert np.max(np.abs(rt.data(as_numpy=True) - g["trimmed_signal"])) <= 1e-4
    rt2 = sa.RawTable(g["trimmed_signal"]).scale()
    assert np.max(np.abs(rt2.data(as_numpy=True) - g["normalised_signal"])) <= 1e-5


def test_stitching_and_homopolymer_match_reference_fixture(L, golden):
    g = golden["ref_decode"]
    for (T, seed, klen, stay, skip, local, slip, hp) in g["transducer_cases"]:
        T, seed, klen, hp = int(T), int(seed), int(klen), int(hp)
        seq = np.ascontiguousarray(g["seq_%d" % seed], dtype=np.int32)
        pos = np.zeros(T + 1, np.int32)
        bases = sa._take_string(L.overlapper(seq.ctypes.data_as(ip), T + 1, 4 ** klen, pos.ctypes.data_as(ip)))
        assert (bases or "") == str(g["bases_%d" % seed])
        assert np.array_equal(pos, g["pos_%d" % seed])
        post = synth.fixture_posterior(T, seed, klen, hp)
        pm = sa.ScrappyMatrix.from_numpy(post, sloika=False)
        hseq = seq.copy()
        assert L.homopolymer_path(pm.data(), hseq.ctypes.data_as(ip), 1) == 0
        assert np.array_equal(hseq, g["hp_seq_%d" % seed])
        same = seq.copy()
        assert L.homopolymer_path(pm.data(), same.ctypes.data_as(ip), 0) == 0 and np.array_equal(same, seq)
    assert L.overlapper(np.full(4, -1, np.int32).ctypes.data_as(ip), 4, 1024, None) is None


def test_crf_host_functions_match_reference_fixture(L, golden):
    g = golden["ref_decode"]
    for T, seed in g["crf_cases"]:
        T, seed = int(T), int(seed)
        path = np.ascontiguousarray(g["crf_path_%d" % seed], dtype=np.int32)
        pos = np.zeros(T + 1, np.int32)
        bc = sa._take_string(L.crfpath_to_basecall(path.ctypes.data_as(ip), T, pos.ctypes.data_as(ip)))
        assert bc == str(g["crf_bases_%d" % seed]) and not pos.any()
        tr = synth.simulated_crf_transitions(T, seed)
        pm = sa.ScrappyMatrix.from_numpy(tr, sloika=False)
        pp = sa.ScrappyMatrix(L.posterior_crf(pm.data())).data(as_numpy=True, sloika=False)
        assert np.array_equal(pp.view(np.uint32), g["crf_post_%d" % seed].view(np.uint32))


def test_matrix_roundtrip_and_layout(L):
    """python/test/test_scrappy.py:208-234 style: numpy -> scrappie_matrix -> numpy"""
    a = np.arange(3 * 7, dtype=np.float32).reshape(3, 7)
    m = sa.ScrappyMatrix.from_numpy(a, sloika=True)
    c = m.data().contents
    assert (c.nr, c.nrq, c.nc, c.stride) == (7, 2, 3, 8)
    assert np.array_equal(m.data(as_numpy=True, sloika=True), a)
    assert np.array_equal(m.data(as_numpy=True, sloika=False)[:, -1], a[:, 0])


def test_fasta_record_format(L):
    """src/scrappie_raw.c:317-325 (two spaces after the id, %f fields)"""
    call = sa._Call()
    s = b"ACGT"
    buf = C.create_string_buffer(s)
    call.score = -12.5
    call.nblock = 10
    call.basecall = C.cast(buf, C.c_void_p)
    call.basecall_length = 4
    out = C.create_string_buffer(1024)
    n = L.scrappie_hip_format_fasta(out, 1024, b"uu-id", b"read1.fast5", False, b"pfx_", C.byref(call), 4000, 200, 3990)
    want = ('>pfx_read1.fast5  { "filename" : "read1.fast5", "uuid" : "uu-id", "normalised_score" : 1.250000,  '
            '"nblock" : 10,  "sequence_length" : 4,  "blocks_per_base" : 2.500000, "nsample" : 4000, '
            '"trim" : [ 200, 3990 ] }\nACGT\n')
    assert out.value.decode() == want and n == len(want)


def test_model_names_and_enum(L):
    """src/networks.c:17-34"""
    assert [L.get_raw_model(n) for n in (b"raw_r94", b"rgrgr_r94", b"rgrgr_r941", b"rgrgr_r10", b"rnnrf_r94", b"nope")] \
        == [0, 1, 2, 3, 4, 5]


def test_model_container_roundtrip(tmp_path):
    for name in model.MODEL_SHAPES:
        w = model.synthetic_model(name, seed=3, size=32)
        p = str(tmp_path / (name + ".scrm"))
        model.save_model(w, p)
        r = model.load_model(p)
        for k in model.matrix_names(w):
            assert np.array_equal(np.asarray(w[k]), r[k]), k
        assert (r["arch"], r["conv_act"], r["stride"]) == (w["arch"], w["conv_act"], w["stride"])
    d = model.model_dims(model.synthetic_model("rgrgr_r94"))
    assert d == dict(F=96, WL=11, S=96, NS=1025, stride=5)
    # 601.5 MFLOP per 800-block read
    assert abs(model.flops_per_block(model.synthetic_model("rgrgr_r94")) * 800 / 1e6 - 601.5) < 0.1


def test_model_header_ingest(tmp_path):
    """A header in the reference's generated-header grammar (misc/parse_rgrgr.py:28-55)
    converts to the same weights."""
    w = model.synthetic_model("rgrgr_r94", seed=4, size=16, nstate=65)
    lines = []

    def emit(name, arr, nr=None):
        arr = np.atleast_2d(np.asarray(arr, np.float32))
        nc, n_in = arr.shape
        nr = nr or n_in
        nrq = (nr + 3) // 4
        cols = []
        for c in range(nc):
            vals = list(arr[c]) + [0.0] * (4 * nrq - n_in)
            cols.append(", ".join(float(v).hex() for v in vals))
        lines.append("float __%s[] = {\n\t%s};" % (name, ",\n\t".join(cols)))
        lines.append("_Mat _%s = {\n\t.nr = %d,\n\t.nrq = %d,\n\t.nc = %d,\n\t.stride = %d,\n\t.data.f = __%s\n};"
                     % (name, nr, nrq, nc, 4 * nrq, name))
        lines.append("const scrappie_matrix %s = &_%s;\n" % (name, name))

    F, WL = w["conv_W"].shape
    spread = np.zeros((F, 4 * WL), np.float32)
    spread[:, 0::4] = w["conv_W"]
    emit("conv_rgrgr_r94_W", spread[:, :4 * WL - 3], nr=4 * WL - 3)
    emit("conv_rgrgr_r94_b", w["conv_b"].reshape(-1, 1))
    lines.append("const int conv_rgrgr_r94_stride = 5;")
    for l, tag in enumerate(("gruB1", "gruF2", "gruB3", "gruF4", "gruB5")):
        for n in ("iW", "sW", "sW2"):
            emit("%s_rgrgr_r94_%s" % (tag, n), w["gru%d_%s" % (l, n)])
        emit("%s_rgrgr_r94_b" % tag, w["gru%d_b" % l].reshape(-1, 1))
    emit("FF_rgrgr_r94_W", w["ff_W"])
    emit("FF_rgrgr_r94_b", w["ff_b"].reshape(-1, 1))
    p = tmp_path / "rgrgr_r94.h"
    p.write_text("\n".join(lines))
    # vectors are emitted as nc=1 matrices in the reference; this test wrote them as
    # (n,1) columns, so transpose back
    r = model.model_from_header(str(p))
    for k in ("conv_W", "ff_W", "gru0_iW", "gru3_sW", "gru4_sW2"):
        assert np.array_equal(r[k], w[k]), k
    assert r["stride"] == 5


# ---------------------------------------------------------------------------
# lane schedule of the recurrent kernel (scrappie_amd/csrc/sh_sched.h)
# ---------------------------------------------------------------------------
def _gru_schedule(tile_T, ncu, lpw=2):
    import ctypes as C
    L = sa.lib()
    L.scrappie_hip_lane_schedule.restype = C.c_long
    L.scrappie_hip_lane_schedule.argtypes = [C.POINTER(C.c_int), C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_int),
                                             C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_size_t]
    tt = np.ascontiguousarray(tile_T, dtype=np.int32)
    cap = len(tt) + 2 * ncu + 4
    lane_off = np.zeros(2 * ncu + 1, np.int32)
    seg = np.zeros((cap, 4), np.int32)
    nwg, capacity = C.c_int(), C.c_int()
    ip = C.POINTER(C.c_int)
    ns = L.scrappie_hip_lane_schedule(tt.ctypes.data_as(ip), len(tt), ncu, lpw, C.byref(nwg), C.byref(capacity),
                                      lane_off.ctypes.data_as(ip), seg.ctypes.data_as(ip), cap)
    assert 0 <= ns <= cap
    return nwg.value, capacity.value, lane_off[:lpw * nwg.value + 1], seg[:ns]


@pytest.mark.parametrize("lpw", [2, 1])
@pytest.mark.parametrize("case", ["uniform625", "ragged", "few", "paired", "zeros", "one", "many"])
def test_gru_lane_schedule(case, lpw):
    rng = np.random.default_rng(7)
    ncu = 256
    if case == "uniform625":
        tt = np.full(625, 800)
    elif case == "ragged":
        tt = np.sort(rng.integers(1, 3000, 1500))[::-1]
    elif case == "few":
        tt = np.sort(rng.integers(1, 900, 37))[::-1]
    elif case == "paired":
        tt = np.sort(rng.integers(200, 8001, 350))[::-1]
    elif case == "zeros":
        tt = np.concatenate([np.sort(rng.integers(1, 500, 700))[::-1], np.zeros(9, int)])
    elif case == "one":
        tt = np.array([17])
    else:
        tt = np.sort(rng.integers(100, 120, 5000))[::-1]
        ncu = 8
    nwg, M, lane_off, seg = _gru_schedule(tt, ncu, lpw)
    live = [i for i, t in enumerate(tt) if t > 0]
    assert nwg == min(ncu, len(live)) and len(lane_off) == lpw * nwg + 1     # a second lane only once every CU has one
    assert lane_off[0] == 0 and lane_off[-1] == len(seg) and np.all(np.diff(lane_off) >= 0)
    W = int(np.sum(tt))
    assert M >= max(tt) and (len(live) <= lpw * nwg or M == max(int(max(tt)), -(-W // (lpw * nwg))))
    pieces = {}
    for ln in range(lpw * nwg):
        t0 = 0
        rows = seg[lane_off[ln]:lane_off[ln + 1]]
        for k, (tile, s0, s1, _) in enumerate(rows):
            assert 0 <= s0 < s1 <= tt[tile]
            pieces.setdefault(int(tile), []).append(dict(lane=ln, s0=int(s0), s1=int(s1), start=t0,
                                                         first=(k == 0), last=(k == len(rows) - 1)))
            t0 += s1 - s0
        assert t0 <= M                                 # lane capacity
    assert sorted(pieces) == live                      # every live tile scheduled, dead ones not
    nsplit = 0
    for tile, ps in pieces.items():
        ps.sort(key=lambda p: p["s0"])
        assert ps[0]["s0"] == 0 and ps[-1]["s1"] == tt[tile] and len(ps) <= 2
        if len(ps) == 2:
            nsplit += 1
            head, tail = ps
            assert head["s1"] == tail["s0"]
            assert head["first"] and tail["last"]                    # producer runs first thing, consumer last thing
            assert head["lane"] // lpw <= tail["lane"] // lpw
            assert head["lane"] < tail["lane"]                       # hand-over goes to a higher-numbered lane
            assert head["start"] + (head["s1"] - head["s0"]) <= tail["start"]   # pieces do not overlap in time
    if len(live) <= lpw * nwg:
        assert nsplit == 0
        # whole tiles: every workgroup has a tile, and the longest tiles share theirs with the shortest or nothing
        per_wg = [[t for t, ps in pieces.items() if ps[0]["lane"] // lpw == w] for w in range(nwg)]
        assert all(1 <= len(v) <= lpw for v in per_wg)
        if len(live) > nwg:
            longest = int(np.argmax(tt))
            mates = [t for v in per_wg if longest in v for t in v if t != longest]
            assert mates and tt[mates[0]] == min(tt[t] for t in live)
            load = [sum(int(tt[t]) for t in v) for v in per_wg if len(v) == 2]
            assert max(load) - min(load) <= max(tt) - min(tt[t] for t in live)
    if case == "uniform625":
        assert nwg == 256 and M == (977 if lpw == 2 else 1954)        # 625 * 800 / 512 = 976.6, / 256 = 1953.1


@pytest.mark.parametrize("n,ncu,expect_k", [(625, 256, 2), (200, 256, 1), (257, 256, 1), (263, 256, 4), (769, 256, 1), (1000, 256, 1), (900, 256, 3)])
def test_decoder_piece_schedule(n, ncu, expect_k):
    """Viterbi pieces (sh_sched.h): every tile's blocks covered once, in order, a tile's
    earlier piece at the lower workgroup index, K = argmin ceil(n K / ncu) / K."""
    import ctypes as C
    L = sa.lib()
    L.scrappie_hip_decoder_pieces.restype = C.c_long
    ip = C.POINTER(C.c_int)
    L.scrappie_hip_decoder_pieces.argtypes = [ip, C.c_size_t, C.c_int, ip, C.c_size_t]
    rng = np.random.default_rng(n)
    tt = np.sort(rng.integers(1, 900, n))[::-1].astype(np.int32)
    tt[-3:] = 0                                   # dead tiles at the end
    tt = np.ascontiguousarray(tt)
    seg = np.zeros((4 * n, 4), np.int32)
    ns = L.scrappie_hip_decoder_pieces(tt.ctypes.data_as(ip), n, ncu, seg.ctypes.data_as(ip), len(seg))
    seg = seg[:ns]
    live = n - 3
    ks = [((live * k + ncu - 1) // ncu) / k for k in (1, 2, 3, 4)]
    k_best = 1 if live <= ncu else 1 + int(np.argmin(np.array(ks)))      # first minimum
    assert k_best == expect_k
    last_end, last_idx = {}, {}
    seen = {}
    for g, (tile, s0, s1, ordinal) in enumerate(seg):
        assert tt[tile] > 0 and 0 <= s0 < s1 <= tt[tile]
        assert ordinal == seen.get(int(tile), 0)          # the consumer waits for `ordinal` finished pieces
        seen[int(tile)] = ordinal + 1
        assert s0 == last_end.get(int(tile), 0)           # contiguous, in order
        assert g > last_idx.get(int(tile), -1)            # earlier piece, lower workgroup
        last_end[int(tile)] = int(s1); last_idx[int(tile)] = g
    assert sorted(last_end) == list(range(live)) and all(last_end[t] == tt[t] for t in last_end)
    assert max(np.bincount(seg[:, 0])) <= expect_k


def test_event_features_host_c_matches_oracle():
    """scrappie_hip_event_features (host C: nnfeatures.c:88 + layers.c:119) == the oracle's
    window(features_from_events(.)), bit for bit (the oracle's features are pinned on the
    compiled reference in test_oracle_golden.py)."""
    import oracle
    from scrappie_amd import synth
    for n, seed in ((300, 21), (2, 22), (17, 23), (1500, 31)):
        ev = synth.synthetic_events(n, seed)
        got = sa.event_features(ev)
        want = oracle.window(oracle.features_from_events(ev), 3, 1)
        assert got.shape == want.shape == (n, 12)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (n, seed)
    # a sub-range of the table
    ev = synth.synthetic_events(64, 5)
    got = sa.event_features(ev, start=10, end=50)
    et_want = oracle.window(oracle.features_from_events(ev[10:50].copy()), 3, 1)
    assert np.array_equal(got, et_want)


# ---------------------------------------------------------------------------
# launch-group planning (scrappie_hip_plan_groups): reads -> groups bounded in reads and column blocks
# ---------------------------------------------------------------------------
def _plan(lengths, stride, max_reads, max_blocks):
    import ctypes as C
    L = sa.lib()
    ln = np.ascontiguousarray(lengths, dtype=np.uint32)
    starts = np.zeros(max(len(ln), 1), dtype=np.uintp)
    ng = L.scrappie_hip_plan_groups(ln.ctypes.data_as(C.POINTER(C.c_uint32)), len(ln), stride, max_reads, max_blocks,
                                    starts.ctypes.data_as(C.POINTER(C.c_size_t)), len(starts))
    return ng, starts[:max(ng, 0)].astype(np.int64)


def test_plan_groups_bounds_reads_and_blocks():
    rng = np.random.default_rng(3)
    lengths = rng.integers(1000, 40001, size=5000)
    T = -(-lengths // 5)
    for max_reads, max_blocks in [(16384, 0), (512, 0), (16384, 20000), (100, 9000)]:
        ng, starts = _plan(lengths, 5, max_reads, max_blocks)
        assert ng >= 1 and starts[0] == 0 and np.all(np.diff(starts) > 0)
        ends = np.append(starts[1:], len(lengths))
        for lo, hi in zip(starts, ends):
            assert hi - lo <= max_reads
            if max_blocks:
                # what build_group will make of it: tiles of 16 from the reads sorted by length
                t = np.sort(T[lo:hi])[::-1]
                ncb = int(t[::16].sum())
                assert ncb <= max_blocks
                assert t.sum() // 16 + t.max() + 1 <= max_blocks
        if max_blocks == 0:
            assert ng == -(-len(lengths) // max_reads)
        else:       # greedy: no group could have taken the next read as well
            for lo, hi in zip(starts[:-1], ends[:-1]):
                t = T[lo:hi + 1]
                assert hi - lo == max_reads or t.sum() // 16 + t.max() + 1 > max_blocks


def test_plan_groups_edge_cases():
    assert _plan([], 5, 10, 0)[0] == 0
    assert _plan([4000], 5, 10, 0)[0] == 1
    assert _plan([4000], 5, 10, 100)[0] == -1          # one read alone exceeds the block bound
    ng, starts = _plan([4000] * 7, 5, 3, 0)
    assert ng == 3 and list(starts) == [0, 3, 6]
    ng, starts = _plan([0, 0, 4000, 0], 5, 16, 0)       # empty reads cost nothing but keep their place
    assert ng == 1 and list(starts) == [0]


def test_integration_cdef_links(tmp_path):
    """INTEGRATION.md section 1, option A: include/pyscrap_raw.h is the cdef text a maintainer gives cffi when the
    raw path is routed to this library.  cffi's API mode compiles a wrapper per prototype and links it; do the
    same with gcc: a translation unit that takes the address of every function the file declares must compile
    against scrappie_hip.h (same types) and link against the built library alone."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "pyscrap_raw.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"([A-Za-z_][A-Za-z_0-9]*)\s*\(", text)
    assert len(names) == 17 and "decode_transducer" in names and "nanonet_rnnrf_r94_transitions" in names
    src = tmp_path / "link.c"
    src.write_text('#include "scrappie_hip.h"\n' + text +          # the prototypes must agree with the library's header
                   "\nvoid *table[] = {" + ", ".join("(void *)" + n for n in names) + "};\n"
                   "int main(void) { return table[0] == 0; }\n")
    exe = tmp_path / "link"
    r = subprocess.run(["gcc", "-std=c99", "-I", os.path.join(root, "include"), str(src), "-o", str(exe),
                        "-L", os.path.join(root, "scrappie_amd"), "-lscrappie_hip",
                        "-Wl,-rpath," + os.path.join(root, "scrappie_amd")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr

def test_one_addition_move_selection_is_exact():
    """k_ff_viterbi's fast form of the state update (sh_decode.h): the three moves INTO a state add the same
    emission l to three per-quad values, so   max(l+sv, l+kv, l+ps) == l + max(sv, kv, ps)   exactly (rounding is
    monotone), and the move code is that of the first of (step, skip, start) holding the maximum m -- PROVIDED the
    quad passes the kernel's test  m - runner_up > 2^-21 (|m| + bound on |l|).  Replayed here in float32 against the
    reference's compare-by-compare form (decode.c:180-335), on values with planted near-ties down to one ulp."""
    rng = np.random.default_rng(7)
    n = 2_000_000
    f = np.float32
    base = rng.normal(-800.0, 300.0, n).astype(f)
    def near(x):
        """x moved by 0, +-1, +-2, ... ulps (most of the time) or by an ordinary amount"""
        k = rng.integers(-6, 7, n)
        y = x.copy()
        for _ in range(6):
            y = np.where(k > 0, np.nextafter(y, f(np.inf)), np.where(k < 0, np.nextafter(y, f(-np.inf)), y))
            k = k - np.sign(k)
        far = rng.random(n) < 0.5
        return np.where(far, (x + rng.normal(0, 5.0, n)).astype(f), y).astype(f)
    sv, kv, ps = base, near(base), near(base)
    perm = rng.integers(0, 3, n)                       # which of the three is the planted one varies
    sv, kv, ps = (np.choose(perm, [sv, kv, ps]), np.choose(perm, [kv, ps, sv]), np.choose(perm, [ps, sv, kv]))
    l = (-rng.random((4, n)) * 11.5).astype(f)         # four emissions of the quad: log-posteriors in [log 1e-5, 0]
    pv = (base + rng.normal(0, 5.0, n)).astype(f)
    stay_v = f(-0.7)
    for e in range(4):
        # reference: stay, then step / skip / start replace on strictly greater
        sc = (pv + stay_v).astype(f); code = np.zeros(n, np.int8)
        for c, x in ((1, sv), (2, kv), (3, ps)):
            cand = (l[e] + x).astype(f)
            up = sc < cand
            sc = np.where(up, cand, sc); code = np.where(up, c, code)
        # fast form
        m = np.maximum(np.maximum(sv, kv), ps)
        md = np.median(np.stack([sv, kv, ps]), axis=0).astype(f)
        # (the kernel bounds max|l| of the quad by -log(min_prob) + 1e-3 instead of looking: l = log(min_prob + ..) <= ~0)
        lbound = f(1.0e-3) - np.log(f(1e-5)).astype(f)
        assert np.max(np.abs(l)) <= lbound
        clear = (m - md).astype(f) > ((lbound + np.abs(m)).astype(f) * f(2.0 ** -21)).astype(f)
        cm = np.where(sv == m, 1, np.where(kv == m, 2, 3))
        s0 = (pv + stay_v).astype(f)
        mv = (l[e] + m).astype(f)
        fsc = np.maximum(s0, mv)
        fcode = np.where(s0 < mv, cm, 0)
        assert np.array_equal(fsc, sc)                                   # the score: always
        assert np.array_equal(fcode[clear], code[clear])                 # the move: wherever the kernel uses the fast form
        assert 0.2 < clear.mean() < 0.9 and (code[~clear] != fcode[~clear]).any()      # the test bites: ties do occur in the rest


def test_host_thread_budget_follows_cgroup_and_local_world_size():
    """Eight ranks on a node share its CPUs: an engine's gathering / stitching threads = min(affinity, cgroup quota, 32) /
    LOCAL_WORLD_SIZE, at least 1 (scrappie_hip_host_thread_budget; read once per process, hence the subprocesses)."""
    import subprocess
    import sys
    cpus = float(len(os.sched_getaffinity(0)))
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cpus = min(cpus, float(q) / float(period))
    except OSError:
        pass
    code = ("import ctypes, scrappie_amd as sa; L = sa.lib(); L.scrappie_hip_host_thread_budget.restype = ctypes.c_uint; "
            "print(L.scrappie_hip_host_thread_budget())")
    got = {}
    for lws in (None, "2", "8", "64"):
        env = {k: v for k, v in os.environ.items() if k not in ("LOCAL_WORLD_SIZE", "SCRAPPIE_HIP_HOST_THREADS")}
        if lws:
            env["LOCAL_WORLD_SIZE"] = lws
        r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-1000:]
        got[lws] = int(r.stdout.strip().splitlines()[-1])
        assert got[lws] == max(1, int(min(cpus / (int(lws) if lws else 1), 32.0))), (lws, got, cpus)
    assert got["8"] <= max(1, got[None] // 8 + 1) and got["64"] >= 1
    env = dict(os.environ, SCRAPPIE_HIP_HOST_THREADS="5", LOCAL_WORLD_SIZE="8")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert int(r.stdout.strip().splitlines()[-1]) == 5


def test_coalescer_under_thread_sanitizer(tmp_path):
    """The leader / follower queue behind the per-read functions (scrappie_amd/csrc/sh_coalesce.h: one template for the posterior,
    decode_transducer and decode_crf coalescers) built with -fsanitize=thread and driven by 32 threads with stub launches
    (tests/coalesce_tsan.cpp): every call gets the result of its own input, every request is served once, no report."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / "coalesce_tsan")
    b = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", "-I" + os.path.join(ROOT, "scrappie_amd", "csrc"),
                        os.path.join(ROOT, "tests", "coalesce_tsan.cpp"), "-o", exe], capture_output=True, text=True, timeout=300)
    if b.returncode != 0 and "tsan" in (b.stderr or "").lower():
        pytest.skip("no ThreadSanitizer runtime in this toolchain")
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe, "32", "120"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    assert "ThreadSanitizer" not in r.stderr and "0 wrong" in r.stdout


def test_cli_pipeline_under_thread_sanitizer(tmp_path):
    """`scrappie raw` (scrappie_raw.c: loader / engine / writer threads over a ring of batches, three preparer slots, streaming engine
    calls, deferred tickets, shares per GPU) built with -fsanitize=thread against tests/cli_pipe_stub.c, which stands in for the GPU
    side and makes every call a checksum of the read's window READ WHEN THE CALL IS DELIVERED -- so a slot reused too early, a batch
    written before it is complete or a record attached to the wrong read changes the output.  1500 files of eight lengths (too short
    for a call ... long enough to be deferred), several batch sizes, one and three "GPUs": records equal the expectation, no report."""
    import shutil
    import subprocess
    import zlib
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    csrc = os.path.join(ROOT, "scrappie_amd", "csrc")
    exe = str(tmp_path / "scrappie_tsan")
    b = subprocess.run(["gcc", "-std=gnu11", "-O1", "-g", "-fsanitize=thread", "-pthread", "-Wno-unknown-pragmas", "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
                        os.path.join(csrc, "scrappie_raw.c"), os.path.join(ROOT, "tests", "cli_pipe_stub.c"), os.path.join(csrc, "sh_host.c"),
                        os.path.join(csrc, "sh_fast5.c"), os.path.join(csrc, "sh_h5mini.c"), os.path.join(csrc, "sh_inflate.c"), "-o", exe, "-lm", "-ldl"], capture_output=True, text=True, timeout=300)
    if b.returncode != 0 and "tsan" in (b.stderr or "").lower():
        pytest.skip("no ThreadSanitizer runtime in this toolchain")
    assert b.returncode == 0, b.stderr[-2000:]
    rdir = tmp_path / "reads"
    rdir.mkdir()
    rng = np.random.default_rng(5)
    want = {}
    for i in range(1500):
        n = int(rng.choice([100, 205, 211, 400, 1000, 2500, 4000, 7000], p=[.03, .02, .02, .2, .3, .3, .1, .03]))
        x = (90 + 10 * rng.standard_normal(n)).astype(np.float32)
        x.tofile(str(rdir / ("r%05d.f32" % i)))
        if n > 210:                                      # the stub's window: [200, n - 10)
            h, s = zlib.crc32(x[200:n - 10].tobytes()), ""
            for _ in range(12):
                s += "ACGT"[h & 3]
                h = ((h >> 2) | (h << 30)) & 0xffffffff
            want["r%05d.f32" % i] = s

    last_err = [""]

    def run(extra, env=None, first=None):
        r = subprocess.run([exe, "raw", "--model-file", "/dev/null"] + extra + ([first] if first else []) + [str(rdir)], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, **(env or {})))
        assert r.returncode == 0 and "Sanitizer" not in r.stderr, r.stderr[-3000:]
        last_err[0] = r.stderr
        lines = r.stdout.split("\n")
        return {l[1:].split()[0]: lines[j + 1] for j, l in enumerate(lines) if l.startswith(">")}
    for extra in (["--batch", "64"], ["--batch", "128", "--threads", "1"], ["--batch", "300", "--devices", "0,1,2"], ["--batch", "4000"]):
        assert run(extra) == want, extra
    host = run(["--batch", "100", "--prep", "host"])      # the reference's own trimming on the loader threads (sh_host.c), calls from the stub
    assert len(host) > 1300 and host == run(["--batch", "333", "--prep", "host"])
    # round 6 (scrappie_raw.c:540): a device-prepared batch is bounded by samples, reservations are checked, failures fall back
    # (a) reads of ~1000 samples under a budget of 100 000 samples per GPU and batch: batches shrink, every record as before
    assert run(["--batch", "1000"], {"SCRAPPIE_PREP_SAMPLES": "20000"}) == want and "reads per GPU (SCRAPPIE_PREP_SAMPLES" in last_err[0]
    # (b) the reservation of the preparers' buffers fails until the batch has been halved twice: same records
    f0 = str(rdir / "r00007.f32")                        # named first on the command line (and again by the directory: same record twice): the file the buffers are sized by
    n0 = np.fromfile(f0, dtype=np.float32).size
    assert run(["--batch", "1024"], {"STUB_RESERVE_FAIL_ABOVE": str(int(1.25 * n0 * 300 + 65536))}, first=f0) == want
    assert "out of memory reserving" in last_err[0] and "reads per GPU" in last_err[0]
    # (c) no batch size fits: the run prepares on the host from the start (= --prep host)
    assert run(["--batch", "333"], {"STUB_RESERVE_FAIL_ABOVE": "1"}) == host and "preparing signals on the host" in last_err[0]
    # (d) one batch's preparation fails at run time: THAT batch is prepared on the host (its records are the host form's), the others unchanged
    for extra in (["--batch", "300"], ["--batch", "300", "--devices", "0,1,2"]):
        got = run(extra, {"STUB_PREP_FAIL_AT": "2"})
        assert "preparing them on the host" in last_err[0]
        diff = [k for k in set(got) | set(want) if got.get(k) != want.get(k)]
        assert 0 < len(diff) <= 300 and all(got.get(k) == host.get(k) for k in diff), (len(diff), extra)


# ---------------------------------------------------------------- the built-in inflater (sh_inflate.c) against zlib
def _inflate_lib(tmp_path):
    """sh_inflate.c compiled on its own (host C: runs without the HIP library)"""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    csrc = os.path.join(ROOT, "scrappie_amd", "csrc")
    so = str(tmp_path / "libinflate.so")
    subprocess.run(["gcc", "-O2", "-std=c99", "-fPIC", "-shared", "-Wall", "-Wextra", "-Werror", "-I" + csrc, os.path.join(csrc, "sh_inflate.c"), "-o", so], check=True)
    L = C.CDLL(so)
    L.sh_zlib_inflate.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
    L.sh_zlib_inflate.restype = C.c_int

    def inf(z, cap):
        out = C.create_string_buffer(cap + 1)
        n = C.c_size_t(0)
        rc = L.sh_zlib_inflate(out, cap, C.byref(n), z, len(z))
        return rc, out.raw[:n.value]
    return inf


def test_builtin_inflater_equals_zlib_on_valid_streams(tmp_path):
    """The fast5 reader's own inflate (what libhdf5's deflate filter does for fast5_interface.c:130-217) against zlib as the oracle: every
    compression level, window size and strategy (stored, fixed and dynamic blocks; literal-only, run-length, long matches, codes of up to
    15 bits), output buffers with slack, exactly full and one byte short, trailing garbage, multi-block streams with sync / full flushes."""
    import zlib
    inf = _inflate_lib(tmp_path)
    rng = np.random.default_rng(1)

    def cases():
        yield b""
        yield b"a"
        yield b"abc" * 1000
        yield bytes(100000)
        yield bytes(range(256)) * 50
        for n in (1, 2, 3, 7, 8, 9, 255, 256, 257, 258, 259, 1000, 8000, 65535, 65536, 70000, 200000):
            yield rng.integers(0, 256, n, dtype=np.uint8).tobytes()
            yield rng.integers(0, 4, n, dtype=np.uint8).tobytes()
            yield rng.integers(0, 2, n, dtype=np.uint8).tobytes()
            yield (500 + 40 * np.repeat(rng.standard_normal(n // 9 + 1), 9)[:n] + 5 * rng.standard_normal(n)).astype(np.int16).tobytes()      # squiggle-like int16
            p = 0.5 ** np.arange(1, 257)
            yield rng.choice(256, size=n, p=p / p.sum()).astype(np.uint8).tobytes()           # skewed: codes of every length up to 15
            blk = rng.integers(0, 256, 1 + n // 7, dtype=np.uint8).tobytes()
            yield (blk * 8)[:n]                                                                # matches at long distances
    nt = 0
    for data in cases():
        for level in (0, 1, 6, 9):
            for wbits in (15, 9):
                for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
                    co = zlib.compressobj(level, zlib.DEFLATED, wbits, 8, strat)
                    z = co.compress(data) + co.flush()
                    rc, out = inf(z, len(data) + 320)
                    assert rc == 0 and out == data, (len(data), level, wbits, strat, rc, len(out))
                    rc, out = inf(z, len(data))
                    assert rc == 0 and out == data, ("exactly full", len(data), level, wbits, strat)
                    if data:
                        assert inf(z, len(data) - 1)[0] != 0, "output overflow not detected"
                    rc, out = inf(z + b"\x55\xaa" * 7, len(data) + 320)
                    assert rc == 0 and out == data, "trailing bytes"
                    nt += 1
    for data in (rng.integers(0, 256, 50000, dtype=np.uint8).tobytes(), b"hello world " * 5000):
        co = zlib.compressobj(6)
        z = b""
        for i in range(0, len(data), 777):
            z += co.compress(data[i:i + 777])
            if (i // 777) % 3 == 0:
                z += co.flush(zlib.Z_SYNC_FLUSH)
            if (i // 777) % 5 == 0:
                z += co.flush(zlib.Z_FULL_FLUSH)
        z += co.flush()
        rc, out = inf(z, len(data) + 320)
        assert rc == 0 and out == data
    assert nt > 3000


def test_builtin_inflater_agrees_with_zlib_on_corrupt_streams(tmp_path):
    """8000 corrupted / truncated copies of a level-1 stream of squiggle-like int16 (1-3 flipped bits, one in five also cut short): never
    a crash, accepted exactly when zlib accepts, and then with zlib's bytes (the Adler-32 catches what still parses)."""
    import zlib
    inf = _inflate_lib(tmp_path)
    rng = np.random.default_rng(2)
    data = (500 + 40 * np.repeat(rng.standard_normal(900), 9)[:8000] + 3 * rng.standard_normal(8000)).astype(np.int16).tobytes()
    z0 = zlib.compress(data, 1)
    both_ok = 0
    for trial in range(8000):
        z = bytearray(z0)
        for _ in range(int(rng.integers(1, 4))):
            z[int(rng.integers(0, len(z)))] ^= 1 << int(rng.integers(0, 8))
        if trial % 5 == 0:
            z = z[:int(rng.integers(1, len(z)))]
        z = bytes(z)
        try:
            d = zlib.decompressobj()
            ref = d.decompress(z, len(data) + 320)
            ok_ref = d.eof and not d.unconsumed_tail
        except zlib.error:
            ok_ref, ref = False, None
        rc, out = inf(z, len(data) + 320)
        assert (rc == 0) == ok_ref, (trial, rc, ok_ref)
        if ok_ref:
            assert out == ref
            both_ok += 1
    assert both_ok < 8000


def test_builtin_inflater_under_address_sanitizer(tmp_path):
    """sh_inflate.c parses untrusted files: valid, truncated, bit-flipped and random streams through it with input and output buffers malloc'd to EXACTLY
    their sizes (room for the output: exact, 320 spare, one byte short, half), built with -fsanitize=address,undefined -- no report, and every valid stream
    with enough room accepted."""
    import shutil
    import struct
    import subprocess
    import zlib
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    csrc = os.path.join(ROOT, "scrappie_amd", "csrc")
    exe = str(tmp_path / "inflate_asan")
    b = subprocess.run(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-std=gnu11", "-I" + csrc, os.path.join(ROOT, "tests", "inflate_asan.c"),
                        os.path.join(csrc, "sh_inflate.c"), "-o", exe], capture_output=True, text=True, timeout=300)
    if b.returncode != 0 and "asan" in (b.stderr or "").lower():
        pytest.skip("no AddressSanitizer runtime in this toolchain")
    assert b.returncode == 0, b.stderr[-2000:]
    rng = np.random.default_rng(7)
    recs, must_accept = [], 0

    def add(z, cap):
        recs.append(struct.pack("<II", len(z), cap) + z)
    p = 0.5 ** np.arange(1, 257)
    datas = [b"", b"a", bytes(70000), rng.integers(0, 256, 70000, dtype=np.uint8).tobytes(), b"abcdefgh" * 9000,
             (500 + 40 * np.repeat(rng.standard_normal(5000), 9)[:40000] + 4 * rng.standard_normal(40000)).astype(np.int16).tobytes(),
             rng.choice(256, size=50000, p=p / p.sum()).astype(np.uint8).tobytes()]
    for d in datas:
        for lvl in (0, 1, 9):
            for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
                co = zlib.compressobj(lvl, zlib.DEFLATED, 15, 8, strat)
                z = co.compress(d) + co.flush()
                for cap in (len(d), len(d) + 320, max(len(d) - 1, 0), len(d) // 2):
                    add(z, cap)
                must_accept += 2 + (1 if len(d) == 0 else 0) * 2          # exact and spare room (an empty output also fits the two smaller ones)
                for t in range(25):
                    zz = bytearray(z)
                    for _ in range(int(rng.integers(1, 4))):
                        zz[int(rng.integers(0, len(zz)))] ^= 1 << int(rng.integers(0, 8))
                    if t % 3 == 0 and len(zz) > 1:
                        zz = zz[:int(rng.integers(1, len(zz)))]
                    add(bytes(zz), len(d) + int(rng.integers(0, 400)))
    for t in range(2000):
        add(bytes([0x78, 0x01]) + rng.integers(0, 256, int(rng.integers(0, 300)), dtype=np.uint8).tobytes(), int(rng.integers(0, 5000)))
    corpus = tmp_path / "corpus.bin"
    corpus.write_bytes(b"".join(recs))
    r = subprocess.run([exe, str(corpus)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, (r.stdout + r.stderr)[-3000:]
    n, ok = (int(x) for x in re.findall(r"\d+", r.stdout)[:2])
    assert n == len(recs) and ok >= must_accept, (n, ok, must_accept)

