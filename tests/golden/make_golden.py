#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz.  Runs ONLY in the build container (needs
/root/reference and oracle/_ref built by `make -C oracle`).

Two kinds of fixture, both DATA (inputs + expected outputs), no reference source:

1. The reference's own test data files (src/test/*.crp, hex-float text:
   src/test/scrappie_util.c:22-47) re-encoded as compact numpy arrays.
2. Outputs of the reference's own code compiled here (oracle/_ref):
   - header-inline vector math (util.h:170-198) on a fixed input grid,
   - signal prep (util.c, scrappie_common.c) on seeded synthetic signals,
   - decode.c (decode_transducer, sloika_viterbi, overlapper, decode_crf,
     crfpath_to_basecall, posterior_crf) and homopolymer.c on seeded simulated
     posteriors (inputs are regenerated from the seed by scrappie_amd.synth).
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from scrappie_amd import synth  # noqa: E402

REFTEST = "/root/reference/src/test"


def read_crp(path):
    """src/test/crp.py:7-15"""
    with open(path) as fh:
        nr, nc = [int(x) for x in fh.readline().split()]
        mat = np.zeros((nc, nr))
        for col in range(nc):
            mat[col] = [float.fromhex(x) for x in fh.readline().split()]
    return mat


def crp_fixtures():
    raw = read_crp(os.path.join(REFTEST, "raw_signal.crp"))[:, 0]
    assert np.all(raw == np.round(raw)) and np.abs(raw).max() < 32768
    trimmed = read_crp(os.path.join(REFTEST, "trimmed_signal.crp"))[:, 0].astype(np.float32)
    normed = read_crp(os.path.join(REFTEST, "normalised_signal.crp"))[:, 0].astype(np.float32)
    path = read_crp(os.path.join(REFTEST, "path.crp"))[:, 0].astype(np.int32)
    tm = read_crp(os.path.join(REFTEST, "test_matrix.crp")).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "ref_test_files.npz"),
                        raw_signal=raw.astype(np.int16), trimmed_signal=trimmed,
                        normalised_signal=normed, path=path, test_matrix=tm)


def math_fixtures(rp):
    xs = np.concatenate([
        np.linspace(-100, 100, 4001), np.linspace(-2, 2, 2001),
        np.array([0.0, -0.0, 1, 2, 3, 4, -1, -2, -3, -4, 88.37, 88.38, -87.6, -87.7, -88.4, 1e-30, -1e-30]),
    ]).astype(np.float32)
    pos = np.concatenate([np.logspace(-44, 2, 3001), np.linspace(1e-6, 1.0, 2001),
                          np.array([1e-5, 1.0, 0.5, 0.70710678, 0.7071068])]).astype(np.float32)
    out = {"x": xs, "xpos": pos}
    fp = C.POINTER(C.c_float)
    for name, inp in (("expfv", xs), ("logisticfv", xs), ("tanhfv", xs), ("elufv", xs), ("logfv", pos)):
        fn = getattr(rp, "ref_" + name)
        fn.argtypes = [fp, fp, C.c_size_t]
        res = np.zeros_like(inp)
        fn(inp.ctypes.data_as(fp), res.ctypes.data_as(fp), len(inp))
        out[name] = res
    np.savez_compressed(os.path.join(HERE, "ref_math.npz"), **out)


def signal_fixtures(rp):
    fp = C.POINTER(C.c_float)
    rp.medmad_normalise_array.argtypes = [fp, C.c_size_t]
    rp.trim_raw_by_mad.restype = oracle.RawTable
    rp.trim_raw_by_mad.argtypes = [oracle.RawTable, C.c_size_t, C.c_float]
    rp.medianf.restype = C.c_float
    rp.medianf.argtypes = [fp, C.c_size_t]
    rp.madf.restype = C.c_float
    rp.madf.argtypes = [fp, C.c_size_t, fp]
    out = {}
    cases = [(4000, 11), (4001, 12), (3790, 13), (10000, 14), (257, 15), (100, 16)]
    out["cases"] = np.array(cases, dtype=np.int64)
    for n, seed in cases:
        sig = synth.synthetic_signal(n, seed, raw_units=True)
        if seed % 2 == 0:           # low-variance stall at both ends so the MAD trim bites
            sig[:250] = sig[:250] * 0.02 + 90
            sig[-130:] = sig[-130:] * 0.02 + 90
        rt, keep = oracle.raw_table(sig.copy())
        for perc in (0.0, 0.25):
            r = rp.trim_raw_by_mad(rt, 100 if n >= 200 else 10, perc)
            out["trim_%d_%g" % (seed, perc)] = np.array([r.start, r.end], dtype=np.int64)
        out["median_%d" % seed] = np.float32(rp.medianf(sig.ctypes.data_as(fp), n))
        out["mad_%d" % seed] = np.float32(rp.madf(sig.ctypes.data_as(fp), n, None))
        nrm = sig.copy()
        rp.medmad_normalise_array(nrm.ctypes.data_as(fp), n)
        out["norm_%d" % seed] = nrm
    np.savez_compressed(os.path.join(HERE, "ref_signal_prep.npz"), **out)


def decode_fixtures(rd, rp):
    PM = C.POINTER(oracle.Mat)
    ip = C.POINTER(C.c_int)
    rd.decode_transducer.restype = C.c_float
    rd.decode_transducer.argtypes = [PM, C.c_float, C.c_float, C.c_float, ip, C.c_bool]
    rd.sloika_viterbi.restype = C.c_float
    rd.sloika_viterbi.argtypes = [PM, C.c_float, C.c_float, C.c_float, ip]
    rd.overlapper.restype = C.c_void_p
    rd.overlapper.argtypes = [ip, C.c_size_t, C.c_int, ip]
    rd.decode_crf.restype = C.c_float
    rd.decode_crf.argtypes = [PM, ip]
    rd.crfpath_to_basecall.restype = C.c_void_p
    rd.crfpath_to_basecall.argtypes = [ip, C.c_size_t, ip]
    rd.posterior_crf.restype = PM
    rd.posterior_crf.argtypes = [PM]
    rd.free_scrappie_matrix.restype = PM
    rd.free_scrappie_matrix.argtypes = [PM]
    rp.homopolymer_path.restype = C.c_int
    rp.homopolymer_path.argtypes = [PM, ip, C.c_int]

    out = {}
    # (T, seed, klen, stay, skip, local, slip, homopolymers)
    cases = [(150, 101, 5, 0.0, 0.0, 2.0, 0, 0), (150, 102, 5, 2.0, 0.0, 2.0, 0, 0),
             (150, 103, 5, 0.0, 2.0, 2.0, 0, 0), (200, 104, 5, 0.0, 0.0, 2.0, 1, 0),
             (300, 105, 5, 0.0, 0.0, 2.0, 0, 6), (64, 106, 4, 0.5, 0.5, 1.0, 1, 0),
             (40, 107, 3, 0.0, 0.0, 2.0, 0, 0), (5, 108, 5, 0.0, 0.0, 2.0, 0, 0),
             (800, 109, 5, 0.0, 0.0, 2.0, 0, 8),
             # round 3: an expensive start state forces the path through the k-mer states (step / skip / slip
             # decide the result, decode.c:326-349); slip at T = 800; flat posteriors (hp < 0: near-ties everywhere)
             (800, 110, 5, 0.0, 0.0, 100.0, 0, 8), (800, 111, 5, 0.0, 0.0, 2.0, 1, 8),
             (400, 112, 5, 0.3, 0.2, 250.0, 1, 4), (300, 113, 5, 1.0, 1.5, 1000.0, 0, 0),
             (256, 114, 4, 0.0, 0.4, 100.0, 1, 2), (120, 115, 3, 0.2, 0.0, 150.0, 0, 0),
             (1500, 116, 5, 0.0, 0.0, 2.0, 0, 10), (500, 117, 5, 0.0, 0.0, 120.0, 0, -1),
             (500, 118, 5, 0.1, 0.3, 200.0, 1, -1), (800, 119, 5, 0.0, 0.0, 2.0, 0, -1),
             (200, 120, 4, 0.0, 0.0, 300.0, 1, -1), (150, 121, 3, 0.0, 0.25, 100.0, 0, -1)]
    out["transducer_cases"] = np.array(cases, dtype=np.float64)
    for (T, seed, klen, stay, skip, local, slip, hp) in cases:
        post = synth.fixture_posterior(T, seed, klen, hp)
        score, seq = oracle.decode_transducer(post, stay, skip, local, bool(slip), fn=rd.decode_transducer)
        out["seq_%d" % seed] = seq
        out["score_%d" % seed] = np.float32(score)
        if not slip:
            s2, seq2 = oracle.decode_transducer(post, stay, skip, local, False,
                                                fn=lambda m, a, b, c, s, sl: rd.sloika_viterbi(m, a, b, c, s))
            out["sloika_seq_%d" % seed] = seq2
            out["sloika_score_%d" % seed] = np.float32(s2)
        bases, pos = oracle.overlapper(seq, 4 ** klen, fn=rd.overlapper)
        out["bases_%d" % seed] = np.array(bases if bases is not None else "")
        out["pos_%d" % seed] = pos
        rc, hseq = oracle.homopolymer_path(post, seq, fn=rp.homopolymer_path, mean_flag=1)
        out["hp_seq_%d" % seed] = hseq
        hb, hpos = oracle.overlapper(hseq, 4 ** klen, fn=rd.overlapper)
        out["hp_bases_%d" % seed] = np.array(hb if hb is not None else "")
    crf_cases = [(100, 201), (800, 202), (2, 203), (1, 204)]
    out["crf_cases"] = np.array(crf_cases, dtype=np.int64)
    for T, seed in crf_cases:
        tr = synth.simulated_crf_transitions(T, seed)
        score, path = oracle.decode_crf(tr, fn=rd.decode_crf)
        out["crf_path_%d" % seed] = path
        out["crf_score_%d" % seed] = np.float32(score)
        out["crf_bases_%d" % seed] = np.array(oracle.crfpath_to_basecall(path, T, fn=rd.crfpath_to_basecall))
        m = oracle.NpMat(tr)
        pp = rd.posterior_crf(m.ptr)
        out["crf_post_%d" % seed] = oracle.mat_to_numpy(pp, rd.free_scrappie_matrix)
    np.savez_compressed(os.path.join(HERE, "ref_decode.npz"), **out)


def events_fixtures(rf):
    """Event features from the reference's nnfeatures.c compiled as shipped
    (oracle/_ref/libref_features.so); inputs are regenerated from the seed by
    scrappie_amd.synth.synthetic_events."""
    PM = C.POINTER(oracle.Mat)
    rf.nanonet_features_from_events.restype = PM
    rf.nanonet_features_from_events.argtypes = [oracle.EventTable, C.c_bool]
    rf.free_scrappie_matrix.restype = PM
    rf.free_scrappie_matrix.argtypes = [PM]
    out = {}
    cases = [(300, 21), (2, 22), (17, 23), (4001, 24)]
    out["cases"] = np.array(cases, dtype=np.int64)
    for n, seed in cases:
        ev = synth.synthetic_events(n, seed)
        for norm in (True, False):
            out["feat_%d_%d" % (seed, int(norm))] = oracle.features_from_events(
                ev, normalise=norm, fn=rf.nanonet_features_from_events, free=rf.free_scrappie_matrix)
    np.savez_compressed(os.path.join(HERE, "ref_events.npz"), **out)


def bundled_reads():
    """BASELINE config 1: the three fast5 files bundled with the reference
    (reads/*.fast5), re-encoded as int16 DAC counts + (offset, range,
    digitisation) so the GPU box (no HDF5, no /root/reference) can run them:
    tests/golden/reads/<name>.i16, format read by scrappie_amd/csrc/sh_fast5.c."""
    import glob
    import scrappie_amd as sa
    L = sa.lib()
    L.scrappie_hip_read_raw.restype = sa._RawTable
    L.scrappie_hip_read_raw.argtypes = [C.c_char_p, C.c_bool]
    L.scrappie_hip_fast5_scaling.argtypes = [C.c_char_p, C.POINTER(C.c_float)]
    outdir = os.path.join(HERE, "reads")
    os.makedirs(outdir, exist_ok=True)
    meta = {}
    for f in sorted(glob.glob("/root/reference/reads/*.fast5")):
        rt = L.scrappie_hip_read_raw(f.encode(), False)
        counts = np.ctypeslib.as_array(rt.raw, shape=(rt.n,)).copy()
        assert np.all(counts == np.round(counts)) and np.abs(counts).max() < 32768
        sc = (C.c_float * 3)()
        assert L.scrappie_hip_fast5_scaling(f.encode(), sc) == 0
        name = os.path.basename(f)[:-6]
        with open(os.path.join(outdir, name + ".i16"), "wb") as fh:
            fh.write(np.array(list(sc), dtype=np.float32).tobytes())
            fh.write(counts.astype(np.int16).tobytes())
        pa = L.scrappie_hip_read_raw(f.encode(), True)
        meta[name] = dict(n=int(rt.n), uuid=rt.uuid.decode(), first_pA=float(pa.raw[0]))
    import json
    json.dump(meta, open(os.path.join(outdir, "reads.json"), "w"), indent=1, sort_keys=True)


def main():
    oracle.build()
    rp, rd = oracle.ref_pure(), oracle.ref_decode()
    assert rp is not None and rd is not None, "oracle/_ref not built (needs /root/reference)"
    crp_fixtures()
    math_fixtures(rp)
    signal_fixtures(rp)
    decode_fixtures(rd, rp)
    rf = oracle.ref_features()
    assert rf is not None, "oracle/_ref/libref_features.so not built"
    events_fixtures(rf)
    bundled_reads()
    sys.path.insert(0, HERE)
    import provenance
    provenance.write()          # sha256 of every fixture and of the reference files it derives from
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print("%-28s %8d bytes" % (f, os.path.getsize(os.path.join(HERE, f))))


if __name__ == "__main__":
    main()
