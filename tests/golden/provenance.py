#!/usr/bin/env python3
"""Where every fixture under tests/golden/ comes from: the sha256 of the fixture file itself and of every file of the reference
checkout it was generated from (read by the generator, compiled into the oracle/_ref library that produced it, or -- for the
float64 network statements -- the sources they restate).  tests/golden/PROVENANCE.json is written by the generators
(make_golden.py, make_net_f64.py) through write(); tests/test_oracle_golden.py::test_fixture_provenance checks the fixture hashes
always and the reference hashes whenever /root/reference is present, so that neither side can drift silently.

    python tests/golden/provenance.py        # rewrite PROVENANCE.json from the files as they are now (needs /root/reference)
"""
import glob
import hashlib
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
OUT = os.path.join(HERE, "PROVENANCE.json")

PURE = ["src/util.c", "src/util.h", "src/sse_mathfun.h", "src/scrappie_common.c", "src/scrappie_common.h", "src/scrappie_seq_helpers.c",
        "src/scrappie_seq_helpers.h", "src/homopolymer.c", "src/homopolymer.h", "src/scrappie_stdlib.h", "src/scrappie_structures.h"]
DECODE = ["src/decode.c", "src/decode.h", "src/util.c", "src/util.h", "src/sse_mathfun.h", "src/scrappie_matrix.h"]
NET = ["src/layers.c", "src/layers.h", "src/networks.c", "src/networks.h", "src/scrappie_matrix.c", "src/scrappie_matrix.h", "src/util.h",
       "src/sse_mathfun.h", "src/nnfeatures.c"]
SOURCES = {
    "ref_test_files.npz": ["src/test/raw_signal.crp", "src/test/trimmed_signal.crp", "src/test/normalised_signal.crp", "src/test/path.crp",
                           "src/test/test_matrix.crp"],
    "ref_math.npz": PURE,
    "ref_signal_prep.npz": PURE + ["src/test/raw_signal.crp"],
    "ref_decode.npz": DECODE + ["src/homopolymer.c", "src/scrappie_seq_helpers.c"],
    "ref_events.npz": ["src/nnfeatures.c", "src/nnfeatures.h", "src/util.h", "src/sse_mathfun.h", "src/scrappie_structures.h"],
    "reads/*.i16": ["reads/*.fast5"],
    "fast5/*.fast5": ["reads/*.fast5"],
    "net_f64_*.npz": NET,
}


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for chunk in iter(lambda: fh.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def collect():
    out = {}
    for pat, srcs in sorted(SOURCES.items()):
        files = sorted(glob.glob(os.path.join(HERE, pat)))
        assert files, pat
        ref = {}
        for s in srcs:
            for f in sorted(glob.glob(os.path.join(REF, s))):
                ref[os.path.relpath(f, REF)] = sha(f)
        assert ref, "no reference file for " + pat
        out[pat] = {"fixtures": {os.path.relpath(f, HERE): sha(f) for f in files}, "reference_files": ref}
    return out


def write():
    json.dump(collect(), open(OUT, "w"), indent=1, sort_keys=True)
    return OUT


if __name__ == "__main__":
    print(write())
