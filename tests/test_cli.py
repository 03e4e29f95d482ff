"""`scrappie raw` command line (host C over the C ABI) and the signal readers.

BASELINE config 1 ("scrappie raw rgrgr_r94 on the bundled reads") runs here on
the three bundled reads re-encoded as tests/golden/reads/*.i16 (the GPU box has
neither HDF5 nor /root/reference); weights are synthetic (the real model headers
are missing blobs), so this checks plumbing + parity with the oracle, not biology.
"""
import ctypes as C
import glob
import json
import os
import re
import subprocess

import numpy as np
import pytest

import scrappie_amd as sa
from scrappie_amd import model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "scrappie_amd", "scrappie")
READS = os.path.join(ROOT, "tests", "golden", "reads")
META = json.load(open(os.path.join(READS, "reads.json")))


def _reader():
    L = sa.lib()
    L.scrappie_hip_read_raw.restype = sa._RawTable
    L.scrappie_hip_read_raw.argtypes = [C.c_char_p, C.c_bool]
    return L


def _read(path, scale=True):
    rt = _reader().scrappie_hip_read_raw(os.fsencode(path), scale)
    assert rt.raw, path
    a = np.ctypeslib.as_array(rt.raw, shape=(rt.n,)).copy()
    sa._libc.free(C.cast(rt.raw, C.c_void_p))
    return a, rt.uuid


@pytest.fixture(scope="module")
def cli():
    if not os.path.exists(CLI):
        subprocess.run(["make", "-C", os.path.join(ROOT, "scrappie_amd", "csrc"), "all"], check=True,
                       stdout=subprocess.DEVNULL)
    return CLI


def test_i16_fixtures_match_survey_probe():
    """n and first pA sample of each bundled read as read_raw() gives them
    (81106 / 64395 / 29150; 203.361877 / 182.803436 / 67.209435)"""
    for name, m in META.items():
        a, _ = _read(os.path.join(READS, name + ".i16"))
        assert len(a) == m["n"] and a[0] == np.float32(m["first_pA"])
    assert sorted(m["n"] for m in META.values()) == [29150, 64395, 81106]


def test_fast5_reader_equals_fixture():
    files = sorted(glob.glob("/root/reference/reads/*.fast5"))
    if not files or not _reader().scrappie_hip_have_hdf5():
        pytest.skip("needs the reference's bundled fast5 files and an HDF5 library (build container only)")
    for f in files:
        name = os.path.basename(f)[:-6]
        a, uuid = _read(f)
        b, _ = _read(os.path.join(READS, name + ".i16"))
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        assert uuid.decode() == META[name]["uuid"]


FAST5 = os.path.join(ROOT, "tests", "golden", "fast5")


@pytest.fixture
def own_reader(monkeypatch):
    monkeypatch.setenv("SCRAPPIE_FAST5_READER", "own")          # read by the library at every call


def test_builtin_fast5_reader_equals_fixture(own_reader):
    """the reader of the HDF5 subset fast5 files use (csrc/sh_h5mini.c: no libhdf5) on the reference's three bundled reads
    (contiguous Signal; chunked + deflate Signal) and on a re-encoding with 114 shuffled + deflated chunks (two-level chunk
    B-tree): samples in pA bit-equal to read_raw()'s (fast5_interface.c:130-217), read_id, offset / range / digitisation"""
    fallbacks = _reader().sh_h5mini_zlib_fallbacks
    fallbacks.restype = C.c_ulong
    before = fallbacks()
    for name, m in META.items():
        b, _ = _read(os.path.join(READS, name + ".i16"))
        hdr = np.fromfile(os.path.join(READS, name + ".i16"), dtype="<f4", count=3)
        variants = [name + ".fast5"] + (["variant_shuf_gzip_c256.fast5"] if name == "read_ch228_file118" else [])
        for v in variants:
            a, uuid = _read(os.path.join(FAST5, v))
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), v
            assert uuid.decode() == m["uuid"]
            sc = (C.c_float * 3)()
            assert _reader().scrappie_hip_fast5_scaling(os.fsencode(os.path.join(FAST5, v)), sc) == 0
            assert np.array_equal(np.array(sc, dtype=np.float32), hdr)
            counts, _ = _read(os.path.join(FAST5, v), scale=False)        # DAC counts, as read_raw(..., false) gives them
            assert np.array_equal(counts, np.round(counts)) and np.array_equal((counts + hdr[0]) * (hdr[1] / hdr[2]), b)
    # the deflate-compressed chunks (MinKNOW's level 1, chunks of 20 000 samples; 114 shuffled chunks) went through the built-in inflater
    # (sh_inflate.c): zlib, its safety net, was not consulted once
    assert fallbacks() == before


def test_builtin_fast5_reader_equals_libhdf5(monkeypatch):
    if not _reader().scrappie_hip_have_hdf5():
        pytest.skip("no HDF5 library on this box")
    for f in sorted(glob.glob(os.path.join(FAST5, "*.fast5"))):
        if "latest" in f:
            continue
        monkeypatch.setenv("SCRAPPIE_FAST5_READER", "hdf5")
        a, ua = _read(f)
        monkeypatch.setenv("SCRAPPIE_FAST5_READER", "own")
        b, ub = _read(f)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and ua == ub


def test_builtin_fast5_reader_under_address_sanitizer(tmp_path):
    """sh_h5mini.c + sh_inflate.c parse untrusted files.  150 damaged copies (1-4 overwritten bytes, every other time inside the first 4 KiB of structure; one in
    five truncated) of each bundled fast5 through the reader built with -fsanitize=address,undefined: no report; the intact files still read."""
    import shutil
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    csrc = os.path.join(ROOT, "scrappie_amd", "csrc")
    exe = str(tmp_path / "fast5_asan")
    b = subprocess.run(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-std=gnu11", "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
                        os.path.join(ROOT, "tests", "fast5_asan.c"), os.path.join(csrc, "sh_h5mini.c"), os.path.join(csrc, "sh_inflate.c"), "-o", exe, "-ldl", "-lpthread", "-lm"],
                       capture_output=True, text=True, timeout=300)
    if b.returncode != 0 and "asan" in (b.stderr or "").lower():
        pytest.skip("no AddressSanitizer runtime in this toolchain")
    assert b.returncode == 0, b.stderr[-2000:]
    rng = np.random.default_rng(11)
    srcs = [f for f in sorted(glob.glob(os.path.join(FAST5, "*.fast5"))) if "latest" not in f]
    files = []
    for s in srcs:
        good = open(s, "rb").read()
        for it in range(150):
            bb = bytearray(good)
            if it % 5 == 0:
                bb = bb[:int(rng.integers(0, len(bb)))]
            else:
                for _ in range(int(rng.integers(1, 5))):
                    bb[int(rng.integers(0, 4096 if it % 2 else len(bb)))] = int(rng.integers(0, 256))
            files.append(str(tmp_path / ("d%05d.fast5" % len(files))))
            open(files[-1], "wb").write(bytes(bb))
    r = subprocess.run([exe] + srcs + files, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, (r.stdout + r.stderr)[-3000:]
    n, ok = (int(x) for x in re.findall(r"\d+", r.stdout)[:2])
    assert n == len(srcs) + len(files) and ok >= len(srcs)


def test_builtin_fast5_reader_refuses_what_it_does_not_parse(own_reader, tmp_path, capfd):
    """a file written with the newer format (superblock v2, new-style groups) is refused with a message, never misread; a
    truncated or damaged file gives no read (and no crash)"""
    rt = _reader().scrappie_hip_read_raw(os.fsencode(os.path.join(FAST5, "variant_latest.fast5")), True)
    assert not rt.raw and rt.n == 0
    assert "needs libhdf5" in capfd.readouterr().err
    good = open(os.path.join(FAST5, "variant_shuf_gzip_c256.fast5"), "rb").read()
    want, _ = _read(os.path.join(FAST5, "variant_shuf_gzip_c256.fast5"))
    rng = np.random.default_rng(5)
    f = str(tmp_path / "damaged.fast5")
    n_ok = 0
    for it in range(150):
        b = bytearray(good)
        if it % 5 == 0:
            b = b[:int(rng.integers(0, len(b)))]
        else:
            for _ in range(int(rng.integers(1, 4))):
                at = int(rng.integers(0, 4096 if it % 2 else len(b)))
                b[at] = int(rng.integers(0, 256))
        open(f, "wb").write(bytes(b))
        rt = _reader().scrappie_hip_read_raw(os.fsencode(f), True)
        if rt.raw:
            n_ok += 1
            assert rt.n > 0
            sa._libc.free(C.cast(rt.raw, C.c_void_p))
    capfd.readouterr()
    assert n_ok < 150


def test_cli_plumbing_without_gpu(cli):
    r = subprocess.run([cli, "version"], capture_output=True, text=True)
    assert r.returncode == 0 and "scrappie" in r.stdout
    r = subprocess.run([cli, "events", "x"], capture_output=True, text=True)
    assert r.returncode != 0 and "not part of this build" in r.stderr
    r = subprocess.run([cli, "raw", "--model", "bogus", READS], capture_output=True, text=True)
    assert r.returncode != 0 and "Invalid model" in r.stderr
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([cli, "raw", READS], capture_output=True, text=True)
        assert r.returncode != 0 and "HIP device" in r.stderr      # fails loudly, no CPU fallback


@pytest.mark.gpu
def test_config1_from_the_fast5_files_themselves(cli, tmp_path):
    """BASELINE config 1 on the reference's bundled fast5 files as they are (no libhdf5 needed: built-in reader), against the
    run on their .i16 re-encodings that test_config1_bundled_reads_vs_oracle checks against the oracle: identical calls,
    scores and trims; --uuid names the records by read_id (scrappie_raw.c:285-300)"""
    w = model.synthetic_model("rgrgr_r94", seed=1)
    mfile = str(tmp_path / "rgrgr_r94.scrm")
    model.save_model(w, mfile)
    base = [cli, "raw", "--model", "rgrgr_r94", "--model-file", mfile, "--local", "150"]
    env = dict(os.environ, SCRAPPIE_FAST5_READER="own")
    files = [os.path.join(FAST5, n + ".fast5") for n in META]
    a = subprocess.run(base + ["--uuid"] + files, capture_output=True, text=True, env=env)
    b = subprocess.run(base + [os.path.join(READS, n + ".i16") for n in META], capture_output=True, text=True, env=env)
    assert a.returncode == 0 and b.returncode == 0, a.stderr + b.stderr
    ra = {FASTA_RE.match(h).group(2)[:-6]: (FASTA_RE.match(h).groups(), q) for h, q in zip(a.stdout.split("\n")[0::2], a.stdout.split("\n")[1::2])}
    rb = {FASTA_RE.match(h).group(2)[:-4]: (FASTA_RE.match(h).groups(), q) for h, q in zip(b.stdout.split("\n")[0::2], b.stdout.split("\n")[1::2])}
    assert sorted(ra) == sorted(rb) == sorted(META)
    for name, m in META.items():
        ga, qa = ra[name]
        gb, qb = rb[name]
        assert qa == qb and len(qa) > 1000
        assert ga[3:] == gb[3:]                              # score, nblock, length, blocks per base, nsample, trim
        assert ga[0] == m["uuid"] and ga[2] == m["uuid"]
    # a directory argument means dir/*.fast5: the five files there, one of which the built-in reader refuses
    d = subprocess.run(base + [FAST5], capture_output=True, text=True, env=env)
    assert d.returncode == 0 and d.stdout.count(">") == 4 and "needs libhdf5" in d.stderr


@pytest.mark.gpu
def test_one_hip_runtime_whatever_the_import_order():
    """scrappie_amd before torch used to leave two HIP runtimes in the process (torch's bundled copy asks for "libamdhip64.so", which
    does not match the soname of the /opt/rocm copy this library had brought in) and the one initialised second saw no device;
    scrappie_amd.lib() now binds to torch's copy when a torch is installed.  With SCRAPPIE_HIP_SYSTEM_RUNTIME=1 the old
    behaviour stays and the error names the two copies."""
    prog = ("import scrappie_amd as sa\n"
            "L = sa.lib()\n"
            "import torch\n"
            "print('torch', torch.cuda.is_available())\n"
            "try:\n"
            "    e = sa.Engine(0); print('engine ok'); e.close()\n"
            "except RuntimeError as err:\n"
            "    print('engine failed:', err)\n")
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run(["python", "-c", prog], capture_output=True, text=True, env=env, cwd=ROOT)
    assert "torch True" in r.stdout and "engine ok" in r.stdout, r.stdout + r.stderr
    r = subprocess.run(["python", "-c", prog], capture_output=True, text=True, env=dict(env, SCRAPPIE_HIP_SYSTEM_RUNTIME="1"), cwd=ROOT)
    if "engine failed" in r.stdout:
        assert "two HIP runtimes are loaded" in r.stdout, r.stdout


@pytest.mark.gpu
def test_reference_loop_body_under_openmp_is_a_drop_in(tmp_path):
    """tests/drop_in_loop.c: the reference's per-read loop (scrappie_raw.c:265-315 under the OpenMP loop of :355,387) written with the reference's
    function names against this library's header.  24 reads over 24 threads (the calls coalesce into a few launches) give, line for line, what one
    thread with one read per launch gives, and the batched engine's calls."""
    exe = str(tmp_path / "drop_in_loop")
    subprocess.run(["gcc", "-O2", "-std=c99", "-fopenmp", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "drop_in_loop.c"), "-o", exe,
                    "-L", os.path.join(ROOT, "scrappie_amd"), "-lscrappie_hip", "-Wl,-rpath," + os.path.join(ROOT, "scrappie_amd"), "-lm"], check=True)
    w = model.synthetic_model("rgrgr_r94", seed=1)
    model.save_model(w, str(tmp_path / "rgrgr_r94.scrm"))
    files = [os.path.join(READS, n + ".i16") for n in META]
    env = dict(os.environ, SCRAPPIE_MODEL_DIR=str(tmp_path))
    many = subprocess.run([exe, "rgrgr_r94", "150", "8"] + files, capture_output=True, text=True, env=dict(env, OMP_NUM_THREADS="24"))
    one = subprocess.run([exe, "rgrgr_r94", "150", "8"] + files, capture_output=True, text=True, env=dict(env, OMP_NUM_THREADS="1", SCRAPPIE_HIP_COALESCE="0"))
    assert many.returncode == 0 and one.returncode == 0, many.stderr + one.stderr
    la, lb = many.stdout.strip().split("\n"), one.stdout.strip().split("\n")
    assert len(la) == 24 and la == lb
    assert all(len(l.split()[3]) > 1000 for l in la)
    # the same reads through the batched engine
    eng = sa.Engine(0)
    eng.load_model("rgrgr_r94", w)
    sigs = []
    for i in range(24):
        raw, _ = _read(files[i % 3])
        rt = sa.RawTable(raw[:len(raw) - (i // 3) * 37].copy())
        rt.trim().scale()                                  # trim_and_segment_raw + medmad_normalise_array, as the loop does
        sigs.append(rt.data(as_numpy=True))
    calls = eng.basecall(sigs, "rgrgr_r94", eng.default_params(local_pen=150.0))
    eng.close()
    same = sum(c["bases"] == l.split()[3] and abs(c["score"] - float(l.split()[1])) <= 1e-3 * abs(c["score"]) for c, l in zip(calls, la))
    assert same >= 22, same          # (a near tie between the fused and the two-kernel decoder's last bits may move a call: see test_config1)


FASTA_RE = re.compile(
    r'^>(\S*)  \{ "filename" : "([^"]*)", "uuid" : "([^"]*)", "normalised_score" : ([-0-9.]+),  "nblock" : (\d+),  '
    r'"sequence_length" : (\d+),  "blocks_per_base" : ([-0-9.a-z]+), "nsample" : (\d+), "trim" : \[ (\d+), (\d+) \] \}$')


@pytest.mark.gpu
@pytest.mark.parametrize("local_pen", [None, 150.0])
def test_config1_bundled_reads_vs_oracle(cli, orc, tmp_path, local_pen):
    """BASELINE config 1: `scrappie raw` on the three bundled reads against the oracle's whole path
    (scrappie_raw.c:265-315).  With --local 150 the start state is expensive and the calls are thousands of bases
    long (the default decodes random weights to a handful).  A call must be IDENTICAL to the oracle's -- unless it
    is, bit for bit, the oracle's decode of the engine's own posterior of that read, i.e. the difference is a near
    tie decided by the last bits of the two posteriors."""
    import ctypes as C
    w = model.synthetic_model("rgrgr_r94", seed=1)
    mfile = str(tmp_path / "rgrgr_r94.scrm")
    model.save_model(w, mfile)
    extra = [] if local_pen is None else ["--local", str(local_pen)]
    r = subprocess.run([cli, "raw", "--model", "rgrgr_r94", "--model-file", mfile, "--prefix", "p_"] + extra + [READS],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().split("\n")
    assert len(lines) == 6
    om = orc.OracleModel(w)
    eng = sa.Engine(0)
    eng.load_model("rgrgr_r94", w)
    seen = set()
    same = explained = 0
    for hdr, seq in zip(lines[0::2], lines[1::2]):
        m = FASTA_RE.match(hdr)
        assert m, hdr
        rid, fname, uuid, nscore, nblock, slen, bpb, nsample, t0, t1 = m.groups()
        name = fname[:-4]
        seen.add(name)
        assert rid == "p_" + fname and uuid == ""          # .i16 carries no uuid; --no-uuid is the default
        raw, _ = _read(os.path.join(READS, fname))
        p = orc.lib().orc_default_params()
        if local_pen is not None:
            p.local_pen = local_pen
        o = orc.basecall_raw(om, raw, p)                    # scrappie_raw.c:265-315
        assert int(nsample) == META[name]["n"]
        assert (int(t0), int(t1)) == (o["start"], o["end"]) and int(nblock) == o["nblock"]
        assert int(slen) == len(seq)
        assert abs(float(nscore) - (-o["score"] / o["nblock"])) <= 1e-3 * max(1.0, abs(o["score"] / o["nblock"]))
        assert abs(float(bpb) - int(nblock) / max(1, len(seq))) < 1e-3
        if local_pen is not None:
            assert len(seq) > 0.3 * int(nblock)             # a real path through the k-mer states
        if seq == o["bases"]:
            same += 1
            continue
        # GPU and CPU posteriors differ in the last bits: the call may differ at a near tie, and then it must be the
        # decode (oracle, bit-exact) of the engine's own posterior of the same trimmed, normalised signal
        x = np.ascontiguousarray(raw[o["start"]:o["end"]], dtype=np.float32).copy()
        sa.lib().medmad_normalise_array(x.ctypes.data_as(C.POINTER(C.c_float)), len(x))
        post = eng.posterior(x, "rgrgr_r94", min_prob=1e-5)
        wsc, wseq = orc.decode_transducer(post, 0.0, 0.0, 2.0 if local_pen is None else local_pen, False)
        rc, wseq = orc.homopolymer_path(post, wseq)
        wb, _ = orc.overlapper(wseq, 1024)
        assert seq == wb, "the call is neither the oracle's nor the decode of the engine's own posterior"
        explained += 1
    eng.close()
    print("config 1 (--local %s): %d calls identical to the oracle's, %d differ at near ties" % (local_pen, same, explained))
    assert seen == set(META) and same + explained == 3
    # trims for the three reads as the survey measured them
    assert sorted(int(FASTA_RE.match(h).group(5)) for h in lines[0::2]) == [5778, 12818, 16158]


@pytest.mark.gpu
def test_cli_options(cli, tmp_path):
    w = model.synthetic_model("rnnrf_r94", seed=2)
    mdir = tmp_path
    model.save_model(w, str(mdir / "rnnrf_r94.scrm"))
    env = dict(os.environ, SCRAPPIE_MODEL_DIR=str(mdir))
    one = os.path.join(READS, "read_ch228_file118.i16")
    out = str(tmp_path / "o.sam")
    r = subprocess.run([cli, "raw", "--model", "rnnrf_r94", "-f", "SAM", "-o", out, "-l", "1", "-t", "100:20", one, READS],
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    rec = open(out).read().strip().split("\n")
    assert len(rec) == 1                                   # --limit 1
    f = rec[0].split("\t")
    assert f[0] == "read_ch228_file118.i16" and f[1] == "4" and len(f[9]) > 500 and set(f[9]) <= set("ACGT")
    r = subprocess.run([cli, "raw", "--model", "rnnrf_r94", str(tmp_path / "nothing_here")], capture_output=True, text=True, env=env)
    assert "does not exist or no fast5 files found" in r.stderr
    # --batch 1: three batches, the loader thread prefetches batch k+1 while batch k is on the GPU;
    # the records must equal those of one batch of three
    outs = []
    for extra in (["--batch", "1"], []):
        r = subprocess.run([cli, "raw", "--model", "rnnrf_r94", "-#", "1"] + extra + [READS], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        outs.append(sorted(r.stdout.strip().split("\n>")))
    assert outs[0] == outs[1] and len(outs[0]) == 3
    # --devices 0,0: two engines, launch groups handed out dynamically; same records
    r = subprocess.run([cli, "raw", "--model", "rnnrf_r94", "-#", "1", "--devices", "0,0", READS], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert sorted(r.stdout.strip().split("\n>")) == outs[1]


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu():
    """bench.py's multi-rank path (sharding, barriers, max-over-ranks time, summed bases, one JSON
    line from rank 0) with two ranks sharing cuda:0 through the BENCH_BACKEND / BENCH_DEVICE test
    hooks (the driver runs it with one rank per GPU over RCCL)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_BACKEND="gloo", BENCH_DEVICE="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(root, "bench.py"),
           "--gpus", "2", "--steps", "1", "--warmup", "1", "--reads", "400", "--samples", "2000", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 1
    assert d["value"] > 0 and abs(d["value"] - 2 * 400 * 2000 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert d["roofline"]["frac"] > 0 and "cpu_baseline" not in d


@pytest.mark.gpu
def test_bench_eight_ranks_on_one_gpu():
    """bench.py as the driver launches it on a node -- torchrun, eight ranks -- with the eight ranks sharing cuda:0 (gloo for the
    barrier and the two reductions: RCCL wants a device per rank).  Eight engines, eight sets of streams and pinned staging
    in eight processes, the host's CPUs divided by LOCAL_WORLD_SIZE; one JSON line with the weak-scaling arithmetic."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_BACKEND="gloo", BENCH_DEVICE="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
           "--master-addr", "127.0.0.1", "--master-port", "29547", os.path.join(root, "bench.py"),
           "--gpus", "8", "--steps", "2", "--warmup", "1", "--reads", "256", "--samples", "1200", "--no-cpu-baseline", "--no-extra"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["steps"] == 2
    assert d["value"] > 0 and abs(d["value"] - 8 * 256 * 1200 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert d["roofline"]["frac"] > 0 and "cpu_baseline" not in d
    # a SCALE run must show skew between ranks, not only the slowest
    pr = d["ms_per_step_per_rank"]
    assert len(pr["all"]) == 8 and pr["max"] == max(pr["all"]) and pr["min"] == min(pr["all"]) and pr["min"] > 0
    assert abs(pr["max"] - d["ms_per_step"]) / d["ms_per_step"] < 1e-6
    assert d["numa"]["local_world_size"] == 8 and d["numa"]["host_threads_per_engine"] >= 1


@pytest.mark.gpu
def test_bench_process_group_over_rccl_on_one_gpu():
    """The path the driver's multi-GPU runs take -- torch.distributed with backend nccl (= RCCL), its communicator and streams in
    the process, the barrier and the two all-reduces on the device -- with a single rank (BENCH_FORCE_DIST=1); the gloo
    two-rank test above never touches RCCL.  Asserts the contract's JSON line."""
    import subprocess
    import sys
    env = dict(os.environ, BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.pop("BENCH_BACKEND", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--reads", "1024",
           "--samples", "2000", "--no-cpu-baseline", "--no-extra"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["scaling"] == "weak" and d["unit"] == "samples/s" and d["higher_is_better"] is True
    assert "rgrgr_r94" in d["metric"] and d["dtype"].startswith("f32") and "split products" in d["dtype"] and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] - 1024 * 2000 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert 0 < d["roofline"]["frac"] < 1 and d["roofline"]["bound"] == "mfma" and d["roofline"]["whole_step_frac"] > 0


@pytest.mark.gpu
def test_cli_several_engines_over_thousands_of_mixed_reads(cli, tmp_path):
    """`scrappie raw --devices 0,0,0` (three engines, launch groups handed out from the atomic cursor) over a few thousand reads of
    mixed lengths, in several batches: the set of records equals the single-engine run's, order aside (the reference's OpenMP
    loop writes records in completion order too: scrappie_raw.c:377,402)."""
    rng = np.random.default_rng(7)
    w = model.synthetic_model("rgrgr_r94", seed=3)
    model.save_model(w, str(tmp_path / "rgrgr_r94.scrm"))
    rdir = tmp_path / "reads"
    rdir.mkdir()
    n = 2600
    lens = np.clip(rng.lognormal(np.log(3000), 0.7, size=n), 300, 40000).astype(int)
    base = (90.0 + 12.0 * np.repeat(rng.standard_normal(6000), 9)[:41000] + 1.5 * rng.standard_normal(41000)).astype(np.float32)
    for i in range(n):
        s = int(rng.integers(0, 41000 - lens[i]))
        base[s:s + lens[i]].tofile(str(rdir / ("r%04d.f32" % i)))
    env = dict(os.environ, SCRAPPIE_MODEL_DIR=str(tmp_path))
    outs = []
    # (--devices 0,0,0,0,0,0,0,0: as many engines in one process as a node has GPUs -- eight sets of streams, arenas and pinned
    # staging, eight host threads taking launch groups from the one cursor; the hardware the round had is one GPU)
    for extra in ([], ["--devices", "0,0,0", "--batch", "500"], ["--devices", "0,0,0,0,0,0,0,0", "--batch", "1300", "--prep", "host"]):
        r = subprocess.run([cli, "raw", "--model", "rgrgr_r94", "--local", "150", "-#", "2"] + extra + [str(rdir)], capture_output=True, text=True,
                           env=env, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        recs = sorted(("\n" + r.stdout).split("\n>")[1:])
        outs.append(recs)
    assert n - 20 <= len(outs[0]) <= n and outs[0] == outs[1] and outs[0] == outs[2]      # (a few reads are too short once trimmed: no record, in any run)
    nb = sum(len(x.split("\n", 1)[1].replace("\n", "")) for x in outs[0])
    assert nb > 0.3 * sum((l + 4) // 5 for l in lens) * 0.5
