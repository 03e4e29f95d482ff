#!/usr/bin/env python3
"""BASELINE config 3 as a rate: rgrgr_r10-shaped model, mixed-length reads N ~ U{lo..hi}, inputs
resident in HBM, launch groups pipelined as in bench.py (a DESIGN.md note, not bench.py's `value`).
usage: mixed_rate.py [reads=3000] [lo=1000] [hi=40000] [model=rgrgr_r10] [steps=4] [warmup=2]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scrappie_amd as sa
from scrappie_amd import model, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 40000
name = sys.argv[4] if len(sys.argv) > 4 else "rgrgr_r10"
rng = np.random.default_rng(1)
if os.environ.get("MIXED_LOGNORMAL"):      # "median,sigma": a long-tailed length distribution clipped to [lo, hi]
    med, sg = (float(v) for v in os.environ["MIXED_LOGNORMAL"].split(","))
    lens = np.clip(rng.lognormal(np.log(med), sg, size=n), lo, hi).astype(np.uint32)
else:
    lens = rng.integers(lo, hi + 1, size=n).astype(np.uint32)
long_sig = synth.medmad_normalise(synth.synthetic_signal(hi + 64 * 7, 5))
off = np.zeros(n, np.uint64)
off[1:] = np.cumsum(lens[:-1].astype(np.uint64))
flat = np.empty(int(lens.sum()), np.float32)
for i in range(n):                       # every read a different window of one long synthetic signal
    s = (i % 64) * 7
    flat[int(off[i]):int(off[i]) + int(lens[i])] = long_sig[s:s + int(lens[i])]
eng = sa.Engine(0)
eng.load_model(name, model.synthetic_model(name, seed=1))
eng.set_profiling(True)
if int(os.environ.get("MIXED_ENGINES", "1")) == 1:     # (several engines: the arena belongs to the host-to-host leg)
    d = eng.upload(flat)
    steps = int(sys.argv[5]) if len(sys.argv) > 5 else 4
    warm = int(sys.argv[6]) if len(sys.argv) > 6 else 2      # both slots allocate their buffers on first use
    for k in range(warm):
        eng.run_device(d, off, lens, name); eng.collect(n, raw=True)
    eng.synchronize()
    t0 = time.perf_counter()
    eng.run_device(d, off, lens, name)
    for k in range(1, steps):
        eng.run_device(d, off, lens, name); eng.collect(n, raw=True)
    eng.collect(n, raw=True)
    dt = (time.perf_counter() - t0) / steps
    t = eng.timing()
    print("%s, %d reads %s{%d..%d} (%.1f M samples per group, longest %d): %.1f ms per group -> %.3e samples/s" %
          (name, n, "lognormal(%s) in " % os.environ["MIXED_LOGNORMAL"] if os.environ.get("MIXED_LOGNORMAL") else "U", lo, hi,
           lens.sum() / 1e6, lens.max(), dt * 1e3, lens.sum() / dt))
    print("stages of the last group, ms: " + ", ".join("%s %.2f" % (f, t[f]) for f in
          ("conv_ms", "affine_ms", "gru_ms", "ff_ms", "decode_ms", "backtrace_ms", "stitch_ms", "total_ms")))
    eng.free(d)
if os.environ.get("MIXED_H2H", "1") != "0":
    # the same reads, twice over, host memory to host memory through ONE scrappie_hip_basecall_batch call: the engine's
    # own planner cuts the launch groups (as large as the arena allows) and keeps two in flight
    import ctypes as C
    L = sa.lib()
    G = 2
    rts = (sa._RawTable * (n * G))()
    for g in range(G):
        for i in range(n):
            rts[g * n + i] = sa._RawTable(None, int(lens[i]), 0, int(lens[i]), C.cast(flat.ctypes.data + 4 * int(off[i]), C.POINTER(C.c_float)))
    calls = (sa._Call * (n * G))()
    params = eng.default_params()
    K = int(os.environ.get("MIXED_ENGINES", "1"))      # K engines on ONE device: launch groups of a call side by side
    engs = [eng]
    if K > 1:
        for k in range(1, K):
            e2 = sa.Engine(0)
            e2.load_model(name, model.synthetic_model(name, seed=1))
            engs.append(e2)
        cap = int(os.environ.get("MIXED_CAP", "5400000")) // K
        for e2 in engs:
            e2.set_max_launch_blocks(cap)
    hs = (C.c_void_p * K)(*[e2._h for e2 in engs])
    ms = (C.c_int * K)(*[e2._models[name] for e2 in engs])
    for rep in range(int(os.environ.get("MIXED_REPS", "4"))):      # the arena grows on the first calls (both slots); the last call is reported
        t0 = time.perf_counter()
        if L.scrappie_hip_basecall_batch_multi(hs, ms, K, rts, n * G, C.byref(params), calls) != 0:
            raise RuntimeError(sa.last_error())
        dt = time.perf_counter() - t0
        L.scrappie_hip_free_calls(calls, n * G)
    print("host to host, %d reads in one call, %d engine(s) on the device: %.1f ms -> %.3e samples/s" % (n * G, K, dt * 1e3, G * lens.sum() / dt))
if int(os.environ.get("MIXED_STREAM", "0")) > 0:
    # a STREAM of calls (what `scrappie raw` does with its batches): every call through scrappie_hip_basecall_batch_deferred, so the
    # chain-bound reads of call k run beside calls k+1, k+2 ... and the helper engine takes all that are waiting as one launch group
    import ctypes as C
    L = sa.lib()
    K = int(os.environ["MIXED_STREAM"])
    L.scrappie_hip_basecall_batch_deferred.restype = C.c_long
    L.scrappie_hip_basecall_batch_deferred.argtypes = [C.c_void_p, C.c_int, C.POINTER(sa._RawTable), C.c_size_t, C.POINTER(sa.Params), C.POINTER(sa._Call), C.POINTER(C.c_ubyte)]
    L.scrappie_hip_deferred_collect.restype = C.c_long
    L.scrappie_hip_deferred_collect.argtypes = [C.c_void_p, C.c_long, C.POINTER(sa._Call), C.c_size_t, C.c_int]
    rts = (sa._RawTable * n)()
    for i in range(n):
        rts[i] = sa._RawTable(None, int(lens[i]), 0, int(lens[i]), C.cast(flat.ctypes.data + 4 * int(off[i]), C.POINTER(C.c_float)))
    calls = (sa._Call * n)()
    lcalls = (sa._Call * n)()
    flags = (C.c_ubyte * n)()
    params = eng.default_params()
    for rep in range(2):          # the first pass warms both engines' arenas
        tickets = []
        nb = 0
        t0 = time.perf_counter()
        for k in range(K):
            tk = L.scrappie_hip_basecall_batch_deferred(eng._h, eng._models[name], rts, n, C.byref(params), calls, flags)
            if tk < 0:
                raise RuntimeError(sa.last_error())
            L.scrappie_hip_free_calls(calls, n)
            if tk > 0:
                tickets.append((tk, int(sum(flags))))
        t_main = time.perf_counter() - t0
        for tk, nl in tickets:
            if L.scrappie_hip_deferred_collect(eng._h, tk, lcalls, nl, 1) != nl:
                raise RuntimeError(sa.last_error())
            L.scrappie_hip_free_calls(lcalls, nl)
        dt = time.perf_counter() - t0
    ng = int(eng.debug_fetch("n_tail_groups", np.uint64)[0])
    print("stream of %d calls of %d reads (deferred chain-bound reads: %d per call; the helper ran %d launch groups in all): %.1f ms "
          "(%.1f until the last call returned) -> %.3e samples/s" % (K, n, tickets[0][1] if tickets else 0, ng, dt * 1e3, t_main * 1e3, K * lens.sum() / dt))
if int(os.environ.get("MIXED_DSTREAM", "0")) > 0:
    # the same stream with the signals already on the device (what `scrappie raw --prep=device` does): streaming calls
    # (scrappie_hip_basecall_device_deferred_stream), the engine's pipeline does not drain between calls
    import ctypes as C
    L = sa.lib()
    K = int(os.environ["MIXED_DSTREAM"])
    fn = L.scrappie_hip_basecall_device_deferred_stream
    fn.restype = C.c_long
    fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(sa.Params), C.POINTER(sa._Call), C.POINTER(C.c_ubyte)]
    L.scrappie_hip_stream_flush.argtypes = [C.c_void_p]
    L.scrappie_hip_deferred_collect.restype = C.c_long
    L.scrappie_hip_deferred_collect.argtypes = [C.c_void_p, C.c_long, C.POINTER(sa._Call), C.c_size_t, C.c_int]
    d = eng.upload(flat)
    params = eng.default_params()
    lcalls = (sa._Call * n)()
    flags = (C.c_ubyte * n)()
    offp, lenp = off.ctypes.data_as(C.POINTER(C.c_uint64)), lens.ctypes.data_as(C.POINTER(C.c_uint32))
    for rep in range(2):          # the first pass warms both engines' arenas
        outs = [(sa._Call * n)() for k in range(K)]
        tickets = []
        t0 = time.perf_counter()
        for k in range(K):
            tk = fn(eng._h, eng._models[name], d, offp, lenp, n, C.byref(params), outs[k], flags)
            if tk < 0:
                raise RuntimeError(sa.last_error())
            if tk > 0:
                tickets.append((tk, int(sum(flags))))
        if L.scrappie_hip_stream_flush(eng._h) != 0:
            raise RuntimeError(sa.last_error())
        t_main = time.perf_counter() - t0
        for tk, nl in tickets:
            if L.scrappie_hip_deferred_collect(eng._h, tk, lcalls, nl, 1) != nl:
                raise RuntimeError(sa.last_error())
            L.scrappie_hip_free_calls(lcalls, nl)
        dt = time.perf_counter() - t0
        ncalled = sum(1 for o in outs for c in o if c.basecall_length > 0)
        for o in outs:
            L.scrappie_hip_free_calls(o, n)
    print("device-resident stream of %d calls of %d reads (deferred: %d per call; %d reads called in the calls themselves): %.1f ms "
          "(%.1f until the last call's groups were delivered) -> %.3e samples/s" % (K, n, tickets[0][1] if tickets else 0, ncalled, dt * 1e3, t_main * 1e3, K * lens.sum() / dt))
    eng.free(d)
