"""Multi-process sharding on CPU (gloo, world_size 2): the N>1 path is the
single-GPU path on a slice of the reads plus one gather of results."""
import os
import sys

import numpy as np
import pytest

from scrappie_amd.parallel import length_balanced_order, shard_range, sharded_basecall

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 64, 10000, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_length_balanced_order():
    rng = np.random.RandomState(0)
    lens = rng.randint(1000, 40000, size=500)
    parts = length_balanced_order(lens, 8)
    assert sorted(i for p in parts for i in p) == list(range(500))
    loads = [int(lens[p].sum()) for p in parts]
    assert (max(loads) - min(loads)) / max(loads) < 0.02


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    signals = [np.full(10 + i, i, np.float32) for i in range(37)]
    calls = []

    def fake_basecall(sigs):        # stands in for Engine.basecall on this rank's GPU
        calls.append(len(sigs))
        return [dict(bases="A" * int(s[0] % 5 + 1), n=len(s), rank=rank) for s in sigs]

    out = sharded_basecall(fake_basecall, signals, dist)
    out_bal = sharded_basecall(fake_basecall, signals, dist, balance=True)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, calls, [(o["bases"], o["n"], o["rank"]) for o in out],
           [(o["bases"], o["n"]) for o in out_bal]))


def test_sharded_basecall_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, calls0, out0, bal0), (r1, calls1, out1, bal1) = res
    assert out0 == out1 and bal0 == bal1                       # every rank holds the full result
    assert calls0[0] + calls1[0] == 37 and abs(calls0[0] - calls1[0]) <= 1
    assert [o[1] for o in out0] == [10 + i for i in range(37)]  # original order
    assert [o[2] for o in out0] == [0] * 19 + [1] * 18          # contiguous shards
    assert [b[1] for b in bal0] == [10 + i for i in range(37)]


def test_dynamic_plan_covers_every_read_once_longest_first():
    """scrappie_hip_plan_dynamic (host only): the hand-out plan behind scrappie_hip_basecall_batch_multi, the
    analogue of the reference's `schedule(dynamic)` over reads (scrappie_raw.c:355,387)."""
    import scrappie_amd as sa
    rng = np.random.RandomState(3)
    for n, ne in ((0, 2), (1, 2), (37, 2), (5000, 2), (70000, 8), (100000, 3)):
        ln = rng.randint(1000, 40001, size=n).astype(np.uint32)
        order, starts = sa.plan_dynamic(ln, 5, ne, max_reads=16384, max_blocks=0)
        assert sorted(order.tolist()) == list(range(n))                      # every read exactly once
        sl = ln[order]
        assert np.all(sl[:-1] >= sl[1:])                                      # longest first
        if n == 0:
            assert len(starts) == 0
            continue
        assert starts[0] == 0 and np.all(np.diff(starts) > 0)
        sizes = np.diff(np.append(starts, n))
        assert sizes.max() <= 16384
        # several groups per engine once there is enough work, but none smaller than a GPU's worth unless it is the tail
        if n >= 4096 * ne:
            assert len(starts) >= min(4 * ne, n // 4096) - 1
            assert np.all(sizes[:-1] >= 4096)
    # the device-memory bound cuts groups as well
    ln = np.full(20000, 40000, np.uint32)
    order, starts = sa.plan_dynamic(ln, 5, 2, max_reads=16384, max_blocks=8000 * 300)
    sizes = np.diff(np.append(starts, len(ln)))
    assert np.all(sizes * (8000 // 16 + 1) + 8000 <= 8000 * 300 + 8000 * 17)
    with pytest.raises(RuntimeError):
        sa.plan_dynamic(np.array([10 ** 7], np.uint32), 5, 2, max_blocks=1000)


def test_dynamic_cursor_balances_uneven_engines():
    """Simulate the atomic cursor with engines of different speed: every group is taken exactly once and the
    makespan stays within one group of the ideal (what a static split cannot do)."""
    import scrappie_amd as sa
    rng = np.random.RandomState(4)
    ln = rng.randint(1000, 40001, size=60000).astype(np.uint32)
    order, starts = sa.plan_dynamic(ln, 5, 4)
    bounds = np.append(starts, len(ln)).astype(np.int64)
    cost = np.array([ln[order[bounds[g]:bounds[g + 1]]].astype(np.float64).sum() for g in range(len(starts))], dtype=np.float64)
    speed = np.array([1.0, 1.0, 0.5, 2.0])
    t = np.zeros(4)
    taken = []
    for g in range(len(cost)):                    # the engine that is free first takes the next group
        k = int(np.argmin(t))
        t[k] += cost[g] / speed[k]
        taken.append(g)
    assert taken == list(range(len(cost)))
    ideal = cost.sum() / speed.sum()
    assert t.max() <= ideal + cost.max() / speed.min()
    static = max(cost[k::4].sum() / speed[k] for k in range(4))        # round-robin split for comparison
    assert t.max() < static


def test_plan_tail_picks_the_chain_bound_reads():
    """scrappie_hip_plan_tail (host only): which reads of a call scrappie_hip_basecall_batch runs on its helper engine"""
    import scrappie_amd as sa
    rng = np.random.default_rng(1)
    lens = np.clip(rng.lognormal(np.log(20000), 0.8, size=24000), 1000, 400000).astype(np.uint32)
    f = sa.plan_tail(lens, 5)
    # the long tail: a few per cent of the reads, all of them longer than every read left behind, a small part of the work
    assert 0 < f.sum() < 0.05 * len(lens) and lens[f].min() >= lens[~f].max() and lens[f].sum() < 0.16 * lens.sum()
    assert lens.max() in lens[f]
    # their own chain (11.4 us per block) is longer than half the call at the device's full rate (3.5 ns per block and read)
    T = (lens.astype(np.int64) + 4) // 5
    assert T[f].min() * 11400.0 > T.sum() * 3.5
    # no tail: equal reads, a narrow distribution, a call of long reads only, tiny calls
    assert not sa.plan_tail(np.full(10000, 4000, np.uint32), 5).any()
    assert not sa.plan_tail(rng.integers(1000, 40001, size=32000).astype(np.uint32), 5).any()
    assert not sa.plan_tail(np.full(20, 400000, np.uint32), 5).any()
    assert not sa.plan_tail(np.array([400000], np.uint32), 5).any()
    # a limit on the helper's arena: the longest reads first, whole tiles of 16, within the limit
    g = sa.plan_tail(lens, 5, max_long_blocks=200000)
    assert 0 < g.sum() < f.sum() and lens[g].min() >= lens[~g].max()
    assert sum(T[np.argsort(-T)[:g.sum()]][::16]) <= 200000


def test_committed_bench_line_keeps_the_contract():
    """profiles/r<N>_bench.json (the bench line of the round's final build, as bench.py printed it on the GPU box) carries every field the driver's contract
    names, the roofline and cpu_baseline objects, and is arithmetically consistent; its traffic figure is only there if measured on the recorded sources."""
    import glob
    import json
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "profiles", "r*_bench.json")), key=lambda f: int(re.match(r"r(\d+)_", os.path.basename(f)).group(1)))
    d = json.load(open(files[-1]))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["unit"] == "samples/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert "rgrgr_r94" in d["metric"] and "split products" in d["dtype"]
    per_gpu = d["config"]["reads_per_gpu_per_step"] * d["config"]["samples_per_read"]
    assert abs(d["value"] - d["n_gpus"] * per_gpu / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    assert abs(r["achieved"] - r["flops_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e12) / r["achieved"] < 1e-6
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1
    assert d["cli_end_to_end"]["input"] == "fast5" and d["cli_end_to_end"]["same_sequences_as_f32_input"] is True
    assert "openmp" in d["batch64"] and d["ms_per_step_per_rank"]["max"] >= d["ms_per_step_per_rank"]["min"] > 0

