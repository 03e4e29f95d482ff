"""Multi-GPU: reads are the shard unit (the reference's only parallel axis is
the OpenMP loop over reads, src/scrappie_raw.c:355,387).  One process per GPU;
each rank basecalls a contiguous slice of the read list on its own engine, with
NO collective on the data path (weights are replicated, activations never leave
the GPU).  torch.distributed (RCCL on GPUs, gloo in CPU tests) is used only to
gather the per-read results / counters at the end.
"""


def shard_range(n_items, world, rank):
    """Contiguous, balanced [lo, hi) slice of n_items for `rank` of `world`;
    the first (n_items % world) ranks get one extra item."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def length_balanced_order(lengths, world):
    """Deal reads to ranks so every rank gets a similar total number of samples
    (mixed-length read sets: BASELINE config 3).  Returns a list of index lists,
    one per rank; longest-first greedy."""
    order = sorted(range(len(lengths)), key=lambda i: -int(lengths[i]))
    loads = [0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: loads[k])
        out[r].append(i)
        loads[r] += int(lengths[i])
    for lst in out:
        lst.sort()
    return out


def sharded_basecall(basecall_fn, signals, dist=None, balance=False):
    """Run `basecall_fn(list_of_signals) -> list_of_results` on this rank's shard
    and return the full, ordered result list on every rank.

    `dist` is torch.distributed (initialised) or None for a single process.  The
    only communication is one all_gather_object of the (small) results."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return basecall_fn(list(signals))
    world, rank = dist.get_world_size(), dist.get_rank()
    n = len(signals)
    if balance:
        mine = length_balanced_order([len(s) for s in signals], world)[rank]
    else:
        lo, hi = shard_range(n, world, rank)
        mine = list(range(lo, hi))
    local = basecall_fn([signals[i] for i in mine])
    gathered = [None] * world
    dist.all_gather_object(gathered, (mine, local))
    out = [None] * n
    for idx, res in gathered:
        for i, r in zip(idx, res):
            out[i] = r
    return out
