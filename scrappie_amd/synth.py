"""Seeded synthetic inputs for tests and bench (there is no network for real
fast5 data, and the bundled reads need HDF5).

* `synthetic_signal`   -- squiggle-like raw signal:
  piecewise-constant levels ~N(0,1), dwell ~Geometric(mean 9 samples), plus
  N(0, 0.1^2) noise.
* `simulated_posterior` -- transducer log-posteriors generated from a simulated
  k-mer path (about 55 % stay / 40 % step / 5 % skip), because i.i.d. random
  posteriors decode to an all-stay path (SURVEY.md section 8c fixture note).
  Homopolymer runs with ambiguous stay/repeat mass can be planted to exercise the
  homopolymer correction (homopolymer.c:175).
"""
import numpy as np


def synthetic_signal(n, seed, mean_dwell=9.0, noise=0.1, raw_units=False):
    rng = np.random.RandomState(seed)
    nlev = int(n / mean_dwell * 1.5) + 16
    dwell = rng.geometric(1.0 / mean_dwell, size=nlev)
    while dwell.sum() < n:
        dwell = np.concatenate([dwell, rng.geometric(1.0 / mean_dwell, size=nlev)])
    levels = rng.normal(0.0, 1.0, size=len(dwell))
    sig = np.repeat(levels, dwell)[:n] + rng.normal(0.0, noise, size=n)
    if raw_units:                      # pA-like scale so trimming/normalising is exercised
        sig = 90.0 + 12.0 * sig
    return sig.astype(np.float32)


def medmad_normalise(x):
    """numpy statement of util.c:190-205 (float32 arithmetic)."""
    x = np.asarray(x, dtype=np.float32)

    def quant(v, p):
        s = np.sort(v)
        pos = np.float32(p) * np.float32(len(v) - 1)
        idx = int(pos)
        rem = np.float32(pos - np.float32(idx))
        if idx < len(v) - 1:
            return np.float32((1.0 - float(rem)) * float(s[idx]) + float(np.float32(rem * s[idx + 1])))
        return s[idx]

    med = quant(x, 0.5)
    mad = np.float32(quant(np.abs(x - med), 0.5) * np.float32(1.4826))
    return ((x - med) / mad).astype(np.float32)


def simulated_posterior(T, seed, klen=5, p_stay=0.55, p_skip=0.05, min_prob=1e-5,
                        plant_homopolymers=0, log=True):
    """Returns (logpost[T, 4**klen + 1] float32, true_path[T]).

    State order is the reference's: k-mers 0..4**k-1 (oldest base most
    significant), stay LAST (decode.c:129-131)."""
    rng = np.random.RandomState(seed)
    nk = 4 ** klen
    NS = nk + 1
    post = rng.dirichlet(np.full(NS, 0.02), size=T).astype(np.float64)
    path = np.full(T, -1, dtype=np.int64)
    kmer = int(rng.randint(nk))
    t = 0
    homo_left = plant_homopolymers
    while t < T:
        r = rng.rand()
        if homo_left > 0 and t > 10 and rng.rand() < 0.02 and t + 12 < T:
            # plant: ...XYYYY then a run of YYYYY/stay with ambiguous mass
            b = int(rng.randint(4))
            x = (b + 1 + int(rng.randint(3))) % 4
            hk = sum(b * 4 ** i for i in range(klen))
            enter = (x * 4 ** (klen - 1)) + (hk % 4 ** (klen - 1))
            path[t] = enter
            runlen = int(rng.randint(3, 9))
            for j in range(1, runlen + 1):
                if t + j >= T:
                    break
                amb = rng.uniform(0.25, 0.75)
                post[t + j] *= 0.05
                post[t + j, hk] += amb * 0.9
                post[t + j, nk] += (1 - amb) * 0.9
                path[t + j] = -2        # marker: ambiguous
            kmer = hk
            t += runlen + 1
            homo_left -= 1
            continue
        if r < p_stay:
            path[t] = -1
        elif r < p_stay + p_skip:
            kmer = ((kmer * 16) % nk) + int(rng.randint(16))
            path[t] = kmer
        else:
            kmer = ((kmer * 4) % nk) + int(rng.randint(4))
            path[t] = kmer
        t += 1
    for t in range(T):
        if path[t] == -2:
            continue
        mass = rng.uniform(0.35, 0.95)
        s = nk if path[t] < 0 else path[t]
        post[t] *= (1.0 - mass)
        post[t, s] += mass
    post /= post.sum(axis=1, keepdims=True)
    post = post.astype(np.float32)
    if log:
        post = np.log(np.float32(min_prob) + np.float32(1.0 - min_prob) * post).astype(np.float32)
    return post, path


def simulated_crf_transitions(T, seed):
    """Random 5x5 transition energies per block (25, T) -> (T, 25) float32,
    mildly structured so the Viterbi path emits ~0.5 bases per block."""
    rng = np.random.RandomState(seed)
    tr = rng.normal(0.0, 1.0, size=(T, 25)).astype(np.float32)
    tr[:, 24] += 1.0          # stay->stay favoured
    base = rng.randint(0, 4, size=T)
    for t in range(T):
        if rng.rand() < 0.45:
            tr[t, base[t] * 5:(base[t] + 1) * 5] += 3.0
    return tr


# scrappie_structures.h:8-15 (event_t): uint64 start; float length, mean, stdv; int pos, state
EVENT_DTYPE = np.dtype([("start", np.uint64), ("length", np.float32), ("mean", np.float32),
                        ("stdv", np.float32), ("pos", np.int32), ("state", np.int32)], align=True)


def synthetic_events(n, seed, mean_dwell=9.0):
    """Seeded event table: a level per event from a random k-mer-like walk, dwell ~ geometric,
    stdv ~ gamma.  Returns a structured array with the reference's event_t layout."""
    rng = np.random.RandomState(seed)
    ev = np.zeros(n, dtype=EVENT_DTYPE)
    length = rng.geometric(1.0 / mean_dwell, size=n).astype(np.float32)
    ev["length"] = length
    ev["start"] = np.concatenate([[0], np.cumsum(length[:-1])]).astype(np.uint64)
    ev["mean"] = (90.0 + 12.0 * rng.standard_normal(n)).astype(np.float32)
    ev["stdv"] = rng.gamma(4.0, 0.4, size=n).astype(np.float32)
    ev["pos"] = -1
    ev["state"] = -1
    return ev
