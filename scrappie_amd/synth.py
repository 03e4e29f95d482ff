"""Seeded synthetic inputs for tests and bench (there is no network for real
fast5 data, and the bundled reads need HDF5).

* `synthetic_signal`   -- squiggle-like raw signal:
  piecewise-constant levels ~N(0,1), dwell ~Geometric(mean 9 samples), plus
  N(0, 0.1^2) noise.
* `hmm_output_layer` / `hmm_trunk` -- an output layer built from +-1 state codes and
  trunk activations that encode a simulated k-mer path, so that the PRODUCTION S1 +
  decoder kernels (scrappie_hip_set_trunk_input) see HMM-like posteriors and decode
  ~0.5 bases per block instead of the handful random weights give.
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


def fixture_posterior(T, seed, klen=5, hp=0):
    """The input of a decode fixture case (tests/golden/ref_decode.npz): hp >= 0: simulated_posterior with that
    many planted homopolymers; hp < 0: a FLAT posterior, Dirichlet(1) over all states -- what a random-weight
    network gives: with the default penalties it decodes to the all-start path, with a large local penalty the
    best path must run through the k-mer states on near-ties."""
    if hp >= 0:
        return simulated_posterior(T, seed, klen=klen, plant_homopolymers=hp)[0]
    rng = np.random.RandomState(seed)
    post = rng.dirichlet(np.ones(4 ** klen + 1), size=T).astype(np.float32)
    return np.log(np.float32(1e-5) + np.float32(1.0 - 1e-5) * post).astype(np.float32)


HMM_STAY_UNIT = 20          # units 0..19: (position, base) codes of the current k-mer; 20: stay indicator


def hmm_output_layer(S=96, klen=5, seed=0, g=0.8, g_stay=8.0, b_stay=8.0, jitter=0.1):
    """(ff_W (4**klen + 1, S), ff_b) of an output layer that reads a k-mer off +-1 state codes:
    row s of a k-mer state has +g on unit 4p + base_p(s), -g on the other three units of position p (p = 0 is
    the oldest base, the most significant digit of s: decode.c:129-131) and -g_stay on the stay unit; the stay
    row (LAST, misc/parse_rgrgr.py:127-130) has +g_stay there and the bias b_stay.  With activations
    a (2 onehot - 1) per position the logit of a state h mismatches away from the encoded k-mer is
    4 g a (5 - h) -+ g_stay sigma: a softmax over Hamming neighbourhoods, as a trained transducer gives.
    The remaining units carry small random weights (jitter)."""
    assert S >= 4 * klen + 1
    rng = np.random.RandomState(7000 + seed)
    nk = 4 ** klen
    W = rng.uniform(-jitter, jitter, size=(nk + 1, S)).astype(np.float32)
    W[:, :4 * klen + 1] = 0.0
    s = np.arange(nk)
    for p in range(klen):
        base = (s >> (2 * (klen - 1 - p))) & 3
        for c in range(4):
            W[:nk, 4 * p + c] = np.where(base == c, g, -g)
    W[:nk, 4 * klen] = -g_stay
    W[nk, 4 * klen] = g_stay
    b = rng.uniform(-0.05, 0.05, size=nk + 1).astype(np.float32)
    b[nk] += b_stay
    return W, b


def hmm_trunk(T, seed, S=96, klen=5, p_stay=0.55, p_skip=0.05, plant_homopolymers=0):
    """Trunk activations (T, S) in (-1, 1) encoding a simulated k-mer path for `hmm_output_layer`
    (about 55 % stay / 40 % step / 5 % skip; confidence a_t ~ U(0.5, 1), stay indicator of either sign with
    |sigma| ~ U(0.05, 1); planted homopolymer runs with an ambiguous stay indicator exercise
    homopolymer.c:175).  Returns (trunk float32, true_path[T] with -1 = stay)."""
    rng = np.random.RandomState(seed)
    nk = 4 ** klen
    x = rng.uniform(-0.5, 0.5, size=(T, S)).astype(np.float32)
    path = np.full(T, -1, dtype=np.int64)
    kmer = int(rng.randint(nk))
    homo_left = plant_homopolymers
    t = 0
    cur = np.zeros(T, dtype=np.int64)
    sigma = np.zeros(T, dtype=np.float32)
    while t < T:
        if homo_left > 0 and t > 10 and rng.rand() < 0.02 and t + 12 < T:
            b = int(rng.randint(4))
            hk = sum(b * 4 ** i for i in range(klen))
            xb = (b + 1 + int(rng.randint(3))) % 4
            kmer = (xb * 4 ** (klen - 1)) + (hk % 4 ** (klen - 1))
            path[t] = kmer; cur[t] = kmer; sigma[t] = -rng.uniform(0.3, 1.0)
            runlen = int(rng.randint(3, 9))
            for j in range(1, min(runlen, T - 1 - t) + 1):
                cur[t + j] = hk
                sigma[t + j] = rng.uniform(-0.12, 0.12)       # stay or another base of the run: ambiguous
                path[t + j] = -2
            kmer = hk
            t += runlen + 1
            homo_left -= 1
            continue
        r = rng.rand()
        if r < p_stay:
            sigma[t] = rng.uniform(0.05, 1.0)
        else:
            n = 2 if r < p_stay + p_skip else 1
            kmer = ((kmer * 4 ** n) % nk) + int(rng.randint(4 ** n))
            path[t] = kmer
            sigma[t] = -rng.uniform(0.05, 1.0)
        cur[t] = kmer
        t += 1
    a = rng.uniform(0.5, 1.0, size=T).astype(np.float32)
    for p in range(klen):
        base = (cur >> (2 * (klen - 1 - p))) & 3
        for c in range(4):
            x[:, 4 * p + c] = np.where(base == c, a, -a)
    x[:, 4 * klen] = sigma
    return x.astype(np.float32), path


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
