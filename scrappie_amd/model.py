"""Model weight sets for the raw basecalling path.

The reference compiles its weights in from generated headers
(src/models/*.h, format: misc/parse_rgrgr.py:28-55,77-130,
misc/parse_rnnrf.py:75-127).  Those headers are missing blobs in the reference
checkout, so this module provides

  * `synthetic_model`  -- seeded weights with the reference's matrix shapes,
  * `save_model` / `load_model` -- the `.scrm` container the C engine loads,
  * `model_from_header` -- ingest of a real scrappie model header, should a user
    have one (symbol names as networks.c:259-288 / :577-611 use them).

A model is a dict:
  arch      'rgrgr' (softmax/transducer) | 'rnnrf' (residual GRUs + globalnorm)
  conv_act  'elu' | 'tanh'          (networks.c:260 vs :358)
  stride    conv stride             (conv_<tag>_stride)
  conv_W    (F, WL)   conv_b (F,)
  gru{l}_iW (3S, I)   gru{l}_sW (2S, S)   gru{l}_sW2 (S, S)   gru{l}_b (3S,)
            l = 0..4 for B1 F2 B3 F4 B5; row = one output unit's weights,
            i.e. one *column* of the reference's `_Mat`
  ff_W      (NS, S)   ff_b (NS,)    (transducer: stay state is the LAST row,
                                     misc/parse_rgrgr.py:127-130)
"""
import re
import struct

import numpy as np

MAGIC = b"SCRMDL01"
ARCH_ID = {"rgrgr": 0, "rnnrf": 1, "raw": 2, "events": 3}
ACT_ID = {"elu": 0, "tanh": 1}

# name -> (arch, conv_act, winlen, nstate): shapes per SURVEY.md section 8
# (NS=1025 and stride=5 are pinned by python/test/test_scrappy.py:46-48;
#  S=F=96 and the window lengths are assumed: the real headers are missing)
MODEL_SHAPES = {
    "raw_r94": ("raw", "tanh", 11, 1025),          # bi-GRU: networks.c:196-247 (size assumed)
    "rgrgr_r94": ("rgrgr", "elu", 11, 1025),
    "rgrgr_r941": ("rgrgr", "elu", 11, 1025),
    "rgrgr_r10": ("rgrgr", "tanh", 19, 1025),
    "rnnrf_r94": ("rnnrf", "elu", 11, 25),
    # events bi-LSTM (networks.c:146-193): no convolution, 12 input features per event
    # (3-event window x 4 features); LSTM size assumed (nanonet_events.h is not in the checkout)
    "nanonet_events": ("events", "tanh", 0, 1025),
}

MATRIX_NAMES = (["conv_W", "conv_b"]
                + ["gru%d_%s" % (l, n) for l in range(5) for n in ("iW", "sW", "sW2", "b")]
                + ["ff_W", "ff_b"])
# raw_r94 (bi-GRU): gru0..3 = F1, B1, F2, B2 (misc/parse_raw.py:66-128); ff1/ff2 combine the two
# directions (feedforward2_tanh), ff_W/ff_b is the softmax layer FF3
RAW_MATRIX_NAMES = (["conv_W", "conv_b"]
                    + ["gru%d_%s" % (l, n) for l in range(4) for n in ("iW", "sW", "sW2", "b")]
                    + ["ff1_Wf", "ff1_Wb", "ff1_b", "ff2_Wf", "ff2_Wb", "ff2_b", "ff_W", "ff_b"])


# events: lstm0..3 = F1, B1, F2, B2 (networks.c:163-180): iW (4S, I), sW (4S, S), b (4S,), p (3S,)
# with the gate order [input | update | forget | output] and peepholes [update | forget | output]
# (layers.c:806-830)
EVENTS_MATRIX_NAMES = (["lstm%d_%s" % (l, n) for l in range(4) for n in ("iW", "sW", "b", "p")]
                       + ["ff1_Wf", "ff1_Wb", "ff1_b", "ff2_Wf", "ff2_Wb", "ff2_b", "ff_W", "ff_b"])
EVENT_FEATURES = 12


def matrix_names(m):
    if m["arch"] == "events":
        return EVENTS_MATRIX_NAMES
    return RAW_MATRIX_NAMES if m["arch"] == "raw" else MATRIX_NAMES


def synthetic_model(name="rgrgr_r94", seed=1, size=96, nfilter=None, winlen=None,
                    stride=5, nstate=None, ff_scale=6.0):
    """Seeded synthetic weights, reference shapes.

    conv/iW/sW/sW2 ~ U(+-sqrt(3/fan_in)), biases ~ U(+-0.1), FF W scaled so
    posteriors are peaked.
    """
    arch, act, wl, ns = MODEL_SHAPES[name]
    winlen = winlen or wl
    nstate = nstate or ns
    S = size
    F = nfilter or S
    rng = np.random.RandomState(seed)

    def u(shape, fan_in):
        a = np.sqrt(3.0 / fan_in)
        return rng.uniform(-a, a, size=shape).astype(np.float32)

    def b(n):
        return rng.uniform(-0.1, 0.1, size=n).astype(np.float32)

    if arch == "events":
        m = {"name": name, "arch": arch, "conv_act": act, "stride": 1}
        for l in range(4):
            I = EVENT_FEATURES if l < 2 else S
            m["lstm%d_iW" % l] = u((4 * S, I), I)
            m["lstm%d_sW" % l] = u((4 * S, S), S)
            m["lstm%d_b" % l] = b(4 * S)
            m["lstm%d_p" % l] = rng.uniform(-0.5, 0.5, size=3 * S).astype(np.float32)
        for k in ("ff1", "ff2"):
            m[k + "_Wf"] = u((S, S), 2 * S)
            m[k + "_Wb"] = u((S, S), 2 * S)
            m[k + "_b"] = b(S)
        m["ff_W"] = (u((nstate, S), S) * ff_scale).astype(np.float32)
        m["ff_b"] = b(nstate)
        return m
    m = {"name": name, "arch": arch, "conv_act": act, "stride": int(stride)}
    m["conv_W"] = u((F, winlen), winlen)
    m["conv_b"] = b(F)
    if arch == "raw":
        for l in range(4):
            I = F if l < 2 else S
            m["gru%d_iW" % l] = u((3 * S, I), I)
            m["gru%d_sW" % l] = u((2 * S, S), S)
            m["gru%d_sW2" % l] = u((S, S), S)
            m["gru%d_b" % l] = b(3 * S)
        for k in ("ff1", "ff2"):
            m[k + "_Wf"] = u((S, S), 2 * S)
            m[k + "_Wb"] = u((S, S), 2 * S)
            m[k + "_b"] = b(S)
        m["ff_W"] = (u((nstate, S), S) * ff_scale).astype(np.float32)
        m["ff_b"] = b(nstate)
        return m
    for l in range(5):
        I = F if l == 0 else S
        m["gru%d_iW" % l] = u((3 * S, I), I)
        m["gru%d_sW" % l] = u((2 * S, S), S)
        m["gru%d_sW2" % l] = u((S, S), S)
        m["gru%d_b" % l] = b(3 * S)
    scale = ff_scale if arch == "rgrgr" else 1.0
    m["ff_W"] = (u((nstate, S), S) * scale).astype(np.float32)
    m["ff_b"] = b(nstate)
    return m


def save_model(m, path):
    """Write the `.scrm` container (read by scrappie_amd/csrc/sh_model.c)."""
    with open(path, "wb") as fh:
        fh.write(MAGIC)
        fh.write(struct.pack("<IIII", ARCH_ID[m["arch"]], ACT_ID[m["conv_act"]],
                             int(m["stride"]), len(matrix_names(m))))
        for nm in matrix_names(m):
            a = np.ascontiguousarray(m[nm], dtype=np.float32)
            if a.ndim == 1:
                a = a.reshape(1, -1)
            fh.write(struct.pack("<32sII", nm.encode(), a.shape[1], a.shape[0]))  # nr(in), nc(out)
            fh.write(a.tobytes())


def load_model(path):
    with open(path, "rb") as fh:
        if fh.read(8) != MAGIC:
            raise ValueError("%s: not a .scrm model container" % path)
        arch, act, stride, nmat = struct.unpack("<IIII", fh.read(16))
        m = {"arch": {v: k for k, v in ARCH_ID.items()}[arch],
             "conv_act": {v: k for k, v in ACT_ID.items()}[act], "stride": stride}
        for _ in range(nmat):
            nm, nr, nc = struct.unpack("<32sII", fh.read(40))
            nm = nm.rstrip(b"\0").decode()
            a = np.frombuffer(fh.read(4 * nr * nc), dtype=np.float32).reshape(nc, nr).copy()
            m[nm] = a[0] if nm.endswith(("_b", "_p")) else a
    return m


_ARRAY_RE = re.compile(r"float\s+__(\w+)\[\]\s*=\s*\{(.*?)\};", re.S)
_MAT_RE = re.compile(r"_Mat\s+_(\w+)\s*=\s*\{\s*\.nr\s*=\s*(\d+),\s*\.nrq\s*=\s*(\d+),"
                     r"\s*\.nc\s*=\s*(\d+),\s*\.stride\s*=\s*(\d+)", re.S)
_STRIDE_RE = re.compile(r"const\s+int\s+conv_\w*stride\s*=\s*(\d+)")


def model_from_header(path, arch=None, conv_act="elu"):
    """Parse a scrappie model header (grammar: misc/parse_rgrgr.py:28-55) into a
    model dict.  Symbol names: conv_<tag>_W/_b, gru{B1,F2,B3,F4,B5}_<tag>_{iW,sW,sW2,b},
    FF_<tag>_W/_b (networks.c:259-288)."""
    text = open(path).read()
    arrays = {k: np.array([float.fromhex(t.strip()) for t in v.replace("\n", " ").split(",") if t.strip()],
                          dtype=np.float32)
              for k, v in _ARRAY_RE.findall(text)}
    mats = {}
    for nm, nr, nrq, nc, stride in _MAT_RE.findall(text):
        nr, nc, stride = int(nr), int(nc), int(stride)
        mats[nm] = arrays[nm].reshape(nc, stride)[:, :nr]
    if arch is None:
        arch = "rnnrf" if any("rnnrf" in k for k in mats) else "rgrgr"
    m = {"arch": arch, "conv_act": conv_act, "stride": int(_STRIDE_RE.search(text).group(1))}

    def find(prefix, suffix):
        ks = [k for k in mats if k.startswith(prefix) and k.endswith(suffix)]
        if len(ks) != 1:
            raise KeyError("header has no unique matrix %s*%s" % (prefix, suffix))
        return mats[ks[0]]

    cw = find("conv_", "_W")            # (F, 4*WL-3), tap w at row 4w
    m["conv_W"] = np.ascontiguousarray(cw[:, 0::4])
    m["conv_b"] = find("conv_", "_b")[0].copy()
    for l, tag in enumerate(("gruB1_", "gruF2_", "gruB3_", "gruF4_", "gruB5_")):
        m["gru%d_iW" % l] = find(tag, "_iW").copy()
        m["gru%d_sW" % l] = find(tag, "_sW").copy()
        m["gru%d_sW2" % l] = find(tag, "_sW2").copy()
        m["gru%d_b" % l] = find(tag, "_b")[0].copy()
    m["ff_W"] = find("FF_", "_W").copy()
    m["ff_b"] = find("FF_", "_b")[0].copy()
    return m


def model_dims(m):
    if m["arch"] == "events":
        return dict(F=EVENT_FEATURES, WL=0, S=m["lstm0_sW"].shape[1], NS=m["ff_W"].shape[0], stride=1)
    F, WL = m["conv_W"].shape
    S = m["gru0_sW2"].shape[0]
    NS = m["ff_W"].shape[0]
    return dict(F=F, WL=WL, S=S, NS=NS, stride=int(m["stride"]))


def flops_per_block(m):
    """Algorithmic FLOPs per output block:
    2*[F*WL + sum_l(I_l*3S + 2S^2 + S^2) + S*NS]."""
    d = model_dims(m)
    F, WL, S, NS = d["F"], d["WL"], d["S"], d["NS"]
    if m["arch"] == "events":     # per event: 4 x (I*4S + 4S*S) + 2 x 2S*S + S*NS
        tot = S * NS + 2 * 2 * S * S
        for l in range(4):
            tot += (F if l < 2 else S) * 4 * S + 4 * S * S
        return 2 * tot
    tot = F * WL + S * NS
    if m["arch"] == "raw":
        for l in range(4):
            I = F if l < 2 else S
            tot += I * 3 * S + 3 * S * S
        return 2 * (tot + 2 * 2 * S * S)
    for l in range(5):
        I = F if l == 0 else S
        tot += I * 3 * S + 3 * S * S
    return 2 * tot
