"""ctypes front-end to the CPU oracle (oracle/oracle.c) and, when built, the
compiled-reference checkers (oracle/_ref/*.so).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg -- never by anything under scrappie_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class Mat(C.Structure):
    """orc_mat == reference _Mat (src/scrappie_matrix.h:10-16)."""
    _fields_ = [("nr", C.c_size_t), ("nrq", C.c_size_t), ("nc", C.c_size_t),
                ("stride", C.c_size_t), ("data", C.c_void_p)]


class RawTable(C.Structure):
    """orc_raw_table == reference raw_table (src/scrappie_structures.h:24-30)."""
    _fields_ = [("uuid", C.c_char_p), ("n", C.c_size_t), ("start", C.c_size_t),
                ("end", C.c_size_t), ("raw", C.POINTER(C.c_float))]


class Model(C.Structure):
    _fields_ = [("arch", C.c_int), ("conv_act", C.c_int), ("stride", C.c_int),
                ("conv_W", C.POINTER(Mat)), ("conv_b", C.POINTER(Mat)),
                ("gru_iW", C.POINTER(Mat) * 5), ("gru_sW", C.POINTER(Mat) * 5),
                ("gru_sW2", C.POINTER(Mat) * 5), ("gru_b", C.POINTER(Mat) * 5),
                ("ff_W", C.POINTER(Mat)), ("ff_b", C.POINTER(Mat)),
                ("ff1_Wf", C.POINTER(Mat)), ("ff1_Wb", C.POINTER(Mat)), ("ff1_b", C.POINTER(Mat)),
                ("ff2_Wf", C.POINTER(Mat)), ("ff2_Wb", C.POINTER(Mat)), ("ff2_b", C.POINTER(Mat)),
                ("lstm_p", C.POINTER(Mat) * 4)]


class Event(C.Structure):           # scrappie_structures.h:8-15
    _fields_ = [("start", C.c_uint64), ("length", C.c_float), ("mean", C.c_float), ("stdv", C.c_float),
                ("pos", C.c_int), ("state", C.c_int)]


class EventTable(C.Structure):      # scrappie_structures.h:17-22
    _fields_ = [("n", C.c_size_t), ("start", C.c_size_t), ("end", C.c_size_t), ("event", C.POINTER(Event))]


class Call(C.Structure):
    _fields_ = [("score", C.c_float), ("nblock", C.c_size_t), ("start", C.c_size_t),
                ("end", C.c_size_t), ("basecall", C.c_void_p), ("pos", C.POINTER(C.c_int))]


class Params(C.Structure):
    _fields_ = [("min_prob", C.c_float), ("tempW", C.c_float), ("tempb", C.c_float),
                ("stay_pen", C.c_float), ("skip_pen", C.c_float), ("local_pen", C.c_float),
                ("use_slip", C.c_int), ("homopolymer_mean", C.c_int),
                ("trim_start", C.c_int), ("trim_end", C.c_int), ("varseg_chunk", C.c_int),
                ("varseg_thresh", C.c_float), ("do_trim", C.c_int)]


def build(fast=False, quiet=True):
    """(Re)build liboracle.so (+ _ref when /root/reference exists)."""
    targets = ["liboracle.so", "ref"] + (["liboracle_fast.so"] if fast else [])
    subprocess.run(["make", "-C", _HERE] + targets, check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def _load(name):
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        return None
    return C.CDLL(path, mode=C.RTLD_LOCAL)


_lib = None
_ref_pure = None
_ref_decode = None
_ref_features = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(os.path.join(_HERE, "liboracle.so")):
            build()
        _lib = _load("liboracle.so")
        _declare(_lib)
    return _lib


def fast_lib():
    if not os.path.exists(os.path.join(_HERE, "liboracle_fast.so")):
        build(fast=True)
    l = _load("liboracle_fast.so")
    _declare(l)
    return l


def ref_pure():
    """Reference util.c/scrappie_common.c/homopolymer.c/seq_helpers compiled as
    shipped + header-inline math exports; None when not built."""
    global _ref_pure
    if _ref_pure is None:
        _ref_pure = _load("_ref/libref_pure.so")
    return _ref_pure


def ref_decode():
    """Reference decode.c compiled as shipped, hosted on the oracle allocator;
    None when not built."""
    global _ref_decode
    if _ref_decode is None:
        _ref_decode = _load("_ref/libref_decode.so")
    return _ref_decode


def ref_features():
    """Reference nnfeatures.c (event features) compiled as shipped, hosted on the oracle
    allocator; None when not built."""
    global _ref_features
    if _ref_features is None:
        _ref_features = _load("_ref/libref_features.so")
    return _ref_features


def _declare(l):
    PM = C.POINTER(Mat)
    fp = C.POINTER(C.c_float)
    ip = C.POINTER(C.c_int)
    l.orc_make_mat.restype = PM
    l.orc_make_mat.argtypes = [C.c_size_t, C.c_size_t]
    l.orc_free_mat.restype = PM
    l.orc_free_mat.argtypes = [PM]
    for f in ("orc_expf", "orc_logf", "orc_logisticf", "orc_tanhf", "orc_eluf"):
        getattr(l, f).restype = C.c_float
        getattr(l, f).argtypes = [C.c_float]
    l.orc_logsumexpf.restype = C.c_float
    l.orc_logsumexpf.argtypes = [C.c_float, C.c_float]
    l.orc_medianf.restype = C.c_float
    l.orc_medianf.argtypes = [fp, C.c_size_t]
    l.orc_madf.restype = C.c_float
    l.orc_madf.argtypes = [fp, C.c_size_t, fp]
    l.orc_medmad_normalise_array.argtypes = [fp, C.c_size_t]
    l.orc_trim_raw_by_mad.restype = RawTable
    l.orc_trim_raw_by_mad.argtypes = [RawTable, C.c_size_t, C.c_float]
    l.orc_trim_and_segment_raw.restype = RawTable
    l.orc_trim_and_segment_raw.argtypes = [RawTable, C.c_size_t, C.c_size_t, C.c_size_t, C.c_float]
    l.orc_convolution.restype = PM
    l.orc_convolution.argtypes = [PM, PM, PM, C.c_size_t, PM]
    for f in ("orc_tanh_activation_inplace", "orc_exp_activation_inplace",
              "orc_elu_activation_inplace", "orc_row_normalise_inplace"):
        getattr(l, f).argtypes = [PM]
    l.orc_robustlog_activation_inplace.argtypes = [PM, C.c_float]
    l.orc_affine_map.restype = PM
    l.orc_affine_map.argtypes = [PM, PM, PM, PM]
    l.orc_gru_forward.restype = PM
    l.orc_gru_forward.argtypes = [PM, PM, PM, PM]
    l.orc_gru_backward.restype = PM
    l.orc_gru_backward.argtypes = [PM, PM, PM, PM]
    l.orc_softmax_with_temperature.restype = PM
    l.orc_softmax_with_temperature.argtypes = [PM, PM, PM, C.c_float, C.c_float, PM]
    l.orc_globalnorm.restype = PM
    l.orc_globalnorm.argtypes = [PM, PM, PM, PM]
    l.orc_crf_partition_function.restype = C.c_float
    l.orc_crf_partition_function.argtypes = [PM]
    l.orc_posterior.restype = PM
    l.orc_posterior.argtypes = [C.POINTER(Model), RawTable, C.c_float, C.c_float, C.c_float, C.c_bool]
    l.orc_trunk.restype = PM
    l.orc_trunk.argtypes = [C.POINTER(Model), RawTable, C.c_int]
    l.orc_decode_transducer.restype = C.c_float
    l.orc_decode_transducer.argtypes = [PM, C.c_float, C.c_float, C.c_float, ip, C.c_bool]
    l.orc_sloika_viterbi.restype = C.c_float
    l.orc_sloika_viterbi.argtypes = [PM, C.c_float, C.c_float, C.c_float, ip]
    l.orc_overlapper.restype = C.c_void_p
    l.orc_overlapper.argtypes = [ip, C.c_size_t, C.c_int, ip]
    l.orc_decode_crf.restype = C.c_float
    l.orc_decode_crf.argtypes = [PM, ip]
    l.orc_crfpath_to_basecall.restype = C.c_void_p
    l.orc_crfpath_to_basecall.argtypes = [ip, C.c_size_t, ip]
    l.orc_posterior_crf.restype = PM
    l.orc_posterior_crf.argtypes = [PM]
    l.orc_homopolymer_path.restype = C.c_int
    l.orc_homopolymer_path.argtypes = [PM, ip, C.c_int]
    if hasattr(l, "orc_lstm_forward"):
        l.orc_features_from_events.restype = PM
        l.orc_features_from_events.argtypes = [EventTable, C.c_bool]
        l.orc_window.restype = PM
        l.orc_window.argtypes = [PM, C.c_size_t, C.c_size_t]
        l.orc_lstm_forward.restype = PM
        l.orc_lstm_forward.argtypes = [PM, PM, PM, PM]
        l.orc_lstm_backward.restype = PM
        l.orc_lstm_backward.argtypes = [PM, PM, PM, PM]
        l.orc_events_trunk.restype = PM
        l.orc_events_trunk.argtypes = [C.POINTER(Model), PM, C.c_int]
        l.orc_events_posterior_from_features.restype = PM
        l.orc_events_posterior_from_features.argtypes = [C.POINTER(Model), PM, C.c_float, C.c_float, C.c_float, C.c_bool]
        l.orc_events_posterior.restype = PM
        l.orc_events_posterior.argtypes = [C.POINTER(Model), EventTable, C.c_float, C.c_float, C.c_float, C.c_bool]
    l.orc_default_params.restype = Params
    l.orc_basecall_raw.restype = C.c_int
    l.orc_basecall_raw.argtypes = [C.POINTER(Model), fp, C.c_size_t, C.POINTER(Params), C.POINTER(Call)]


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def take_string(ptr):
    """Copy and free a malloc'd C string returned by the oracle/reference."""
    if not ptr:
        return None
    s = C.string_at(ptr).decode()
    _libc.free(ptr)
    return s


# ----------------------------------------------------------------------------
# numpy <-> padded column-major _Mat
# ----------------------------------------------------------------------------
class NpMat:
    """A _Mat whose storage is a numpy buffer (kept alive by this object).

    `arr` has shape (nc, nr): row c of the numpy array is column c of the
    scrappie matrix (python/scrappy/__init__.py:259-265 reads it the same way).
    """

    def __init__(self, arr, nr=None):
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        if arr.ndim == 1:
            arr = arr.reshape(1, -1)
        nc, nr_a = arr.shape
        nr = nr_a if nr is None else nr
        nrq = (nr + 3) // 4
        self.buf = np.zeros((nc, 4 * nrq), dtype=np.float32)
        self.buf[:, :nr_a] = arr
        self.mat = Mat(nr, nrq, nc, 4 * nrq, self.buf.ctypes.data)

    @property
    def ptr(self):
        p = C.pointer(self.mat)
        p._owner = self          # keep the numpy buffer alive as long as the pointer is
        return p


def mat_to_numpy(pm, free_with=None, padded=False):
    """Copy a returned `_Mat*` into an (nc, nr) float32 array (optionally with
    the pad lanes) and free it with `free_with` (e.g. lib().orc_free_mat)."""
    if not pm:
        return None
    m = pm.contents
    n = m.nc * m.stride
    flat = np.ctypeslib.as_array(C.cast(m.data, C.POINTER(C.c_float)), shape=(n,)).copy()
    out = flat.reshape(m.nc, m.stride)
    if not padded:
        out = np.ascontiguousarray(out[:, :m.nr])
    if free_with is not None:
        free_with(pm)
    return out


def conv_filter_mat(conv_W):
    """(F, WL) taps -> the reference's padded filter layout: nr = 4*WL-3,
    tap w at row 4w (misc/parse_rgrgr.py:77-81)."""
    F, WL = conv_W.shape
    spread = np.zeros((F, 4 * WL), dtype=np.float32)
    spread[:, 0::4] = conv_W
    m = NpMat(spread[:, :4 * WL - 3])
    assert m.mat.stride == 4 * WL and m.mat.nrq == WL
    return m


class OracleModel:
    """Holds numpy-backed _Mat objects + the orc_model struct.

    `w` is a dict (see scrappie_amd.model): conv_W (F,WL), conv_b (F,),
    gru{0..4}_iW (3S,I), gru{l}_sW (2S,S), gru{l}_sW2 (S,S), gru{l}_b (3S,),
    ff_W (NS,S), ff_b (NS,); meta: arch ('rgrgr'|'rnnrf'), conv_act ('elu'|'tanh'),
    stride.
    """

    def __init__(self, w):
        self.keep = {}
        k = self.keep
        if w["arch"] == "events":
            m = Model()
            m.arch, m.conv_act, m.stride = 3, 1, 1
            for l in range(4):
                for nm, slot in (("iW", m.gru_iW), ("sW", m.gru_sW)):
                    k["lstm%d_%s" % (l, nm)] = NpMat(w["lstm%d_%s" % (l, nm)])
                    slot[l] = k["lstm%d_%s" % (l, nm)].ptr
                k["lstm%d_b" % l] = NpMat(w["lstm%d_b" % l].reshape(1, -1))
                m.gru_b[l] = k["lstm%d_b" % l].ptr
                k["lstm%d_p" % l] = NpMat(w["lstm%d_p" % l].reshape(1, -1))
                m.lstm_p[l] = k["lstm%d_p" % l].ptr
            for nm in ("ff1_Wf", "ff1_Wb", "ff2_Wf", "ff2_Wb", "ff_W"):
                k[nm] = NpMat(w[nm])
                setattr(m, nm, k[nm].ptr)
            for nm in ("ff1_b", "ff2_b", "ff_b"):
                k[nm] = NpMat(w[nm].reshape(1, -1))
                setattr(m, nm, k[nm].ptr)
            self.struct = m
            self.w = w
            return
        k["conv_W"] = conv_filter_mat(w["conv_W"])
        k["conv_b"] = NpMat(w["conv_b"].reshape(1, -1))
        ngru = 4 if w["arch"] == "raw" else 5
        for l in range(ngru):
            for nm in ("iW", "sW", "sW2"):
                k["gru%d_%s" % (l, nm)] = NpMat(w["gru%d_%s" % (l, nm)])
            k["gru%d_b" % l] = NpMat(w["gru%d_b" % l].reshape(1, -1))
        k["ff_W"] = NpMat(w["ff_W"])
        k["ff_b"] = NpMat(w["ff_b"].reshape(1, -1))
        m = Model()
        m.arch = {"rgrgr": 0, "rnnrf": 1, "raw": 2}[w["arch"]]
        m.conv_act = 1 if w["conv_act"] == "tanh" else 0
        m.stride = int(w["stride"])
        m.conv_W, m.conv_b = k["conv_W"].ptr, k["conv_b"].ptr
        if w["arch"] == "raw":
            for nm in ("ff1_Wf", "ff1_Wb", "ff2_Wf", "ff2_Wb"):
                k[nm] = NpMat(w[nm])
                setattr(m, nm, k[nm].ptr)
            for nm in ("ff1_b", "ff2_b"):
                k[nm] = NpMat(w[nm].reshape(1, -1))
                setattr(m, nm, k[nm].ptr)
        for l in range(ngru):
            m.gru_iW[l] = k["gru%d_iW" % l].ptr
            m.gru_sW[l] = k["gru%d_sW" % l].ptr
            m.gru_sW2[l] = k["gru%d_sW2" % l].ptr
            m.gru_b[l] = k["gru%d_b" % l].ptr
        m.ff_W, m.ff_b = k["ff_W"].ptr, k["ff_b"].ptr
        self.struct = m
        self.w = w

    @property
    def ptr(self):
        return C.pointer(self.struct)


def raw_table(signal, start=0, end=None):
    signal = np.ascontiguousarray(signal, dtype=np.float32)
    rt = RawTable(None, len(signal), start, len(signal) if end is None else end,
                  signal.ctypes.data_as(C.POINTER(C.c_float)))
    return rt, signal   # keep `signal` alive


# ----------------------------------------------------------------------------
# convenience wrappers used by tests / bench
# ----------------------------------------------------------------------------
def posterior(model, signal, min_prob=1e-5, tempW=1.0, tempb=1.0, log=True, L=None):
    L = L or lib()
    rt, keep = raw_table(signal)
    pm = L.orc_posterior(model.ptr, rt, min_prob, tempW, tempb, log)
    return mat_to_numpy(pm, L.orc_free_mat)


def softmax_posterior(model, top, min_prob=1e-5, tempW=1.0, tempb=1.0, log=True):
    """S1 (+ S2) on a given trunk output `top` (T, S): softmax_with_temperature (layers.c:340) and, if `log`,
    robustlog (layers.c:79), as networks.c:290-294 applies them."""
    L = lib()
    x = NpMat(np.array(top, dtype=np.float32, copy=True))      # scaled in place by tempW / tempb (Q5)
    k = model.keep
    pm = L.orc_softmax_with_temperature(x.ptr, k["ff_W"].ptr, k["ff_b"].ptr, tempW, tempb, None)
    if pm and log:
        L.orc_robustlog_activation_inplace(pm, min_prob)
    return mat_to_numpy(pm, L.orc_free_mat)


def trunk(model, signal, upto):
    L = lib()
    rt, keep = raw_table(signal)
    return mat_to_numpy(L.orc_trunk(model.ptr, rt, upto), L.orc_free_mat)


def decode_transducer(post, stay_pen=0.0, skip_pen=0.0, local_pen=2.0, slip=False, fn=None):
    """post: (T, NS) log-posterior.  Returns (score, seq[T+1])."""
    m = NpMat(post)
    seq = np.zeros(post.shape[0] + 1, dtype=np.int32)
    fn = fn or lib().orc_decode_transducer
    score = fn(m.ptr, stay_pen, skip_pen, local_pen, seq.ctypes.data_as(C.POINTER(C.c_int)), slip)
    return float(score), seq


def overlapper(seq, nkmer, fn=None):
    seq = np.ascontiguousarray(seq, dtype=np.int32)
    pos = np.zeros(len(seq), dtype=np.int32)
    fn = fn or lib().orc_overlapper
    p = fn(seq.ctypes.data_as(C.POINTER(C.c_int)), len(seq), nkmer,
           pos.ctypes.data_as(C.POINTER(C.c_int)))
    return take_string(p), pos


def decode_crf(trans, fn=None):
    m = NpMat(trans)
    path = np.zeros(trans.shape[0] + 1, dtype=np.int32)
    fn = fn or lib().orc_decode_crf
    score = fn(m.ptr, path.ctypes.data_as(C.POINTER(C.c_int)))
    return float(score), path


def crfpath_to_basecall(path, npos, fn=None):
    path = np.ascontiguousarray(path, dtype=np.int32)
    pos = np.zeros(len(path), dtype=np.int32)
    fn = fn or lib().orc_crfpath_to_basecall
    p = fn(path.ctypes.data_as(C.POINTER(C.c_int)), npos, pos.ctypes.data_as(C.POINTER(C.c_int)))
    return take_string(p)


def homopolymer_path(post, path, fn=None, mean_flag=1):
    m = NpMat(post)
    path = np.ascontiguousarray(path, dtype=np.int32).copy()
    fn = fn or lib().orc_homopolymer_path
    rc = fn(m.ptr, path.ctypes.data_as(C.POINTER(C.c_int)), mean_flag)
    return rc, path


def basecall_raw(model, raw, params=None, L=None):
    """Full per-read path as scrappie_raw.c:265-315.  Returns dict or None."""
    L = L or lib()
    p = params or L.orc_default_params()
    raw = np.ascontiguousarray(raw, dtype=np.float32)
    out = Call()
    rc = L.orc_basecall_raw(model.ptr, raw.ctypes.data_as(C.POINTER(C.c_float)), len(raw),
                            C.byref(p), C.byref(out))
    if rc != 0:
        return None
    pos = np.ctypeslib.as_array(out.pos, shape=(out.nblock + 1,)).copy()
    _libc.free(C.cast(out.pos, C.c_void_p))
    return dict(bases=take_string(out.basecall), score=float(out.score), nblock=int(out.nblock),
                start=int(out.start), end=int(out.end), pos=pos)


# ----------------------------------------------------------------------------
# events path
# ----------------------------------------------------------------------------
def event_table(ev, start=0, end=None):
    """`ev`: structured array with scrappie_amd.synth.EVENT_DTYPE.  Returns (EventTable, keepalive)."""
    ev = np.ascontiguousarray(ev)
    assert ev.dtype.itemsize == C.sizeof(Event)
    et = EventTable(len(ev), start, len(ev) if end is None else end, C.cast(ev.ctypes.data, C.POINTER(Event)))
    return et, ev


def features_from_events(ev, normalise=True, fn=None, free=None):
    """(nevent, 4) features (nnfeatures.c:88); `fn`/`free` select the compiled reference."""
    et, keep = event_table(ev)
    L = lib()
    return mat_to_numpy((fn or L.orc_features_from_events)(et, normalise), free or L.orc_free_mat)


def window(feat, w=3, stride=1):
    """(nc, nr) -> (nc', nr*w) (layers.c:119)"""
    L = lib()
    return mat_to_numpy(L.orc_window(NpMat(feat).ptr, w, stride), L.orc_free_mat)


def lstm(xaff, sW, p, backward=False):
    """xaff (T, 4S), sW (4S, S), p (3S,) -> (T, S)"""
    L = lib()
    f = L.orc_lstm_backward if backward else L.orc_lstm_forward
    return mat_to_numpy(f(NpMat(xaff).ptr, NpMat(sW).ptr, NpMat(np.asarray(p).reshape(1, -1)).ptr, None), L.orc_free_mat)


def events_trunk(om, feature3, upto=2):
    L = lib()
    return mat_to_numpy(L.orc_events_trunk(om.ptr, NpMat(feature3).ptr, upto), L.orc_free_mat)


def events_posterior(om, feature3, min_prob=1e-5, tempW=1.0, tempb=1.0, log=True, padded=False):
    L = lib()
    return mat_to_numpy(L.orc_events_posterior_from_features(om.ptr, NpMat(feature3).ptr, min_prob, tempW, tempb, log),
                        L.orc_free_mat, padded=padded)
