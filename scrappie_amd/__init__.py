"""scrappie_amd -- Python host side over libscrappie_hip.so (the MI355X-native
`scrappie raw` hot path).

It mirrors the reference's `scrappy` binding (python/scrappy/__init__.py:47-430:
RawTable, ScrappyMatrix, calc_post, decode_post, basecall_raw, get_model_stride)
so code and tests written against `scrappy` read the same, and adds the batched
`Engine`, which is the fast path.  Bindings are plain ctypes over the C ABI in
include/scrappie_hip.h -- no torch types cross the boundary.

The HIP library is required: importing the compute entry points without
scrappie_amd/libscrappie_hip.so raises (there is no CPU fallback).
"""
import ctypes as C
import os
import sys
import subprocess

import numpy as np

from . import model as _model

__version__ = "0.1.0"
_HERE = os.path.dirname(os.path.abspath(__file__))
# the experiments build (kernel forms measured and not adopted, cycle stamps: make -C csrc ../libscrappie_hip_exp.so) is loaded only when
# SCRAPPIE_HIP_LIB names it -- the tests of those forms do, in processes of their own
EXP_LIB_PATH = os.path.join(_HERE, "libscrappie_hip_exp.so")
LIB_PATH = os.environ.get("SCRAPPIE_HIP_LIB") or os.path.join(_HERE, "libscrappie_hip.so")

ftype = np.float32
vsize = 4


class _Mat(C.Structure):
    _fields_ = [("nr", C.c_size_t), ("nrq", C.c_size_t), ("nc", C.c_size_t),
                ("stride", C.c_size_t), ("data", C.c_void_p)]


class _RawTable(C.Structure):
    _fields_ = [("uuid", C.c_char_p), ("n", C.c_size_t), ("start", C.c_size_t),
                ("end", C.c_size_t), ("raw", C.POINTER(C.c_float))]


class Params(C.Structure):
    """scrappie_hip_params; defaults are the CLI's (src/scrappie_raw.c:98-121)."""
    _fields_ = [("min_prob", C.c_float), ("tempW", C.c_float), ("tempb", C.c_float),
                ("stay_pen", C.c_float), ("skip_pen", C.c_float), ("local_pen", C.c_float),
                ("use_slip", C.c_int), ("homopolymer", C.c_int), ("want_pos", C.c_int)]


class _Call(C.Structure):
    _fields_ = [("score", C.c_float), ("nblock", C.c_size_t), ("basecall", C.c_void_p),
                ("basecall_length", C.c_size_t), ("pos", C.POINTER(C.c_int))]


class Timing(C.Structure):
    _fields_ = [("conv_ms", C.c_float), ("affine_ms", C.c_float), ("gru_ms", C.c_float),
                ("ff_ms", C.c_float), ("decode_ms", C.c_float), ("backtrace_ms", C.c_float),
                ("total_ms", C.c_float), ("n_gru_launches", C.c_int), ("n_affine_launches", C.c_int),
                ("gru_flops", C.c_double), ("affine_flops", C.c_double), ("ff_flops", C.c_double),
                ("fused_ms", C.c_float), ("n_fused_launches", C.c_int), ("fused_flops", C.c_double),
                ("stitch_ms", C.c_float)]


def build(verbose=False):
    """Compile libscrappie_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "../libscrappie_hip.so"], check=True,
                   stdout=None if verbose else subprocess.DEVNULL)


_lib = None


def _one_hip_runtime():
    """A process must hold ONE HIP runtime: a second copy finds no device once the first has initialised
    (hipGetDeviceCount: "no ROCm-capable device is detected").  torch wheels bundle their own copy (torch/lib/libamdhip64.so) and ask
    for it as "libamdhip64.so", which does not match the soname (libamdhip64.so.7) of a copy libscrappie_hip.so brought in from
    /opt/rocm/lib earlier -- so `import scrappie_amd; ...; import torch` used to end with two.  Loading torch's copy first (when a torch
    is installed and not imported yet) makes this library bind to it by soname and torch find it again by file: one runtime whatever the
    import order.  SCRAPPIE_HIP_SYSTEM_RUNTIME=1 keeps /opt/rocm's (then import torch BEFORE this library, or not at all)."""
    import importlib.util
    if "torch" in sys.modules or os.environ.get("SCRAPPIE_HIP_SYSTEM_RUNTIME"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.origin:
        p = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(p):
            C.CDLL(p, mode=C.RTLD_GLOBAL)


def lib():
    """The loaded C library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(scrappie_amd has no CPU fallback)" % LIB_PATH)
    # an engine's four streams want a hardware queue each (scrappie_hip.hip: hw_queue_default); effective only if HIP
    # has not initialised in this process yet -- with torch imported first, set GPU_MAX_HW_QUEUES before importing it
    if not os.environ.get("SH_NO_PY_QUEUE_DEFAULT"):        # (switch for checking the C side's own default)
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    _one_hip_runtime()
    L = C.CDLL(LIB_PATH, mode=C.RTLD_LOCAL)
    PM = C.POINTER(_Mat)
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
    L.scrappie_hip_last_error.restype = C.c_char_p
    L.scrappie_hip_device_count.restype = C.c_int
    L.scrappie_hip_engine_create.restype = C.c_void_p
    L.scrappie_hip_engine_create.argtypes = [C.c_int]
    L.scrappie_hip_engine_destroy.argtypes = [C.c_void_p]
    L.scrappie_hip_default_params.restype = Params
    L.scrappie_hip_load_model.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    L.scrappie_hip_load_model_mem.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]
    L.scrappie_hip_find_model.argtypes = [C.c_void_p, C.c_char_p]
    L.scrappie_hip_register_model.argtypes = [C.c_char_p, C.c_char_p]
    L.scrappie_hip_basecall_batch.argtypes = [C.c_void_p, C.c_int, C.POINTER(_RawTable), C.c_size_t,
                                              C.POINTER(Params), C.POINTER(_Call)]
    L.scrappie_hip_basecall_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_uint64),
                                               C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(Params),
                                               C.POINTER(_Call)]
    L.scrappie_hip_run_device.restype = C.c_long
    L.scrappie_hip_run_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_uint64),
                                          C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(Params)]
    L.scrappie_hip_collect.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(_Call), C.c_size_t]
    L.scrappie_hip_free_calls.argtypes = [C.POINTER(_Call), C.c_size_t]
    L.scrappie_hip_basecall_batch_multi.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_size_t, C.POINTER(_RawTable),
                                                    C.c_size_t, C.POINTER(Params), C.POINTER(_Call)]
    L.scrappie_hip_plan_dynamic.restype = C.c_long
    L.scrappie_hip_plan_dynamic.argtypes = [C.POINTER(C.c_uint32), C.c_size_t, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t,
                                            C.POINTER(C.c_uint32), C.POINTER(C.c_size_t), C.c_size_t]
    L.scrappie_hip_set_decoder_input.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t]
    L.scrappie_hip_set_trunk_input.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t]
    L.scrappie_hip_debug_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.scrappie_hip_debug_fetch.restype = C.c_longlong
    L.scrappie_hip_debug_fetch.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]
    L.scrappie_hip_debug_stitch.restype = C.c_long
    L.scrappie_hip_debug_stitch.argtypes = [C.c_void_p, ip, fp, C.c_size_t, C.c_int, C.c_int, C.c_char_p, C.c_size_t, ip, ip]
    L.scrappie_hip_posterior.restype = PM
    L.scrappie_hip_posterior.argtypes = [C.c_void_p, C.c_int, _RawTable, C.c_float, C.c_float, C.c_float, C.c_bool]
    L.scrappie_hip_trunk.restype = PM
    L.scrappie_hip_trunk.argtypes = [C.c_void_p, C.c_int, _RawTable, C.c_int]
    L.scrappie_hip_min_samples.restype = C.c_size_t
    L.scrappie_hip_min_samples.argtypes = [C.c_void_p, C.c_int]
    L.scrappie_hip_model_stride.argtypes = [C.c_void_p, C.c_int]
    L.scrappie_hip_set_profiling.argtypes = [C.c_void_p, C.c_int]
    L.scrappie_hip_get_timing.argtypes = [C.c_void_p, C.POINTER(Timing)]
    L.scrappie_hip_set_max_launch_reads.argtypes = [C.c_void_p, C.c_size_t]
    L.scrappie_hip_set_max_launch_blocks.argtypes = [C.c_void_p, C.c_size_t]
    L.scrappie_hip_plan_groups.restype = C.c_long
    L.scrappie_hip_plan_groups.argtypes = [C.POINTER(C.c_uint32), C.c_size_t, C.c_int, C.c_size_t, C.c_size_t,
                                           C.POINTER(C.c_size_t), C.c_size_t]
    L.scrappie_hip_device_alloc.restype = C.c_void_p
    L.scrappie_hip_device_alloc.argtypes = [C.c_void_p, C.c_size_t]
    L.scrappie_hip_device_free.argtypes = [C.c_void_p, C.c_void_p]
    L.scrappie_hip_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.scrappie_hip_synchronize.argtypes = [C.c_void_p]
    L.scrappie_hip_format_fasta.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_char_p, C.c_bool, C.c_char_p,
                                            C.POINTER(_Call), C.c_size_t, C.c_size_t, C.c_size_t]
    # per-read reference surface (python/pyscrap.h)
    for nm in ("nanonet_raw_posterior", "nanonet_rgrgr_r94_posterior", "nanonet_rgrgr_r941_posterior", "nanonet_rgrgr_r10_posterior",
               "nanonet_rnnrf_r94_transitions"):
        getattr(L, nm).restype = PM
        getattr(L, nm).argtypes = [_RawTable, C.c_float, C.c_float, C.c_float, C.c_bool]
    L.decode_transducer.restype = C.c_float
    L.decode_transducer.argtypes = [PM, C.c_float, C.c_float, C.c_float, ip, C.c_bool]
    L.overlapper.restype = C.c_void_p
    L.overlapper.argtypes = [ip, C.c_size_t, C.c_int, ip]
    L.decode_crf.restype = C.c_float
    L.decode_crf.argtypes = [PM, ip]
    L.crfpath_to_basecall.restype = C.c_void_p
    L.crfpath_to_basecall.argtypes = [ip, C.c_size_t, ip]
    L.posterior_crf.restype = PM
    L.posterior_crf.argtypes = [PM]
    L.homopolymer_path.argtypes = [PM, ip, C.c_int]
    L.medmad_normalise_array.argtypes = [fp, C.c_size_t]
    L.trim_raw_by_mad.restype = _RawTable
    L.trim_raw_by_mad.argtypes = [_RawTable, C.c_size_t, C.c_float]
    L.trim_and_segment_raw.restype = _RawTable
    L.trim_and_segment_raw.argtypes = [_RawTable, C.c_size_t, C.c_size_t, C.c_size_t, C.c_float]
    L.mat_from_array.restype = PM
    L.mat_from_array.argtypes = [fp, C.c_size_t, C.c_size_t]
    L.make_scrappie_matrix.restype = PM
    L.make_scrappie_matrix.argtypes = [C.c_size_t, C.c_size_t]
    L.free_scrappie_matrix.restype = PM
    L.free_scrappie_matrix.argtypes = [PM]
    L.scrappie_hip_prep_create.restype = C.c_void_p
    L.scrappie_hip_prep_create.argtypes = [C.c_int]
    L.scrappie_hip_prep_destroy.argtypes = [C.c_void_p]
    L.scrappie_hip_prep_run.argtypes = [C.c_void_p, C.c_int, C.POINTER(_RawTable), C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
                                        C.c_float, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32),
                                        C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.scrappie_hip_prep_begin.restype = C.c_void_p
    L.scrappie_hip_prep_begin.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    L.scrappie_hip_prep_alloc.restype = C.c_void_p
    L.scrappie_hip_prep_alloc.argtypes = [C.c_void_p, C.c_size_t]
    L.scrappie_hip_prep_owns.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.scrappie_hip_prep_fetch.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_size_t, fp]
    L.scrappie_hip_prep_timing.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
    L.get_raw_model_stride_from_string.argtypes = [C.c_char_p]
    L.get_raw_model.argtypes = [C.c_char_p]
    _lib = L
    return L


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def last_error():
    return lib().scrappie_hip_last_error().decode()


def _take_string(ptr):
    if not ptr:
        return None
    s = C.string_at(ptr).decode()
    _libc.free(ptr)
    return s


# ---------------------------------------------------------------------------
# scrappy-compatible objects (python/scrappy/__init__.py:47-273)
# ---------------------------------------------------------------------------
class RawTable(object):
    """Representation of a scrappie `raw_table` (python/scrappy/__init__.py:47-112)."""

    def __init__(self, data, start=0, end=None):
        if end is None:
            end = len(data)
        self._data = np.ascontiguousarray(np.asarray(data).astype(ftype, order='C', copy=True))
        self._rt = _RawTable(None, len(self._data), start, end,
                             self._data.ctypes.data_as(C.POINTER(C.c_float)))

    def data(self, as_numpy=False):
        if as_numpy:
            return np.copy(self._data[self.start:self.end])
        return self._rt

    @property
    def start(self):
        return self._rt.start

    @property
    def end(self):
        return self._rt.end

    def trim(self, start=200, end=10, varseg_chunk=100, varseg_thresh=0.0):
        """python/scrappy/__init__.py:92-133 (== trim_and_segment_raw, scrappie_common.c:11-17,
        except that the data array is kept when the window comes out empty)."""
        rt = lib().trim_raw_by_mad(self._rt, varseg_chunk, varseg_thresh)
        rt.start = rt.start + start if (rt.n - rt.start) > start else rt.n
        rt.end = rt.end - end if rt.end > end else 0
        if rt.start >= rt.end:
            rt.start, rt.end = 0, 0
        self._rt = rt
        return self

    def scale(self):
        """python/scrappy/__init__.py:107-112, :136-147"""
        n = self._rt.end - self._rt.start
        if n > 0:
            ptr = C.cast(C.addressof(self._rt.raw.contents) + 4 * self._rt.start, C.POINTER(C.c_float))
            lib().medmad_normalise_array(ptr, n)
        return self


class ScrappyMatrix(object):
    """Owns a `scrappie_matrix` returned by the library (python/scrappy/__init__.py:150-200)."""

    def __init__(self, scrappy_matrix):
        self._data = scrappy_matrix
        self.shape = (self._data.contents.nc, self._data.contents.nr)

    def __del__(self):
        if getattr(self, "_data", None):
            lib().free_scrappie_matrix(self._data)
            self._data = None

    def data(self, as_numpy=False, sloika=True):
        if as_numpy:
            return _scrappie_to_numpy(self._data, sloika=sloika)
        return self._data

    @classmethod
    def from_numpy(cls, array, sloika=True):
        """array is (blocks, states); with sloika=True the stay state is first."""
        a = np.asarray(array, dtype=ftype)
        if sloika:
            a = np.hstack((a[:, 1:], a[:, 0:1]))
        a = np.ascontiguousarray(a)
        m = lib().mat_from_array(a.ctypes.data_as(C.POINTER(C.c_float)), a.shape[1], a.shape[0])
        return cls(m)


def _scrappie_to_numpy(matrix, sloika=True):
    """python/scrappy/__init__.py:247-273: drop the SSE padding; optionally roll
    the stay state to the front (sloika order)."""
    m = matrix.contents
    flat = np.ctypeslib.as_array(C.cast(m.data, C.POINTER(C.c_float)), shape=(m.nc * vsize * m.nrq,))
    np_matrix = flat.reshape(m.nc, vsize * m.nrq)[:, :m.nr]
    if sloika:
        np_matrix = np.hstack((np_matrix[:, m.nr - 1:m.nr], np_matrix[:, 0:m.nr - 1]))
    return np.array(np_matrix, dtype=ftype, order='C', copy=True)   # a copy: the matrix may be freed


_model_fn_ = {
    'raw_r94': 'nanonet_raw_posterior',
    'rgrgr_r94': 'nanonet_rgrgr_r94_posterior',
    'rgrgr_r941': 'nanonet_rgrgr_r941_posterior',
    'rgrgr_r10': 'nanonet_rgrgr_r10_posterior',
    'rnnrf_r94': 'nanonet_rnnrf_r94_transitions',
}


class _Event(C.Structure):          # scrappie_structures.h:8-15
    _fields_ = [("start", C.c_uint64), ("length", C.c_float), ("mean", C.c_float), ("stdv", C.c_float),
                ("pos", C.c_int), ("state", C.c_int)]


class _EventTable(C.Structure):     # scrappie_structures.h:17-22
    _fields_ = [("n", C.c_size_t), ("start", C.c_size_t), ("end", C.c_size_t), ("event", C.POINTER(_Event))]


def event_features(events, start=0, end=None):
    """Windowed, studentised features of an event table (networks.c:155-157) as an
    (nevent, 12) float32 array -- the input of an events model.  `events`: structured array
    with the reference's event_t layout (scrappie_amd.synth.EVENT_DTYPE)."""
    ev = np.ascontiguousarray(events)
    if ev.dtype.itemsize != C.sizeof(_Event):
        raise ValueError("events must have the event_t layout (%d bytes per event)" % C.sizeof(_Event))
    end = len(ev) if end is None else end
    et = _EventTable(len(ev), start, end, C.cast(ev.ctypes.data, C.POINTER(_Event)))
    out = np.zeros((end - start, 12), dtype=ftype)
    L = lib()
    L.scrappie_hip_event_features.restype = C.c_int
    L.scrappie_hip_event_features.argtypes = [_EventTable, C.POINTER(C.c_float)]
    if L.scrappie_hip_event_features(et, out.ctypes.data_as(C.POINTER(C.c_float))) != 0:
        raise RuntimeError("event_features failed")
    return out


def register_model(name, path):
    """Bind a `.scrm` weight container to a reference model name for the per-read
    surface (the reference compiles its weights in; here they are data)."""
    if lib().scrappie_hip_register_model(name.encode(), os.fsencode(path)) < 0:
        raise RuntimeError(last_error())


def calc_post(rt, model='rgrgr_r94', min_prob=1e-6, log=True, tempW=1.0, tempb=1.0):
    """python/scrappy/__init__.py:276-299"""
    if not log and model == 'rnnrf_r94':
        raise ValueError("Returning non-log transformed matrix not supported for model type 'rnnrf_r94'.")
    if not isinstance(rt, RawTable):
        raise TypeError('`rt` should be a RawTable.')
    try:
        fn = getattr(lib(), _model_fn_[model])
    except KeyError:
        raise KeyError("Model type '{}' not recognised.".format(model))
    matrix = fn(rt.data(), min_prob, tempW, tempb, log)
    if not matrix:
        raise RuntimeError('An unknown error occurred during posterior calculation: ' + last_error())
    return ScrappyMatrix(matrix)


def _decode_post(post, stay_pen=0.0, skip_pen=0.0, local_pen=2.0, use_slip=False):
    """python/scrappy/__init__.py:323-346"""
    nblock, nstate = post.shape
    path = np.zeros(nblock + 1, dtype=np.int32)
    score = lib().decode_transducer(post.data(), stay_pen, skip_pen, local_pen,
                                    path.ctypes.data_as(C.POINTER(C.c_int)), use_slip)
    pos = np.zeros(nblock + 1, dtype=np.int32)
    basecall = lib().overlapper(path.ctypes.data_as(C.POINTER(C.c_int)), nblock + 1, nstate - 1,
                                pos.ctypes.data_as(C.POINTER(C.c_int)))
    return _take_string(basecall), score, pos


def _decode_post_crf(post):
    """python/scrappy/__init__.py:349-365"""
    nblock, nstate = post.shape
    path = np.zeros(nblock + 1, dtype=np.int32)
    score = lib().decode_crf(post.data(), path.ctypes.data_as(C.POINTER(C.c_int)))
    pos = np.zeros(nblock + 1, dtype=np.int32)
    basecall = lib().crfpath_to_basecall(path.ctypes.data_as(C.POINTER(C.c_int)), nblock,
                                         pos.ctypes.data_as(C.POINTER(C.c_int)))
    return _take_string(basecall), score, pos


_decoders_ = {'raw_r94': _decode_post, 'rgrgr_r94': _decode_post, 'rgrgr_r941': _decode_post, 'rgrgr_r10': _decode_post,
              'rnnrf_r94': _decode_post_crf}


def decode_post(post, model='rgrgr_r94', **kwargs):
    """python/scrappy/__init__.py:302-320"""
    if not isinstance(post, ScrappyMatrix):
        raise TypeError('`post` should be a ScrappyMatrix.')
    try:
        decoder = _decoders_[model]
    except KeyError:
        raise KeyError("Model type '{}' not recognised.".format(model))
    return decoder(post, **kwargs)


def get_model_stride(model):
    """python/scrappy/__init__.py:389-400"""
    stride = lib().get_raw_model_stride_from_string(model.encode())
    if stride == -1:
        raise ValueError("Invalid scrappie model '{}'.".format(model))
    return stride


def basecall_raw(data, model='rgrgr_r94', with_base_probs=False, **kwargs):
    """python/scrappy/__init__.py:403-430: trim -> scale -> posterior (min_prob
    1e-6) -> decode; no homopolymer correction on this path (quirk Q10)."""
    raw = RawTable(data)
    raw.trim().scale()
    post = calc_post(raw, model, log=True)
    seq, score, pos = decode_post(post, model, **kwargs)
    base_probs = None
    if with_base_probs:
        bp = lib().posterior_crf(post.data())
        base_probs = _scrappie_to_numpy(bp, sloika=False)
        lib().free_scrappie_matrix(bp)
    return seq, score, pos, raw.start, raw.end, base_probs


# ---------------------------------------------------------------------------
# batched engine (additive; the fast path)
# ---------------------------------------------------------------------------
def plan_tail(lengths, stride, max_long_blocks=0):
    """scrappie_hip_plan_tail (host only): boolean array, True for the chain-bound reads a call runs beside the others"""
    ln = np.ascontiguousarray(lengths, dtype=np.uint32)
    flags = np.zeros(len(ln), np.uint8)
    L = lib()
    L.scrappie_hip_plan_tail.restype = C.c_long
    L.scrappie_hip_plan_tail.argtypes = [C.POINTER(C.c_uint32), C.c_size_t, C.c_int, C.c_size_t, C.POINTER(C.c_ubyte)]
    n = L.scrappie_hip_plan_tail(ln.ctypes.data_as(C.POINTER(C.c_uint32)), len(ln), stride, max_long_blocks, flags.ctypes.data_as(C.POINTER(C.c_ubyte)))
    if n < 0:
        raise RuntimeError("plan_tail: invalid arguments")
    assert int(flags.sum()) == n
    return flags.astype(bool)


def plan_dynamic(lengths, stride, nengine, max_reads=16384, max_blocks=0):
    """The hand-out plan of basecall_multi (host only): (order, starts): read indices sorted by length,
    longest first, and the first position of each launch group in that order."""
    ln = np.ascontiguousarray(lengths, dtype=np.uint32)
    n = len(ln)
    order = np.zeros(max(n, 1), dtype=np.uint32)
    starts = np.zeros(max(n, 1), dtype=np.uintp)
    ng = lib().scrappie_hip_plan_dynamic(ln.ctypes.data_as(C.POINTER(C.c_uint32)), n, stride, nengine, max_reads, max_blocks,
                                         order.ctypes.data_as(C.POINTER(C.c_uint32)),
                                         starts.ctypes.data_as(C.POINTER(C.c_size_t)), len(starts))
    if ng < 0:
        raise RuntimeError("plan_dynamic: a read alone exceeds max_blocks")
    return order[:n], starts[:ng]


def basecall_multi(engines, signals, model='rgrgr_r94', params=None):
    """scrappie_hip_basecall_batch_multi: `signals` spread over several engines (one per GPU), launch groups
    handed out from an atomic cursor over the reads sorted by length.  Returns the calls in input order."""
    n = len(signals)
    p = params or engines[0].default_params()
    keep = [np.ascontiguousarray(s, dtype=ftype) for s in signals]
    rts = (_RawTable * max(n, 1))()
    for i, s in enumerate(keep):
        rts[i] = _RawTable(None, len(s), 0, len(s), s.ctypes.data_as(C.POINTER(C.c_float)))
    calls = (_Call * max(n, 1))()
    hs = (C.c_void_p * len(engines))(*[e._h for e in engines])
    ms = (C.c_int * len(engines))(*[e._models[model] for e in engines])
    if lib().scrappie_hip_basecall_batch_multi(hs, ms, len(engines), rts, n, C.byref(p), calls) != 0:
        raise RuntimeError("basecall_batch_multi: " + last_error())
    return Engine._unpack(calls, n, p.want_pos)


class Prep(object):
    """Signal preparation of a batch on the device (scrappie_hip_prep_*, k_p0): trim_and_segment_raw +
    medmad_normalise_array of the reference (scrappie_raw.c:270-277) for every read of a batch in one launch."""

    def __init__(self, device=0):
        self._h = lib().scrappie_hip_prep_create(device)
        if not self._h:
            raise RuntimeError("prep_create: " + last_error())

    def close(self):
        if self._h:
            lib().scrappie_hip_prep_destroy(self._h)
            self._h = None

    def run(self, raws, trim_start=200, trim_end=10, varseg_chunk=100, varseg_thresh=0.0, slot=0, windows=None, stage_capacity=None):
        """raws: list of float32 arrays of RAW samples (windows: optional list of (start, end) at entry).
        stage_capacity (samples): place the reads in the slot's pinned staging buffer first, as a loader does
        (scrappie_hip_prep_begin / _alloc); the ones that do not fit stay where they are and are gathered by the call.
        Returns (device pointer, offsets, lengths, start, end): offsets / lengths as run_device / basecall_device take them."""
        n = len(raws)
        self._keep = [np.ascontiguousarray(r, dtype=ftype) for r in raws]
        rts = (_RawTable * max(n, 1))()
        ctx = None
        self.n_staged = 0
        if stage_capacity is not None:
            ctx = lib().scrappie_hip_prep_begin(self._h, slot, int(stage_capacity))
            if not ctx:
                raise RuntimeError("prep_begin: " + last_error())
        for i, r in enumerate(self._keep):
            st, en = windows[i] if windows is not None else (0, len(r))
            ptr = lib().scrappie_hip_prep_alloc(ctx, len(r)) if ctx else None
            if ptr:
                C.memmove(ptr, r.ctypes.data, r.nbytes)
                assert lib().scrappie_hip_prep_owns(self._h, slot, ptr) == 1
                self.n_staged += 1
                rts[i] = _RawTable(None, len(r), st, en, C.cast(ptr, C.POINTER(C.c_float)))
                continue
            rts[i] = _RawTable(None, len(r), st, en, r.ctypes.data_as(C.POINTER(C.c_float)))
        d = C.c_void_p()
        off = np.zeros(max(n, 1), np.uint64); ln = np.zeros(max(n, 1), np.uint32)
        st = np.zeros(max(n, 1), np.uint32); en = np.zeros(max(n, 1), np.uint32)
        u64, u32 = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
        if lib().scrappie_hip_prep_run(self._h, slot, rts, n, trim_start, trim_end, varseg_chunk, varseg_thresh, C.byref(d),
                                       off.ctypes.data_as(u64), ln.ctypes.data_as(u32), st.ctypes.data_as(u32), en.ctypes.data_as(u32)) != 0:
            raise RuntimeError("prep_run: " + last_error())
        return d.value, off[:n], ln[:n], st[:n], en[:n]

    def fetch(self, offset, count, slot=0):
        out = np.empty(int(count), ftype)
        if count and lib().scrappie_hip_prep_fetch(self._h, slot, int(offset), int(count), out.ctypes.data_as(C.POINTER(C.c_float))) != 0:
            raise RuntimeError("prep_fetch: " + last_error())
        return out

    def timing(self, slot=0):
        t = (C.c_double * 3)()
        lib().scrappie_hip_prep_timing(self._h, slot, t)
        return {"gather_ms": t[0], "h2d_ms": t[1], "k_p0_ms": t[2]}


class Engine(object):
    """One GPU.  `basecall(signals)` takes a list of trimmed, normalised float32
    arrays and returns a list of dicts (bases, score, nblock[, pos]); reads are
    coalesced into launch groups, decoded on device."""

    def __init__(self, device=0):
        self._h = lib().scrappie_hip_engine_create(device)
        if not self._h:
            raise RuntimeError("engine_create(%d): %s" % (device, last_error()))
        self.device = device
        self._models = {}

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self, "_alt_dev", None):
                self.set_decoder_input(None)
            if getattr(self, "_trk_dev", None):
                self.set_trunk_input(None)
            lib().scrappie_hip_engine_destroy(self._h)
            self._h = None

    __del__ = close

    def load_model(self, name, weights):
        """`weights` is a model dict (scrappie_amd.model) or a path to a .scrm file."""
        if isinstance(weights, dict):
            import tempfile
            with tempfile.NamedTemporaryFile(suffix=".scrm", delete=False) as fh:
                path = fh.name
            try:
                _model.save_model(weights, path)
                h = lib().scrappie_hip_load_model(self._h, name.encode(), os.fsencode(path))
            finally:
                os.unlink(path)
        else:
            h = lib().scrappie_hip_load_model(self._h, name.encode(), os.fsencode(weights))
        if h < 0:
            raise RuntimeError("load_model(%s): %s" % (name, last_error()))
        self._models[name] = h
        return h

    def default_params(self, **kw):
        p = lib().scrappie_hip_default_params()
        for k, v in kw.items():
            setattr(p, k, v)
        return p

    def min_samples(self, model):
        return lib().scrappie_hip_min_samples(self._h, self._models[model])

    def set_profiling(self, on=True):
        lib().scrappie_hip_set_profiling(self._h, 1 if on else 0)

    def set_max_launch_reads(self, n):
        lib().scrappie_hip_set_max_launch_reads(self._h, n)

    def set_max_launch_blocks(self, n):
        lib().scrappie_hip_set_max_launch_blocks(self._h, n)

    def timing(self):
        t = Timing()
        lib().scrappie_hip_get_timing(self._h, C.byref(t))
        return {f[0]: getattr(t, f[0]) for f in Timing._fields_}

    @staticmethod
    def _unpack(calls, n, want_pos):
        out = []
        for i in range(n):
            c = calls[i]
            if not c.basecall:
                out.append(None)
                continue
            d = dict(bases=C.string_at(c.basecall).decode(), score=float(c.score), nblock=int(c.nblock))
            if want_pos and c.pos:
                d["pos"] = np.ctypeslib.as_array(c.pos, shape=(c.nblock + 1,)).copy()
            out.append(d)
        lib().scrappie_hip_free_calls(calls, n)
        return out

    def basecall(self, signals, model='rgrgr_r94', params=None):
        n = len(signals)
        p = params or self.default_params()
        keep = [np.ascontiguousarray(s, dtype=ftype) for s in signals]
        rts = (_RawTable * n)()
        for i, s in enumerate(keep):
            rts[i] = _RawTable(None, len(s), 0, len(s), s.ctypes.data_as(C.POINTER(C.c_float)))
        calls = (_Call * n)()
        if lib().scrappie_hip_basecall_batch(self._h, self._models[model], rts, n, C.byref(p), calls) != 0:
            raise RuntimeError("basecall_batch: " + last_error())
        return self._unpack(calls, n, p.want_pos)

    # -- device-resident path (bench) ------------------------------------
    def basecall_deferred(self, signals, model='rgrgr_r94', params=None):
        """scrappie_hip_basecall_batch_deferred: (calls, ticket, deferred) -- calls[i] is None where deferred[i]; pass the returned
        ticket to collect_deferred() (the signals are kept alive with it)."""
        n = len(signals)
        p = params or self.default_params()
        keep = [np.ascontiguousarray(s, dtype=ftype) for s in signals]
        rts = (_RawTable * max(n, 1))()
        for i, s in enumerate(keep):
            rts[i] = _RawTable(None, len(s), 0, len(s), s.ctypes.data_as(C.POINTER(C.c_float)))
        calls = (_Call * max(n, 1))()
        flags = np.zeros(max(n, 1), np.uint8)
        L = lib()
        L.scrappie_hip_basecall_batch_deferred.restype = C.c_long
        L.scrappie_hip_basecall_batch_deferred.argtypes = [C.c_void_p, C.c_int, C.POINTER(_RawTable), C.c_size_t, C.POINTER(Params),
                                                           C.POINTER(_Call), C.POINTER(C.c_ubyte)]
        tk = L.scrappie_hip_basecall_batch_deferred(self._h, self._models[model], rts, n, C.byref(p), calls, flags.ctypes.data_as(C.POINTER(C.c_ubyte)))
        if tk < 0:
            raise RuntimeError("basecall_batch_deferred: " + last_error())
        out = Engine._unpack(calls, n, p.want_pos)
        deferred = flags[:n].astype(bool)
        if tk > 0:
            self._deferred = getattr(self, "_deferred", {})
            self._deferred[tk] = (keep, rts, int(deferred.sum()), p.want_pos)
        return out, tk, deferred

    def basecall_device_deferred(self, dptr, offsets, lengths, model='rgrgr_r94', params=None):
        """scrappie_hip_basecall_device_deferred on signals already in HBM (e.g. from Prep.run): (calls, ticket, deferred) as above;
        the device buffer may be reused as soon as this returns."""
        p = params or self.default_params()
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        ln = np.ascontiguousarray(lengths, dtype=np.uint32)
        n = len(ln)
        calls = (_Call * max(n, 1))()
        flags = np.zeros(max(n, 1), np.uint8)
        L = lib()
        L.scrappie_hip_basecall_device_deferred.restype = C.c_long
        L.scrappie_hip_basecall_device_deferred.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_size_t,
                                                            C.POINTER(Params), C.POINTER(_Call), C.POINTER(C.c_ubyte)]
        tk = L.scrappie_hip_basecall_device_deferred(self._h, self._models[model], dptr, off.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                     ln.ctypes.data_as(C.POINTER(C.c_uint32)), n, C.byref(p), calls,
                                                     flags.ctypes.data_as(C.POINTER(C.c_ubyte)))
        if tk < 0:
            raise RuntimeError("basecall_device_deferred: " + last_error())
        out = Engine._unpack(calls, n, p.want_pos)
        deferred = flags[:n].astype(bool)
        if tk > 0:
            self._deferred = getattr(self, "_deferred", {})
            self._deferred[tk] = (None, None, int(deferred.sum()), p.want_pos)
        return out, tk, deferred

    def collect_deferred(self, ticket, wait=True):
        """the calls of a ticket's deferred reads in their call's order; None if wait is False and they are not ready"""
        keep, rts, nl, want_pos = self._deferred[ticket]
        calls = (_Call * max(nl, 1))()
        L = lib()
        L.scrappie_hip_deferred_collect.restype = C.c_long
        L.scrappie_hip_deferred_collect.argtypes = [C.c_void_p, C.c_long, C.POINTER(_Call), C.c_size_t, C.c_int]
        k = L.scrappie_hip_deferred_collect(self._h, ticket, calls, nl, 1 if wait else 0)
        if k == -2:
            return None
        del self._deferred[ticket]
        if k < 0:
            raise RuntimeError("deferred_collect: " + last_error())
        return Engine._unpack(calls, k, want_pos)

    def upload(self, flat_signal):
        flat = np.ascontiguousarray(flat_signal, dtype=ftype)
        d = lib().scrappie_hip_device_alloc(self._h, flat.nbytes)
        if not d:
            raise RuntimeError("device_alloc: " + last_error())
        if lib().scrappie_hip_memcpy_h2d(self._h, d, flat.ctypes.data, flat.nbytes) != 0:
            raise RuntimeError("memcpy_h2d: " + last_error())
        return d

    def free(self, dptr):
        lib().scrappie_hip_device_free(self._h, dptr)

    def run_device(self, dptr, offsets, lengths, model='rgrgr_r94', params=None):
        """Launch the device pipeline on reads already in HBM; returns #blocks."""
        p = params or self.default_params()
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        ln = np.ascontiguousarray(lengths, dtype=np.uint32)
        r = lib().scrappie_hip_run_device(self._h, self._models[model], dptr,
                                          off.ctypes.data_as(C.POINTER(C.c_uint64)),
                                          ln.ctypes.data_as(C.POINTER(C.c_uint32)), len(ln), C.byref(p))
        if r < 0:
            raise RuntimeError("run_device: " + last_error())
        return r

    def collect(self, n, params=None, raw=False):
        p = params or self.default_params()
        calls = (_Call * n)()
        if lib().scrappie_hip_collect(self._h, C.byref(p), calls, n) != 0:
            raise RuntimeError("collect: " + last_error())
        if raw:     # bench: only count bases, then free
            nb = int(np.frombuffer(calls, dtype=np.uint64).reshape(n, C.sizeof(_Call) // 8)[:, 3].sum()) if n else 0
            lib().scrappie_hip_free_calls(calls, n)
            return nb
        return self._unpack(calls, n, p.want_pos)

    def synchronize(self):
        lib().scrappie_hip_synchronize(self._h)

    def set_decoder_input(self, probs):
        """Measurement / test hook (scrappie_hip_set_decoder_input): `probs` = list of (T, NS) float32
        probability matrices (reference state order); read i of every later launch group is decoded
        from probs[i % len(probs)] instead of the network's own posterior.  None switches it off."""
        if getattr(self, "_alt_dev", None):
            lib().scrappie_hip_set_decoder_input(self._h, None, None, 0)
            self.free(self._alt_dev)
            self._alt_dev = None
        if probs is None:
            return
        mats = [np.ascontiguousarray(p, dtype=ftype) for p in probs]
        off = np.zeros(len(mats), dtype=np.uint64)
        tot = 0
        for i, m in enumerate(mats):
            off[i] = tot
            tot += m.size
        self._alt_dev = self.upload(np.concatenate([m.ravel() for m in mats]))
        if lib().scrappie_hip_set_decoder_input(self._h, self._alt_dev, off.ctypes.data_as(C.POINTER(C.c_uint64)), len(mats)) != 0:
            raise RuntimeError("set_decoder_input: " + last_error())

    def set_trunk_input(self, trunks):
        """Measurement / test hook (scrappie_hip_set_trunk_input): `trunks` = list of (T, S) float32 activation
        matrices; in every later launch group the output layer of read i reads trunks[i % len(trunks)] instead of
        the trunk's own output, so the default decode path (S1 inside the decoder) sees the posteriors those
        activations encode.  None switches it off."""
        if getattr(self, "_trk_dev", None):
            lib().scrappie_hip_set_trunk_input(self._h, None, None, 0)
            self.free(self._trk_dev)
            self._trk_dev = None
        if trunks is None:
            return
        mats = [np.ascontiguousarray(t, dtype=ftype) for t in trunks]
        off = np.zeros(len(mats), dtype=np.uint64)
        tot = 0
        for i, m in enumerate(mats):
            off[i] = tot
            tot += m.size
        self._trk_dev = self.upload(np.concatenate([m.ravel() for m in mats]))
        if lib().scrappie_hip_set_trunk_input(self._h, self._trk_dev, off.ctypes.data_as(C.POINTER(C.c_uint64)), len(mats)) != 0:
            raise RuntimeError("set_trunk_input: " + last_error())

    def debug_option(self, name, value):
        if lib().scrappie_hip_debug_option(self._h, name.encode(), int(value)) != 0:
            raise RuntimeError("debug_option: " + last_error())

    def debug_fetch(self, what, dtype=np.uint8):
        """A device buffer of the most recent transducer launch group (scrappie_hip_debug_fetch) as a flat array."""
        n = lib().scrappie_hip_debug_fetch(self._h, what.encode(), None, 0)
        if n < 0:
            raise RuntimeError("debug_fetch: " + last_error())
        buf = np.zeros(n, dtype=np.uint8)
        if n and lib().scrappie_hip_debug_fetch(self._h, what.encode(), buf.ctypes.data, n) < 0:
            raise RuntimeError("debug_fetch: " + last_error())
        return buf.view(dtype)

    def debug_stitch(self, path, side=None, nstate=1025, crf=False):
        """k_stitch on one read given on the host: (bases or None, pos, redo)."""
        path = np.ascontiguousarray(path, dtype=np.int32)
        T = len(path) - 1
        sd = None if side is None else np.ascontiguousarray(side, dtype=ftype)
        buf = C.create_string_buffer(5 * (T + 1) + 16)
        pos = np.zeros(T + 1, np.int32)
        redo = C.c_int(0)
        n = lib().scrappie_hip_debug_stitch(self._h, path.ctypes.data_as(C.POINTER(C.c_int)),
                                            None if sd is None else sd.ctypes.data_as(C.POINTER(C.c_float)), T, nstate, 1 if crf else 0,
                                            buf, len(buf), pos.ctypes.data_as(C.POINTER(C.c_int)), C.byref(redo))
        if n == -2:
            raise RuntimeError("debug_stitch: " + last_error())
        return (buf.value.decode() if n >= 0 else None), pos, redo.value

    def posterior(self, signal, model='rgrgr_r94', min_prob=1e-5, tempW=1.0, tempb=1.0, log=True):
        """(T, NS) array, reference state order (stay last)."""
        rt = RawTable(signal)
        m = lib().scrappie_hip_posterior(self._h, self._models[model], rt.data(), min_prob, tempW, tempb, log)
        if not m:
            raise RuntimeError("posterior: " + last_error())
        return ScrappyMatrix(m).data(as_numpy=True, sloika=False)

    def trunk(self, signal, model='rgrgr_r94', upto=5):
        rt = RawTable(signal)
        m = lib().scrappie_hip_trunk(self._h, self._models[model], rt.data(), upto)
        if not m:
            raise RuntimeError("trunk: " + last_error())
        return ScrappyMatrix(m).data(as_numpy=True, sloika=False)
