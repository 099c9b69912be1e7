"""ctypes view of libagx.so (include/agx.h).  No compute happens in Python and there is no
fallback: if the library or a HIP device is missing, calls raise AgxError."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# The shipped library reads no environment variable.  The experiment knobs of tools/ exist in the tuning build
# only (libagx_tuning.so, -DAGX_TUNING): a process that sets one of them gets that library.
TUNING_KNOBS = ("AGX_SW_KERNEL", "AGX_SW_TAIL_BETA", "AGX_SW_MAX_C", "AGX_SW_FORCE_C", "AGX_SW_MAX_CLASSES", "AGX_SW_WAVES_PER_CLASS",
                "AGX_SW_SORT_WAVES", "AGX_SW_ONE_LAUNCH", "AGX_SW_DNA", "AGX_SW_RISE", "AGX_PHMM_PLAIN_CELL", "AGX_TRACE_CREATE", "AGX_HOST_THREADS", "AGX_FANOUT", "AGX_PHMM_TAB_BUDGET", "AGX_PHMM_MAX_C",
                "AGX_PHMM_FORCE_C", "AGX_PHMM_TAIL_BETA", "AGX_PHMM_MAX_CLASSES", "AGX_PHMM_NO_LUT", "AGX_TRACE_POOL", "AGX_NO_STREAM_PRIO", "AGX_SW_I32_CLASSIC", "AGX_SW_PIECE_MB", "AGX_SW_PIECE_MIN_PAIRS", "AGX_PHMM_NO_ROWS", "AGX_PHMM_NO_TRAINS", "AGX_PHMM_LUT_ONE_LOOP", "AGX_SW_HOST_PLAN", "AGX_PHMM_NO_RING")
_DEFAULT_LIB = "libagx_tuning.so" if any(k in os.environ for k in TUNING_KNOBS) else "libagx.so"
LIB_PATH = os.environ.get("AGX_LIB_PATH", os.path.join(_HERE, _DEFAULT_LIB))  # override: kernel experiments only

OK, E_ARG, E_NODEVICE, E_HIP, E_NOMEM, E_SYMBOL, E_LIMIT, E_IO = 0, -1, -2, -3, -4, -5, -6, -7
OPT_SW_KERNEL = 1
OPT_SW_PLANNER = 2
SW_PLANNER_AUTO, SW_PLANNER_HOST, SW_PLANNER_DEVICE = 0, 1, 2
OPT_PHMM_TRAINS = 3
PHMM_TRAINS_AUTO, PHMM_TRAINS_OFF, PHMM_TRAINS_ON = 0, 1, 2
SW_KERNEL_AUTO, SW_KERNEL_INT32, SW_KERNEL_PACKED_SIGNED, SW_KERNEL_PACKED_BIASED = 0, 1, 2, 3
PHMM_F64, PHMM_F64_FMA, PHMM_F32, PHMM_F32_FMA = 0, 1, 2, 3
PHMM_GATK_PRIOR = 0x100  # OR-able into the precision

# every symbol include/agx.h declares (tests check the library exports all of them)
SYMBOLS = [
    "agx_version", "agx_last_error", "agx_device_count", "agx_device_name", "agx_ctx_create", "agx_ctx_destroy", "agx_ctx_device",
    "agx_ctx_stream", "agx_ctx_set_stream", "agx_ctx_sync", "agx_ctx_set_option", "agx_warmup_devices", "agx_host_alloc", "agx_host_free",
    "agx_ctx_timer_start", "agx_ctx_timer_stop", "agx_ctx_timer_mark", "agx_ctx_timer_elapsed",
    "agx_sw_batch_create", "agx_sw_batch_create_scored", "agx_sw_batch_create_matrix", "agx_sw_batch_launch", "agx_sw_batch_scores", "agx_sw_batch_bind_scores", "agx_sw_batch_info", "agx_sw_batch_destroy",
    "agx_sw_score", "agx_sw_score_multi", "agx_sw_score_devices", "agx_sw_shard_cuts",
    "agx_phmm_batch_create", "agx_phmm_batch_launch", "agx_phmm_batch_results", "agx_phmm_batch_bind_results", "agx_phmm_batch_info",
    "agx_phmm_batch_destroy", "agx_phmm_forward", "agx_phmm_forward_multi", "agx_phmm_forward_devices", "agx_phmm_shard_cuts",
    "agx_pairHMM",
    "agx_sw_text_read", "agx_sw_text_free", "agx_sw_reader_open", "agx_sw_reader_line_num", "agx_sw_reader_set_threads", "agx_sw_reader_next",
    "agx_sw_reader_done", "agx_sw_reader_close", "agx_phmm_text_read", "agx_phmm_text_free",
    "agx_phmm_reader_open", "agx_phmm_reader_next", "agx_phmm_reader_done", "agx_phmm_reader_close",
]


class AgxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("agx error %d: %s" % (code, msg))
        self.code = code


class SwInfo(C.Structure):
    _fields_ = [("n_pairs", C.c_int64), ("cells", C.c_int64), ("padded_cells", C.c_int64), ("input_bytes", C.c_int64),
                ("n_launches", C.c_int32), ("n_waves", C.c_int32), ("planned_on_device", C.c_int32), ("reserved", C.c_int32)]


class SwScoring(C.Structure):
    _fields_ = [("match", C.c_int32), ("mismatch", C.c_int32), ("gap_open", C.c_int32), ("gap_extend", C.c_int32)]


class SwMatrix(C.Structure):
    """agx_sw_matrix: substitution matrix over up to 32 symbols (include/agx.h)."""
    _fields_ = [("n_symbols", C.c_int32), ("gap_open", C.c_int32), ("gap_extend", C.c_int32), ("code", C.c_uint8 * 256),
                ("score", (C.c_int8 * 32) * 32)]

    @classmethod
    def build(cls, alphabet: bytes, scores, gap_open: int, gap_extend: int, case_insensitive: bool = True):
        """alphabet: one byte per symbol; scores: n x n integers in alphabet order."""
        m = cls()
        m.n_symbols, m.gap_open, m.gap_extend = len(alphabet), gap_open, gap_extend
        for k in range(256):
            m.code[k] = 0xff
        for i, ch in enumerate(alphabet):
            m.code[ch] = i
            if case_insensitive and bytes([ch]).isalpha():
                m.code[ord(bytes([ch]).swapcase())] = i
        for a in range(len(alphabet)):
            for c in range(len(alphabet)):
                m.score[a][c] = int(scores[a][c])
        return m


class PhmmDesc(C.Structure):
    _fields_ = [("read_bases", C.c_void_p), ("q_base", C.c_void_p), ("q_ins", C.c_void_p), ("q_del", C.c_void_p),
                ("q_gcp", C.c_void_p), ("read_off", C.c_void_p), ("n_reads", C.c_uint32),
                ("hap_bases", C.c_void_p), ("hap_off", C.c_void_p), ("n_haps", C.c_uint32),
                ("region_read", C.c_void_p), ("region_hap", C.c_void_p), ("n_regions", C.c_uint32)]


class PhmmInfo(C.Structure):
    _fields_ = [("n_pairs", C.c_int64), ("cells", C.c_int64), ("padded_cells", C.c_int64), ("input_bytes", C.c_int64),
                ("n_launches", C.c_int32), ("n_waves", C.c_int32), ("n_rescued", C.c_int64)]


class SwText(C.Structure):
    _fields_ = [("line_num", C.c_int32), ("n_pairs", C.c_int64), ("bases", C.c_void_p), ("off", C.c_void_p),
                ("len", C.c_void_p), ("dangling", C.c_char_p)]


class PhmmText(C.Structure):
    _fields_ = [("desc", PhmmDesc), ("n_pairs", C.c_int64), ("n_regions_seen", C.c_int32), ("truncated", C.c_int32)]


_lib = None


def build(verbose: bool = False) -> None:
    """Compile libagx.so and the drop-in command lines for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.run(["make", "-C", _HERE, "-j8", "all"], check=True,
                   stdout=None if verbose else subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AgxError(E_NODEVICE, "libagx.so is not built (run __graft_entry__.build()); there is no fallback")
        l = C.CDLL(LIB_PATH)
        l.agx_version.restype = C.c_char_p
        l.agx_last_error.restype = C.c_char_p
        l.agx_ctx_stream.restype = C.c_void_p
        l.agx_ctx_stream.argtypes = [C.c_void_p]
        l.agx_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        l.agx_ctx_destroy.argtypes = [C.c_void_p]
        l.agx_ctx_destroy.restype = None
        l.agx_ctx_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        l.agx_ctx_sync.argtypes = [C.c_void_p]
        l.agx_ctx_set_option.argtypes = [C.c_void_p, C.c_int, C.c_int64]
        l.agx_warmup_devices.argtypes = [C.c_void_p, C.c_int]
        l.agx_host_alloc.argtypes = [C.c_size_t]
        l.agx_host_alloc.restype = C.c_void_p
        l.agx_host_free.argtypes = [C.c_void_p]
        l.agx_host_free.restype = None
        l.agx_ctx_timer_start.argtypes = [C.c_void_p]
        l.agx_ctx_timer_stop.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        l.agx_ctx_timer_mark.argtypes = [C.c_void_p]
        l.agx_ctx_timer_elapsed.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        l.agx_sw_batch_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                          C.POINTER(C.c_void_p)]
        l.agx_sw_batch_create_scored.argtypes = [C.c_void_p, C.POINTER(SwScoring), C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_int64, C.POINTER(C.c_void_p)]
        l.agx_sw_batch_create_matrix.argtypes = [C.c_void_p, C.POINTER(SwMatrix), C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_int64, C.POINTER(C.c_void_p)]
        l.agx_sw_batch_launch.argtypes = [C.c_void_p]
        l.agx_sw_batch_scores.argtypes = [C.c_void_p, C.c_void_p]
        l.agx_sw_batch_bind_scores.argtypes = [C.c_void_p, C.c_void_p]
        l.agx_sw_batch_info.argtypes = [C.c_void_p, C.POINTER(SwInfo)]
        l.agx_sw_batch_destroy.argtypes = [C.c_void_p]
        l.agx_sw_batch_destroy.restype = None
        l.agx_sw_score.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        l.agx_sw_score_multi.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        l.agx_sw_score_devices.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        l.agx_sw_shard_cuts.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        l.agx_phmm_forward_devices.argtypes = [C.c_void_p, C.c_int, C.POINTER(PhmmDesc), C.c_int, C.c_void_p]
        l.agx_phmm_shard_cuts.argtypes = [C.POINTER(PhmmDesc), C.c_int, C.c_void_p]
        l.agx_phmm_batch_create.argtypes = [C.c_void_p, C.POINTER(PhmmDesc), C.c_int, C.POINTER(C.c_void_p)]
        l.agx_phmm_batch_launch.argtypes = [C.c_void_p]
        l.agx_phmm_batch_results.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        l.agx_phmm_batch_info.argtypes = [C.c_void_p, C.POINTER(PhmmInfo)]
        l.agx_phmm_batch_bind_results.argtypes = [C.c_void_p, C.c_void_p]
        l.agx_phmm_batch_destroy.argtypes = [C.c_void_p]
        l.agx_phmm_batch_destroy.restype = None
        l.agx_phmm_forward.argtypes = [C.c_void_p, C.POINTER(PhmmDesc), C.c_int, C.c_void_p]
        l.agx_phmm_forward_multi.argtypes = [C.c_int, C.POINTER(PhmmDesc), C.c_int, C.c_void_p]
        l.agx_pairHMM.argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_int] + [C.c_void_p] * 4
        l.agx_pairHMM.restype = None
        l.agx_sw_text_read.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.POINTER(SwText))]
        l.agx_sw_text_free.argtypes = [C.POINTER(SwText)]
        l.agx_sw_text_free.restype = None
        l.agx_sw_reader_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
        l.agx_sw_reader_line_num.argtypes = [C.c_void_p]
        l.agx_sw_reader_line_num.restype = C.c_int32
        l.agx_sw_reader_next.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.POINTER(SwText))]
        l.agx_sw_reader_done.argtypes = [C.c_void_p]
        l.agx_sw_reader_close.argtypes = [C.c_void_p]
        l.agx_sw_reader_close.restype = None
        l.agx_phmm_text_read.argtypes = [C.c_char_p, C.POINTER(C.POINTER(PhmmText))]
        l.agx_phmm_text_free.argtypes = [C.POINTER(PhmmText)]
        l.agx_phmm_text_free.restype = None
        l.agx_phmm_reader_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        l.agx_phmm_reader_next.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.POINTER(PhmmText))]
        l.agx_phmm_reader_done.argtypes = [C.c_void_p]
        l.agx_phmm_reader_close.argtypes = [C.c_void_p]
        l.agx_phmm_reader_close.restype = None
        _lib = l
    return _lib


def _check(rc):
    if rc != 0:
        raise AgxError(rc, lib().agx_last_error().decode(errors="replace"))


def device_count() -> int:
    return lib().agx_device_count()


def _ptr(a):
    return a.ctypes.data if a is not None and a.size else None


class Context:
    """agx_ctx: one device, one HIP stream."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        _check(lib().agx_ctx_create(device, C.byref(self._h)))
        self.device = device

    def close(self):
        if self._h:
            lib().agx_ctx_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def stream(self) -> int:
        return lib().agx_ctx_stream(self._h) or 0

    def set_stream(self, hip_stream: int):
        _check(lib().agx_ctx_set_stream(self._h, C.c_void_p(hip_stream)))

    def sync(self):
        _check(lib().agx_ctx_sync(self._h))

    def set_option(self, key: int, value: int):
        _check(lib().agx_ctx_set_option(self._h, key, value))

    def timer_start(self):
        _check(lib().agx_ctx_timer_start(self._h))

    def timer_stop(self) -> float:
        ms = C.c_float()
        _check(lib().agx_ctx_timer_stop(self._h, C.byref(ms)))
        return ms.value

    def timer_mark(self):
        _check(lib().agx_ctx_timer_mark(self._h))

    def timer_elapsed(self) -> float:
        ms = C.c_float()
        _check(lib().agx_ctx_timer_elapsed(self._h, C.byref(ms)))
        return ms.value

    # ---- Smith-Waterman
    def sw_batch(self, b, scoring=None, matrix=None) -> "SwBatch":
        return SwBatch(self, b, scoring, matrix)

    def sw_score(self, b) -> np.ndarray:
        """b: synth.SWBatch (bases/off/len) -> int32 scores, one-shot."""
        out = np.empty(b.n_pairs, np.int32)
        _check(lib().agx_sw_score(self._h, _ptr(b.bases), _ptr(b.off), _ptr(b.len), b.n_pairs, _ptr(out)))
        return out

    # ---- PairHMM
    def phmm_batch(self, b, precision=PHMM_F64) -> "PhmmBatchDev":
        return PhmmBatchDev(self, b, precision)

    def phmm_forward(self, b, precision=PHMM_F64) -> np.ndarray:
        out = np.empty(b.n_pairs, np.float64)
        d, _keep = phmm_desc(b)
        _check(lib().agx_phmm_forward(self._h, C.byref(d), precision, _ptr(out)))
        return out


class SwBatch:
    """agx_sw_batch: a scheduled batch resident in HBM (ctx=None: planned on the host only)."""

    def __init__(self, ctx, b, scoring=None, matrix=None):
        """scoring: None (the reference's +1/-1/-3/-1) or (match, mismatch, gap_open, gap_extend);
        matrix: an SwMatrix instead."""
        self.ctx = ctx
        self.n_pairs = b.n_pairs
        self._h = C.c_void_p()
        if matrix is not None:
            _check(lib().agx_sw_batch_create_matrix(ctx._h if ctx else None, C.byref(matrix), _ptr(b.bases), _ptr(b.off),
                                                    _ptr(b.len), b.n_pairs, C.byref(self._h)))
            return
        sc = C.byref(SwScoring(*scoring)) if scoring is not None else None
        _check(lib().agx_sw_batch_create_scored(ctx._h if ctx else None, sc, _ptr(b.bases), _ptr(b.off), _ptr(b.len),
                                                b.n_pairs, C.byref(self._h)))

    def launch(self):
        _check(lib().agx_sw_batch_launch(self._h))

    def scores(self, out=None) -> np.ndarray:
        if out is None:
            out = np.empty(self.n_pairs, np.int32)
        _check(lib().agx_sw_batch_scores(self._h, _ptr(out)))
        return out

    def bind_scores(self, out):
        """agx_sw_batch_bind_scores: out = a page-locked int32 array (host_array) or None."""
        _check(lib().agx_sw_batch_bind_scores(self._h, _ptr(out) if out is not None else None))

    def info(self) -> SwInfo:
        i = SwInfo()
        _check(lib().agx_sw_batch_info(self._h, C.byref(i)))
        return i

    def close(self):
        if self._h:
            lib().agx_sw_batch_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close


def phmm_desc(b):
    """synth.PhmmBatch -> (PhmmDesc, keepalive)."""
    d = PhmmDesc()
    d.read_bases, d.q_base, d.q_ins, d.q_del, d.q_gcp = (_ptr(x) for x in (b.read_bases, b.q_base, b.q_ins, b.q_del, b.q_gcp))
    d.read_off = _ptr(b.roff)
    d.n_reads = b.roff.size - 1
    d.hap_bases = _ptr(b.hap_bases)
    d.hap_off = _ptr(b.hoff)
    d.n_haps = b.hoff.size - 1
    d.region_read = _ptr(b.rreg)
    d.region_hap = _ptr(b.hreg)
    d.n_regions = b.n_regions
    return d, b


class PhmmBatchDev:
    """agx_phmm_batch: a scheduled PairHMM batch resident in HBM."""

    def __init__(self, ctx, b, precision=PHMM_F64):
        self.ctx = ctx
        self.n_pairs = b.n_pairs
        self._h = C.c_void_p()
        d, self._keep = phmm_desc(b)
        _check(lib().agx_phmm_batch_create(ctx._h if ctx else None, C.byref(d), precision, C.byref(self._h)))

    def launch(self):
        _check(lib().agx_phmm_batch_launch(self._h))

    def results(self, out=None, want_sums=True):
        """-> (log10 likelihoods, raw sums), float64; out = (l, s) arrays to reuse; want_sums=False fetches the
        likelihoods only (raw_sum = NULL in the C call) and returns (l, None)."""
        l, s = out if out is not None else (np.empty(self.n_pairs, np.float64), np.empty(self.n_pairs, np.float64) if want_sums else None)
        if not want_sums:
            s = None
        _check(lib().agx_phmm_batch_results(self._h, _ptr(l), _ptr(s) if s is not None else None))
        return l, s

    def bind_results(self, out):
        """agx_phmm_batch_bind_results: out = a page-locked float64 array (host_array) or None."""
        _check(lib().agx_phmm_batch_bind_results(self._h, _ptr(out) if out is not None else None))

    def info(self) -> PhmmInfo:
        i = PhmmInfo()
        _check(lib().agx_phmm_batch_info(self._h, C.byref(i)))
        return i

    def close(self):
        if self._h:
            lib().agx_phmm_batch_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close


def sw_score_multi(b, n_devices: int = 0) -> np.ndarray:
    out = np.empty(b.n_pairs, np.int32)
    _check(lib().agx_sw_score_multi(n_devices, _ptr(b.bases), _ptr(b.off), _ptr(b.len), b.n_pairs, _ptr(out)))
    return out


def sw_score_devices(b, devices) -> np.ndarray:
    """agx_sw_score_devices: shard k of the batch runs on devices[k] (an ordinal may repeat)."""
    out = np.empty(b.n_pairs, np.int32)
    dv = np.asarray(devices, np.int32)
    _check(lib().agx_sw_score_devices(_ptr(dv), dv.size, _ptr(b.bases), _ptr(b.off), _ptr(b.len), b.n_pairs, _ptr(out)))
    return out


def sw_shard_cuts(b, n_shards: int) -> np.ndarray:
    cut = np.zeros(n_shards + 1, np.int64)
    _check(lib().agx_sw_shard_cuts(_ptr(b.len), b.n_pairs, n_shards, _ptr(cut)))
    return cut


def phmm_forward_multi(b, precision=PHMM_F64, n_devices: int = 0) -> np.ndarray:
    out = np.empty(b.n_pairs, np.float64)
    d, _keep = phmm_desc(b)
    _check(lib().agx_phmm_forward_multi(n_devices, C.byref(d), precision, _ptr(out)))
    return out


def phmm_forward_devices(b, devices, precision=PHMM_F64) -> np.ndarray:
    out = np.empty(b.n_pairs, np.float64)
    d, _keep = phmm_desc(b)
    dv = np.asarray(devices, np.int32)
    _check(lib().agx_phmm_forward_devices(_ptr(dv), dv.size, C.byref(d), precision, _ptr(out)))
    return out


def phmm_shard_cuts(b, n_shards: int) -> np.ndarray:
    cut = np.zeros(n_shards + 1, np.uint32)
    d, _keep = phmm_desc(b)
    _check(lib().agx_phmm_shard_cuts(C.byref(d), n_shards, _ptr(cut)))
    return cut


def host_array(n: int, dtype) -> np.ndarray:
    """A numpy array in page-locked memory (agx_host_alloc); never freed explicitly: the few the bench and the
    tests make live as long as the process."""
    dt = np.dtype(dtype)
    p = lib().agx_host_alloc(max(1, n) * dt.itemsize)
    if not p:
        raise AgxError(E_NOMEM, lib().agx_last_error().decode(errors="replace"))
    buf = (C.c_uint8 * (max(1, n) * dt.itemsize)).from_address(p)
    return np.frombuffer(buf, dtype=dt, count=n)


def _sw_text_to_batch(t):
    from . import synth

    if True:
        n = t.contents.n_pairs
        off = np.ctypeslib.as_array(C.cast(t.contents.off, C.POINTER(C.c_uint64)), shape=(2 * n,)).copy() if n else np.zeros(0, np.uint64)
        ln = np.ctypeslib.as_array(C.cast(t.contents.len, C.POINTER(C.c_uint32)), shape=(2 * n,)).copy() if n else np.zeros(0, np.uint32)
        total = int(off[-1] + ln[-1]) if n else 0
        bases = np.ctypeslib.as_array(C.cast(t.contents.bases, C.POINTER(C.c_uint8)), shape=(total,)).copy() if total else np.zeros(0, np.uint8)
        return t.contents.line_num, synth.SWBatch(bases, off, ln), t.contents.dangling


def read_sw_text(path: str, line_buf: int = 0):
    """agx_sw_text_read -> (line_num, synth.SWBatch, dangling line or None)."""
    t = C.POINTER(SwText)()
    _check(lib().agx_sw_text_read(path.encode(), line_buf, C.byref(t)))
    try:
        return _sw_text_to_batch(t)
    finally:
        lib().agx_sw_text_free(t)


def read_sw_text_chunks(path: str, max_pairs: int, line_buf: int = 0):
    """agx_sw_reader_*: yields (line_num, synth.SWBatch, dangling) per chunk of up to max_pairs pairs."""
    r = C.c_void_p()
    _check(lib().agx_sw_reader_open(path.encode(), line_buf, C.byref(r)))
    try:
        while not lib().agx_sw_reader_done(r):
            t = C.POINTER(SwText)()
            _check(lib().agx_sw_reader_next(r, max_pairs, C.byref(t)))
            try:
                yield _sw_text_to_batch(t)
            finally:
                lib().agx_sw_text_free(t)
    finally:
        lib().agx_sw_reader_close(r)


def _phmm_text_to_batch(t):
    from . import synth

    if True:
        d = t.contents.desc
        arr = lambda p, ty, n: (np.ctypeslib.as_array(C.cast(p, C.POINTER(ty)), shape=(n,)).copy() if n else np.zeros(0, ty))
        roff = arr(d.read_off, C.c_uint64, d.n_reads + 1)
        hoff = arr(d.hap_off, C.c_uint64, d.n_haps + 1)
        nb, nh = int(roff[-1]), int(hoff[-1])
        b = synth.PhmmBatch(arr(d.read_bases, C.c_uint8, nb), arr(d.q_base, C.c_uint8, nb), arr(d.q_ins, C.c_uint8, nb),
                            arr(d.q_del, C.c_uint8, nb), arr(d.q_gcp, C.c_uint8, nb), roff,
                            arr(d.hap_bases, C.c_uint8, nh), hoff, arr(d.region_read, C.c_uint32, d.n_regions + 1),
                            arr(d.region_hap, C.c_uint32, d.n_regions + 1))
        return b, t.contents.n_regions_seen, t.contents.truncated


def read_phmm_text(path: str):
    """agx_phmm_text_read -> (synth.PhmmBatch, n_regions_seen, truncated)."""
    t = C.POINTER(PhmmText)()
    _check(lib().agx_phmm_text_read(path.encode(), C.byref(t)))
    try:
        return _phmm_text_to_batch(t)
    finally:
        lib().agx_phmm_text_free(t)


def read_phmm_text_chunks(path: str, max_pairs: int):
    """agx_phmm_reader_*: yields (synth.PhmmBatch, n_regions_seen, truncated) per chunk of whole regions."""
    r = C.c_void_p()
    _check(lib().agx_phmm_reader_open(path.encode(), C.byref(r)))
    try:
        while not lib().agx_phmm_reader_done(r):
            t = C.POINTER(PhmmText)()
            _check(lib().agx_phmm_reader_next(r, max_pairs, C.byref(t)))
            try:
                yield _phmm_text_to_batch(t)
            finally:
                lib().agx_phmm_text_free(t)
    finally:
        lib().agx_phmm_reader_close(r)
