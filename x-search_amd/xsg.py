"""ctypes binding of the C ABI in include/xsg.h (libxsg.so).

Plain pointers and sizes only: device memory is passed as integer addresses
(e.g. torch.Tensor.data_ptr()), streams as integer hipStream_t handles.  This
module never falls back to a CPU implementation: if the library or the GPU is
missing, the call raises.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent

OK = 0
EINVAL, ENODEV, EHIP, ENOMEM, ENOTSUP, EIO, ESTATE = -1, -2, -3, -4, -5, -6, -7

COUNT_MATCHES, COUNT_LINES, MATCH_BYTE_OFFSETS, LINE_BYTE_OFFSETS, LINE_INDICES, LINES = range(6)
FLAG_EXACT_TAIL = 0x1
FLAG_IGNORE_CASE = 0x2
FLAG_REGEX = 0x4
WITH_NEWLINES = 0x100
CTR_MATCHES, CTR_LINES, CTR_NEWLINES, CTR_BYTES = range(4)
NUM_COUNTERS = 4
LINE_BASE_AUTO = (1 << 64) - 1
MAX_PATTERN = 32768
MAX_REGEX = 1024
STATUS_OK, STATUS_OVERFLOW, STATUS_NONASCII = 0, 1, 2
TILE = 16384

CHUNK_DTYPE = np.dtype([("offset", "<u8"), ("length", "<u8"), ("global_offset", "<u8"), ("line_base", "<u8")])

_u64p = C.POINTER(C.c_uint64)

FILE_CHUNK_DTYPE = np.dtype([("original_offset", "<u8"), ("actual_offset", "<u8"), ("original_size", "<u8"),
                             ("actual_size", "<u8"), ("first_line", "<u8"), ("n_mappings", "<u8")])
COMPRESSION_NONE, COMPRESSION_ZSTD, COMPRESSION_LZ4 = 1, 2, 3


class JobOpts(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("mode", C.c_uint32), ("pattern_flags", C.c_uint32),
                ("device", C.c_int32), ("num_threads", C.c_int32), ("num_max_readers", C.c_int32),
                ("chunk_bytes", C.c_uint64), ("chunk_begin", C.c_uint64), ("chunk_end", C.c_uint64)]


class JobStats(C.Structure):
    _fields_ = [("bytes_scanned", C.c_uint64), ("bytes_read", C.c_uint64), ("chunks", C.c_uint64),
                ("seconds_total", C.c_double), ("seconds_read", C.c_double), ("seconds_decompress", C.c_double),
                ("seconds_device", C.c_double), ("newlines", C.c_uint64), ("plan_chunks", C.c_uint64)]


class RegexDfaInfo(C.Structure):
    _fields_ = [("ncls", C.c_uint32), ("minlen", C.c_uint32), ("ascii_only", C.c_uint32), ("multiline", C.c_uint32),
                ("prefix_positions", C.c_uint32), ("prefix_alternatives", C.c_uint32), ("factor_positions", C.c_uint32),
                ("fwd_states", C.c_uint32), ("fwd_start", C.c_uint32), ("fwd_first_acc", C.c_uint32),
                ("rev_states", C.c_uint32), ("rev_start", C.c_uint32), ("rev_first_acc", C.c_uint32),
                ("class_of", C.c_uint8 * 256)]


class XsgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"xsg error {code}: {msg}")
        self.code = code


def lib_path() -> Path:
    env = os.environ.get("XSG_LIB")
    return Path(env) if env else HERE / "lib" / "libxsg.so"


_lib = None


def load():
    """Load libxsg.so (raises if it has not been built: there is no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 /
    # libhsa-runtime64 and a second HSA runtime in the same process sees no GPU.
    # libxsg.so only NEEDs the SONAME libamdhip64.so.7, so when torch is loaded
    # first the dynamic loader binds libxsg to torch's copy and both share it.
    # (A process without torch simply gets /opt/rocm's runtime.)
    if "torch" not in sys.modules and not os.environ.get("XSG_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    p = lib_path()
    if not p.exists() and not os.environ.get("XSG_LIB"):
        # a fresh checkout: compile the HIP library in-tree (hipcc cross-compiles gfx950 anywhere; ~1 min).
        # This builds the real extension -- there is still no CPU fallback behind it.
        import subprocess
        print(f"[xsg] {p} missing: running make -C {HERE}", file=sys.stderr, flush=True)
        subprocess.run(["make", "-C", str(HERE), "--no-print-directory"], check=False)
    if not p.exists():
        raise FileNotFoundError(f"{p} not found: build it with `make -C x-search_amd` (or __graft_entry__.build())")
    lib = C.CDLL(str(p))
    vp, u64, u32, ci, sz = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_size_t
    sig = {
        "xsg_abi_version": (ci, []),
        "xsg_strerror": (C.c_char_p, [ci]),
        "xsg_last_error": (C.c_char_p, []),
        "xsg_device_count": (ci, [C.POINTER(ci)]),
        "xsg_ctx_create": (ci, [ci, C.POINTER(vp)]),
        "xsg_ctx_destroy": (None, [vp]),
        "xsg_set_pattern": (ci, [vp, C.c_char_p, sz, u32]),
        "xsg_regex_check": (ci, [C.c_char_p, sz, u32, C.POINTER(u32), C.POINTER(u32)]),
        "xsg_regex_info": (ci, [C.c_char_p, sz, u32, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]),
        "xsg_regex_factor": (ci, [C.c_char_p, sz, u32, C.POINTER(u32), C.POINTER(u32)]),
        "xsg_regex_prefix": (ci, [C.c_char_p, sz, u32, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]),
        "xsg_regex_dfa_info": (ci, [C.c_char_p, sz, u32, C.POINTER(RegexDfaInfo), C.POINTER(C.c_uint16), C.POINTER(C.c_uint16), sz]),
        "xsg_shard_create": (ci, [vp, vp, u64, vp, u64, C.POINTER(vp)]),
        "xsg_shard_rebind": (ci, [vp, vp, u64, vp, u64]),
        "xsg_shard_invalidate": (ci, [vp]),
        "xsg_shard_destroy": (None, [vp]),
        "xsg_shard_set_line_base": (ci, [vp, u64]),
        "xsg_count_async": (ci, [vp, u32, vp, vp]),
        "xsg_count_async_status": (ci, [vp, u32, vp, vp, vp]),
        "xsg_count": (ci, [vp, u32, _u64p]),
        "xsg_count_begin": (ci, [vp, u32]),
        "xsg_count_end": (ci, [vp, _u64p]),
        "xsg_search": (ci, [vp, u32, _u64p]),
        "xsg_result_u64": (ci, [vp, _u64p, u64]),
        "xsg_result_u64_view": (ci, [vp, C.POINTER(_u64p), _u64p]),
        "xsg_result_lines_size": (ci, [vp, _u64p, _u64p]),
        "xsg_result_lines": (ci, [vp, _u64p, vp, u64, _u64p]),
        "xsg_result_lines_view": (ci, [vp, C.POINTER(_u64p), C.POINTER(C.c_char_p), C.POINTER(_u64p), _u64p, _u64p]),
        "xsg_result_newlines": (ci, [vp, _u64p]),
        "xsg_job_opts_init": (None, [C.POINTER(JobOpts)]),
        "xsg_job_start": (ci, [C.c_char_p, sz, C.c_char_p, C.c_char_p, C.POINTER(JobOpts), C.POINTER(vp)]),
        "xsg_job_join": (ci, [vp]),
        "xsg_job_destroy": (None, [vp]),
        "xsg_job_total": (ci, [vp, _u64p]),
        "xsg_job_wait": (ci, [vp, u64, _u64p, C.POINTER(ci)]),
        "xsg_job_poll": (ci, [vp, _u64p, C.POINTER(ci)]),
        "xsg_job_get_u64": (ci, [vp, u64, u64, _u64p]),
        "xsg_job_get_line": (ci, [vp, u64, C.POINTER(C.c_char_p), _u64p]),
        "xsg_job_stats_get": (ci, [vp, C.POINTER(JobStats)]),
        "xsg_host_searcher_create": (ci, [ci, C.c_char_p, sz, u32, ci, C.POINTER(vp)]),
        "xsg_host_searcher_destroy": (None, [vp]),
        "xsg_host_count": (ci, [vp, vp, u64, ci, _u64p]),
        "xsg_host_offsets": (ci, [vp, u32, vp, u64, C.POINTER(vp), _u64p]),
        "xsg_host_lines": (ci, [vp, vp, u64, C.POINTER(vp), C.POINTER(vp), _u64p, _u64p]),
        "xsg_plan_chunks": (ci, [C.c_char_p, u64, C.POINTER(vp), _u64p]),
        "xsg_meta_read": (ci, [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(vp), _u64p, C.POINTER(vp), _u64p]),
        "xsg_meta_write": (ci, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int32, u64, u64, ci]),
        "xsg_free": (None, [vp]),
        "xsg_codec_name": (C.c_char_p, [C.c_int32]),
        "xsg_ctx_info": (ci, [vp, C.c_char_p, sz, C.POINTER(ci), _u64p]),
        "xsg_time_scan_kernel": (ci, [vp, u32, ci, C.POINTER(C.c_float)]),
        "xsg_comm_unique_id": (ci, [vp, sz]),
        "xsg_comm_create_rank": (ci, [vp, ci, ci, vp, C.POINTER(vp)]),
        "xsg_comm_create_local": (ci, [C.POINTER(vp), ci, C.POINTER(vp)]),
        "xsg_comm_destroy": (None, [vp]),
        "xsg_comm_size": (ci, [vp, C.POINTER(ci), C.POINTER(ci)]),
        "xsg_comm_library": (C.c_char_p, []),
        "xsg_reduce_counts_async": (ci, [vp, vp, ci, vp]),
        "xsg_reduce_counts": (ci, [vp, C.POINTER(vp), ci, _u64p]),
        "xsg_allgather_u64": (ci, [vp, _u64p, _u64p]),
        "xsg_device_numa": (ci, [ci, C.POINTER(ci), C.c_char_p, sz]),
        "xsg_jobs_reduce_total": (ci, [C.POINTER(vp), ci, _u64p, C.POINTER(ci)]),
        "xsg_scan_kernel_name": (ci, [vp, u32, C.c_char_p, sz]),
        "xsg_shard_tune": (ci, [vp, u32, C.POINTER(u32)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


EXPORTS = ["xsg_abi_version", "xsg_strerror", "xsg_last_error", "xsg_device_count", "xsg_ctx_create",
           "xsg_ctx_destroy", "xsg_set_pattern", "xsg_regex_check", "xsg_shard_create", "xsg_shard_rebind", "xsg_shard_destroy",
           "xsg_shard_set_line_base", "xsg_count_async", "xsg_count", "xsg_search", "xsg_result_u64",
           "xsg_result_lines_size", "xsg_result_lines", "xsg_ctx_info", "xsg_time_scan_kernel", "xsg_result_newlines",
           "xsg_job_opts_init", "xsg_job_start", "xsg_job_join", "xsg_job_destroy", "xsg_job_total", "xsg_job_wait", "xsg_job_poll",
           "xsg_job_get_u64", "xsg_job_get_line", "xsg_job_stats_get", "xsg_plan_chunks", "xsg_meta_read",
           "xsg_meta_write", "xsg_free", "xsg_host_searcher_create", "xsg_host_searcher_destroy", "xsg_host_count",
           "xsg_host_offsets", "xsg_host_lines", "xsg_scan_kernel_name", "xsg_shard_tune", "xsg_count_begin",
           "xsg_count_end", "xsg_comm_unique_id", "xsg_comm_create_rank", "xsg_comm_create_local", "xsg_comm_destroy",
           "xsg_comm_size", "xsg_comm_library", "xsg_reduce_counts_async", "xsg_reduce_counts", "xsg_allgather_u64",
           "xsg_jobs_reduce_total", "xsg_device_numa", "xsg_regex_info", "xsg_regex_dfa_info", "xsg_regex_prefix", "xsg_regex_factor",
           "xsg_result_u64_view", "xsg_shard_invalidate", "xsg_result_lines_view", "xsg_count_async_status", "xsg_codec_name"]


def _check(rc):
    if rc != OK:
        raise XsgError(rc, load().xsg_last_error().decode("utf-8", "replace"))


def regex_check(expr: bytes, flags: int = 0):
    """-> (positions, sets[positions, 8] uint32) if XSG_FLAG_REGEX serves `expr`; raises XsgError otherwise.  No GPU needed."""
    lib = load()
    n = C.c_uint32(0)
    sets = np.zeros((32, 8), dtype=np.uint32)
    _check(lib.xsg_regex_check(expr, len(expr), flags, C.byref(n), sets.ctypes.data_as(C.POINTER(C.c_uint32))))
    return int(n.value), sets[:n.value].copy()


def regex_info(expr: bytes, flags: int = 0):
    """-> (positions, alternatives, ascii_only, sets[alternatives, positions, 8] uint32); raises XsgError if refused"""
    lib = load()
    n, na, ao = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    sets = np.zeros((64, 8), dtype=np.uint32)
    _check(lib.xsg_regex_info(expr, len(expr), flags, C.byref(n), C.byref(na), C.byref(ao),
                              sets.ctypes.data_as(C.POINTER(C.c_uint32))))
    return int(n.value), int(na.value), bool(ao.value), sets[:n.value * na.value].reshape(na.value, n.value, 8).copy()


def regex_dfa(expr: bytes, flags: int = 0):
    """The automata of a variable-length expression (include/xsg.h: xsg_regex_dfa_info) -> (info, fwd, rev): the
    tables as uint16 arrays [states, ncls] of pre-multiplied row offsets.  Raises XsgError if the route refuses."""
    lib = load()
    info = RegexDfaInfo()
    cap = 16384
    fwd = np.zeros(cap, dtype=np.uint16)
    rev = np.zeros(cap, dtype=np.uint16)
    _check(lib.xsg_regex_dfa_info(expr, len(expr), flags, C.byref(info), fwd.ctypes.data_as(C.POINTER(C.c_uint16)),
                                  rev.ctypes.data_as(C.POINTER(C.c_uint16)), cap))
    return (info, fwd[:info.fwd_states * info.ncls].reshape(info.fwd_states, info.ncls).copy(),
            rev[:info.rev_states * info.ncls].reshape(info.rev_states, info.ncls).copy())


def regex_prefix(expr: bytes, flags: int = 0):
    """-> (positions, sets[alternatives, positions, 8] uint32): the prefilter of a variable-length expression; positions == 0: none"""
    lib = load()
    n, na = C.c_uint32(0), C.c_uint32(0)
    sets = np.zeros((64, 8), dtype=np.uint32)
    _check(lib.xsg_regex_prefix(expr, len(expr), flags, C.byref(n), C.byref(na), sets.ctypes.data_as(C.POINTER(C.c_uint32))))
    return int(n.value), sets[:n.value * na.value].reshape(na.value, n.value, 8).copy()


def regex_factor(expr: bytes, flags: int = 0):
    """-> (positions, sets[positions, 8] uint32): a class sequence every match contains; positions == 0: none"""
    lib = load()
    n = C.c_uint32(0)
    sets = np.zeros((32, 8), dtype=np.uint32)
    _check(lib.xsg_regex_factor(expr, len(expr), flags, C.byref(n), sets.ctypes.data_as(C.POINTER(C.c_uint32))))
    return int(n.value), sets[:n.value].copy()


def device_count() -> int:
    n = C.c_int(0)
    _check(load().xsg_device_count(C.byref(n)))
    return n.value


def make_chunks(offsets, lengths, global_offsets=None, line_bases=None) -> np.ndarray:
    n = len(offsets)
    a = np.zeros(n, dtype=CHUNK_DTYPE)
    a["offset"] = offsets
    a["length"] = lengths
    if global_offsets is None:
        go = np.zeros(n, dtype=np.uint64)
        if n:
            go[1:] = np.cumsum(np.asarray(lengths, dtype=np.uint64))[:-1]
        a["global_offset"] = go
    else:
        a["global_offset"] = global_offsets
    a["line_base"] = LINE_BASE_AUTO if line_bases is None else line_bases
    return a


class Context:
    def __init__(self, device: int = 0):
        self._lib = load()
        h = C.c_void_p()
        _check(self._lib.xsg_ctx_create(device, C.byref(h)))
        self.h = h
        self.device = device
        # The C ABI wants shards and communicators destroyed before their context.  Python finalises the members
        # of a reference cycle in no particular order (e.g. everything a caught exception's traceback keeps
        # alive), so the context closes whatever is still open on it first; closing twice is harmless.
        self._children = {}

    def _adopt(self, child):
        self._children[id(child)] = child

    def _release(self, child):
        self._children.pop(id(child), None)

    def close(self):
        for child in list(self._children.values()):
            child.close()
        self._children.clear()
        if self.h:
            self._lib.xsg_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_pattern(self, pattern: bytes, flags: int = 0):
        _check(self._lib.xsg_set_pattern(self.h, pattern, len(pattern), flags))

    def info(self):
        arch = C.create_string_buffer(128)
        cus = C.c_int(0)
        hbm = C.c_uint64(0)
        _check(self._lib.xsg_ctx_info(self.h, arch, 128, C.byref(cus), C.byref(hbm)))
        return {"arch": arch.value.decode(), "compute_units": cus.value, "hbm_bytes": hbm.value}


class Shard:
    """Chunks already resident in HBM, bound for searching (xsg_shard_create).

    The bound bytes are IMMUTABLE while the binding lives, as the reference's chunk is a value its searchers only read
    (concepts.h:36-39).  The library remembers things it derived from them -- newline counts per tile, which hot filter
    won, and for a literal that can overlap itself (`aa`, `abab`, `that`) whether it DOES overlap in these bytes, which
    decides how it is counted.  Refill the buffer in place and you must call rebind() or invalidate() before the next
    search, or such a needle is counted with the old buffer's verdict.
    """

    def __init__(self, ctx: Context, d_base: int, capacity: int, chunks: np.ndarray):
        self._lib = ctx._lib
        self.ctx = ctx
        chunks = np.ascontiguousarray(chunks, dtype=CHUNK_DTYPE)
        h = C.c_void_p()
        _check(self._lib.xsg_shard_create(ctx.h, C.c_void_p(d_base), capacity, chunks.ctypes.data, len(chunks),
                                          C.byref(h)))
        self.h = h
        self.nchunks = len(chunks)
        ctx._adopt(self)

    def close(self):
        if self.h:
            self._lib.xsg_shard_destroy(self.h)
            self.h = None
            self.ctx._release(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def rebind(self, d_base: int, capacity: int, chunks: np.ndarray):
        chunks = np.ascontiguousarray(chunks, dtype=CHUNK_DTYPE)
        _check(self._lib.xsg_shard_rebind(self.h, C.c_void_p(d_base), capacity, chunks.ctypes.data, len(chunks)))
        self.nchunks = len(chunks)

    def invalidate(self):
        """the bytes behind the binding were rewritten in place: forget what was derived from them"""
        _check(self._lib.xsg_shard_invalidate(self.h))

    def set_line_base(self, base: int):
        _check(self._lib.xsg_shard_set_line_base(self.h, base))

    def count_async(self, mode: int, stream: int, d_counters: int):
        _check(self._lib.xsg_count_async(self.h, mode, C.c_void_p(stream), C.c_void_p(d_counters)))

    def count_async_status(self, mode: int, stream: int, d_counters: int, d_status: int):
        """the stream-ordered count with a status word (device uint64): 0, or STATUS_* bits and zeroed counters"""
        _check(self._lib.xsg_count_async_status(self.h, mode, C.c_void_p(stream), C.c_void_p(d_counters), C.c_void_p(d_status)))

    def count(self, mode: int) -> np.ndarray:
        out = np.zeros(NUM_COUNTERS, dtype=np.uint64)
        _check(self._lib.xsg_count(self.h, mode, out.ctypes.data_as(_u64p)))
        return out

    def count_begin(self, mode: int):
        _check(self._lib.xsg_count_begin(self.h, mode))

    def count_end(self) -> np.ndarray:
        out = np.zeros(NUM_COUNTERS, dtype=np.uint64)
        _check(self._lib.xsg_count_end(self.h, out.ctypes.data_as(_u64p)))
        return out

    def search_u64(self, mode: int) -> np.ndarray:
        n = C.c_uint64(0)
        _check(self._lib.xsg_search(self.h, mode, C.byref(n)))
        out = np.empty(n.value, dtype=np.uint64)
        _check(self._lib.xsg_result_u64(self.h, out.ctypes.data_as(_u64p), n.value))
        return out

    def search_u64_view(self, mode: int) -> np.ndarray:
        """like search_u64, but the array is a VIEW of the shard's pinned result buffer: valid until the next search"""
        n = C.c_uint64(0)
        _check(self._lib.xsg_search(self.h, mode, C.byref(n)))
        ptr, cnt = _u64p(), C.c_uint64(0)
        _check(self._lib.xsg_result_u64_view(self.h, C.byref(ptr), C.byref(cnt)))
        if cnt.value == 0:
            return np.zeros(0, dtype=np.uint64)
        return np.ctypeslib.as_array(ptr, shape=(cnt.value,))

    def search_lines(self):
        """-> (list of bytes, global byte offset of every line start)"""
        n = C.c_uint64(0)
        _check(self._lib.xsg_search(self.h, LINES, C.byref(n)))
        nl, nb = C.c_uint64(0), C.c_uint64(0)
        _check(self._lib.xsg_result_lines_size(self.h, C.byref(nl), C.byref(nb)))
        lens = np.empty(nl.value, dtype=np.uint64)
        offs = np.empty(nl.value, dtype=np.uint64)
        buf = np.empty(max(nb.value, 1), dtype=np.uint8)
        _check(self._lib.xsg_result_lines(self.h, lens.ctypes.data_as(_u64p), buf.ctypes.data, nb.value,
                                          offs.ctypes.data_as(_u64p)))
        out, pos = [], 0
        raw = buf.tobytes()
        for ln in lens:
            out.append(raw[pos:pos + int(ln)])
            pos += int(ln)
        return out, offs

    def search_lines_view(self):
        """-> (lengths, packed bytes, global offsets of the line starts): numpy VIEWS of the shard's pinned buffers, valid
        until the next search on it; no Python object per line"""
        n = C.c_uint64(0)
        _check(self._lib.xsg_search(self.h, LINES, C.byref(n)))
        lens, offs, nl, nb = _u64p(), _u64p(), C.c_uint64(0), C.c_uint64(0)
        data = C.c_void_p()
        _check(self._lib.xsg_result_lines_view(self.h, C.byref(lens), C.cast(C.byref(data), C.POINTER(C.c_char_p)), C.byref(offs),
                                               C.byref(nl), C.byref(nb)))
        if nl.value == 0:
            return np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=np.uint8), np.zeros(0, dtype=np.uint64)
        raw = (np.ctypeslib.as_array(C.cast(data, C.POINTER(C.c_uint8)), shape=(nb.value,)) if nb.value
               else np.zeros(0, dtype=np.uint8))
        return (np.ctypeslib.as_array(lens, shape=(nl.value,)), raw, np.ctypeslib.as_array(offs, shape=(nl.value,)))

    def scan_kernel_name(self, mode: int) -> str:
        buf = C.create_string_buffer(160)
        _check(self._lib.xsg_scan_kernel_name(self.h, mode, buf, 160))
        return buf.value.decode()

    def tune(self, mode: int) -> int | None:
        """measure and fix the wave stagger for this shard/pattern/mode; None = default kept"""
        v = C.c_uint32(0)
        _check(self._lib.xsg_shard_tune(self.h, mode, C.byref(v)))
        return None if v.value == 0xFFFFFFFF else int(v.value)

    def time_scan_kernel(self, mode: int, iters: int) -> float:
        ms = C.c_float(0)
        _check(self._lib.xsg_time_scan_kernel(self.h, mode, iters, C.byref(ms)))
        return ms.value


COMM_ID_BYTES = 128


def comm_unique_id() -> bytes:
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _check(load().xsg_comm_unique_id(buf, COMM_ID_BYTES))
    return buf.raw


def comm_library() -> str:
    return load().xsg_comm_library().decode()


def device_numa(device: int = 0):
    """-> (numa node or -1, local cpulist string)"""
    node = C.c_int(-1)
    buf = C.create_string_buffer(1024)
    _check(load().xsg_device_numa(device, C.byref(node), buf, 1024))
    return node.value, buf.value.decode()


def jobs_reduce_total(jobs) -> tuple[int, bool]:
    """-> (sum of the count jobs' totals, whether it was exchanged over RCCL)"""
    lib = load()
    arr = (C.c_void_p * len(jobs))(*[j.h for j in jobs])
    tot, via = C.c_uint64(0), C.c_int(0)
    _check(lib.xsg_jobs_reduce_total(arr, len(jobs), C.byref(tot), C.byref(via)))
    return tot.value, bool(via.value)


class Comm:
    """RCCL communicator of the library (include/xsg.h, 'Multi-GPU').  Rank form: Comm.rank(ctx, nranks, rank, id);
    local form: Comm.local([ctx0, ctx1, ...])."""

    def __init__(self, h, ctxs, lib):
        self.h, self.ctxs, self._lib = h, ctxs, lib
        for c in ctxs:
            c._adopt(self)

    @classmethod
    def rank(cls, ctx: "Context", nranks: int, rank: int, uid: bytes):
        lib = load()
        h = C.c_void_p()
        _check(lib.xsg_comm_create_rank(ctx.h, nranks, rank, uid, C.byref(h)))
        return cls(h, [ctx], lib)

    @classmethod
    def local(cls, ctxs):
        lib = load()
        arr = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
        h = C.c_void_p()
        _check(lib.xsg_comm_create_local(arr, len(ctxs), C.byref(h)))
        return cls(h, list(ctxs), lib)

    def size(self):
        n, r = C.c_int(0), C.c_int(0)
        _check(self._lib.xsg_comm_size(self.h, C.byref(n), C.byref(r)))
        return n.value, r.value

    def reduce_counts_async(self, d_counters: int, k: int, stream: int):
        _check(self._lib.xsg_reduce_counts_async(self.h, C.c_void_p(d_counters), k, C.c_void_p(stream)))

    def reduce_counts(self, d_counters, k: int = NUM_COUNTERS) -> np.ndarray:
        """d_counters: device address (rank form) or one per local device"""
        ptrs = [d_counters] if isinstance(d_counters, int) else list(d_counters)
        arr = (C.c_void_p * len(ptrs))(*ptrs)
        out = np.zeros(k, dtype=np.uint64)
        _check(self._lib.xsg_reduce_counts(self.h, arr, k, out.ctypes.data_as(_u64p)))
        return out

    def allgather_u64(self, mine) -> np.ndarray:
        mine = np.ascontiguousarray(np.atleast_1d(mine), dtype=np.uint64)
        n, _ = self.size()
        out = np.zeros(n, dtype=np.uint64)
        _check(self._lib.xsg_allgather_u64(self.h, mine.ctypes.data_as(_u64p), out.ctypes.data_as(_u64p)))
        return out

    def close(self):
        if self.h:
            self._lib.xsg_comm_destroy(self.h)
            self.h = None
            for c in self.ctxs:
                c._release(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------
# host-only helpers (no GPU needed): chunk plans and metafiles
# ---------------------------------------------------------------------------
def _take_chunks(ptr, n):
    arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint64)), shape=(max(n, 1) * 6,))[:n * 6].copy()
    load().xsg_free(ptr)
    return arr.view(FILE_CHUNK_DTYPE)


def plan_chunks(path: str, target_bytes: int = 16 << 20) -> np.ndarray:
    ptr, n = C.c_void_p(), C.c_uint64(0)
    _check(load().xsg_plan_chunks(os.fsencode(path), target_bytes, C.byref(ptr), C.byref(n)))
    return _take_chunks(ptr, n.value)


def meta_read(path: str, with_mappings: bool = False):
    lib = load()
    comp, ptr, n = C.c_int32(0), C.c_void_p(), C.c_uint64(0)
    mp, nm = C.c_void_p(), C.c_uint64(0)
    _check(lib.xsg_meta_read(os.fsencode(path), C.byref(comp), C.byref(ptr), C.byref(n),
                             C.byref(mp) if with_mappings else None, C.byref(nm) if with_mappings else None))
    chunks = _take_chunks(ptr, n.value)
    if not with_mappings:
        return comp.value, chunks
    maps = np.ctypeslib.as_array(C.cast(mp, C.POINTER(C.c_uint64)), shape=(max(nm.value, 1) * 2,))[:nm.value * 2].copy()
    lib.xsg_free(mp)
    return comp.value, chunks, maps.reshape(-1, 2)


def codec_name(compression: int) -> str:
    """which decoder serves a compression type on this host ("liblz4", "built-in LZ4 block codec", "libzstd", "none", "")"""
    return load().xsg_codec_name(compression).decode()


def meta_write(path: str, meta_out: str, data_out: str | None = None, compression: int = COMPRESSION_NONE,
               chunk_bytes: int = 16 << 20, mapping_gap: int = 500, hc: bool = False):
    _check(load().xsg_meta_write(os.fsencode(path), os.fsencode(meta_out), os.fsencode(data_out) if data_out else None,
                                 compression, chunk_bytes, mapping_gap, 1 if hc else 0))


class Job:
    """A file search (what xs::extern_search returns a handle to)."""

    def __init__(self, pattern: bytes, path: str, mode: int, meta_path: str | None = None, device: int = 0,
                 num_threads: int = 1, num_max_readers: int = 1, chunk_bytes: int = 16 << 20, flags: int = 0,
                 chunk_range: tuple[int, int] | None = None):
        self._lib = load()
        o = JobOpts()
        self._lib.xsg_job_opts_init(C.byref(o))
        o.mode, o.pattern_flags, o.device = mode, flags, device
        o.num_threads, o.num_max_readers, o.chunk_bytes = num_threads, num_max_readers, chunk_bytes
        if chunk_range is not None:
            o.chunk_begin, o.chunk_end = chunk_range
        h = C.c_void_p()
        _check(self._lib.xsg_job_start(pattern, len(pattern), os.fsencode(path),
                                       os.fsencode(meta_path) if meta_path else None, C.byref(o), C.byref(h)))
        self.h = h
        self.mode = mode

    def join(self):
        _check(self._lib.xsg_job_join(self.h))

    def close(self):
        if self.h:
            self._lib.xsg_job_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def total(self) -> int:
        t = C.c_uint64(0)
        _check(self._lib.xsg_job_total(self.h, C.byref(t)))
        return t.value

    def __iter__(self):
        """Live iteration: blocks until the next element exists or the job is done."""
        i = 0
        while True:
            avail, fin = C.c_uint64(0), C.c_int(0)
            _check(self._lib.xsg_job_wait(self.h, i, C.byref(avail), C.byref(fin)))
            if avail.value <= i:
                return
            while i < avail.value:
                yield self._get(i)
                i += 1

    def _get(self, i):
        if self.mode == LINES:
            p, n = C.c_char_p(), C.c_uint64(0)
            _check(self._lib.xsg_job_get_line(self.h, i, C.byref(p), C.byref(n)))
            return C.string_at(p, n.value)
        v = C.c_uint64(0)
        _check(self._lib.xsg_job_get_u64(self.h, i, 1, C.byref(v)))
        return v.value

    def result(self):
        """join + copy everything (copyResultSafe)."""
        self.join()
        if self.mode in (COUNT_MATCHES, COUNT_LINES):
            return self.total()
        n = self.total()
        if self.mode == LINES:
            return [self._get(i) for i in range(n)]
        out = np.empty(n, dtype=np.uint64)
        if n:
            _check(self._lib.xsg_job_get_u64(self.h, 0, n, out.ctypes.data_as(_u64p)))
        return out

    def stats(self) -> dict:
        st = JobStats()
        _check(self._lib.xsg_job_stats_get(self.h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in JobStats._fields_}
