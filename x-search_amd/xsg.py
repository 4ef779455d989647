"""ctypes binding of the C ABI in include/xsg.h (libxsg.so).

Plain pointers and sizes only: device memory is passed as integer addresses
(e.g. torch.Tensor.data_ptr()), streams as integer hipStream_t handles.  This
module never falls back to a CPU implementation: if the library or the GPU is
missing, the call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent

OK = 0
EINVAL, ENODEV, EHIP, ENOMEM, ENOTSUP, EIO, ESTATE = -1, -2, -3, -4, -5, -6, -7

COUNT_MATCHES, COUNT_LINES, MATCH_BYTE_OFFSETS, LINE_BYTE_OFFSETS, LINE_INDICES, LINES = range(6)
FLAG_EXACT_TAIL = 0x1
WITH_NEWLINES = 0x100
CTR_MATCHES, CTR_LINES, CTR_NEWLINES, CTR_BYTES = range(4)
NUM_COUNTERS = 4
LINE_BASE_AUTO = (1 << 64) - 1
MAX_PATTERN = 1024
TILE = 16384

CHUNK_DTYPE = np.dtype([("offset", "<u8"), ("length", "<u8"), ("global_offset", "<u8"), ("line_base", "<u8")])

_u64p = C.POINTER(C.c_uint64)


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
    p = lib_path()
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
        "xsg_shard_create": (ci, [vp, vp, u64, vp, u64, C.POINTER(vp)]),
        "xsg_shard_rebind": (ci, [vp, vp, u64, vp, u64]),
        "xsg_shard_destroy": (None, [vp]),
        "xsg_shard_set_line_base": (ci, [vp, u64]),
        "xsg_count_async": (ci, [vp, u32, vp, vp]),
        "xsg_count": (ci, [vp, u32, _u64p]),
        "xsg_search": (ci, [vp, u32, _u64p]),
        "xsg_result_u64": (ci, [vp, _u64p, u64]),
        "xsg_result_lines_size": (ci, [vp, _u64p, _u64p]),
        "xsg_result_lines": (ci, [vp, _u64p, vp, u64, _u64p]),
        "xsg_ctx_info": (ci, [vp, C.c_char_p, sz, C.POINTER(ci), _u64p]),
        "xsg_time_scan_kernel": (ci, [vp, u32, ci, C.POINTER(C.c_float)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


EXPORTS = ["xsg_abi_version", "xsg_strerror", "xsg_last_error", "xsg_device_count", "xsg_ctx_create",
           "xsg_ctx_destroy", "xsg_set_pattern", "xsg_shard_create", "xsg_shard_rebind", "xsg_shard_destroy",
           "xsg_shard_set_line_base", "xsg_count_async", "xsg_count", "xsg_search", "xsg_result_u64",
           "xsg_result_lines_size", "xsg_result_lines", "xsg_ctx_info", "xsg_time_scan_kernel"]


def _check(rc):
    if rc != OK:
        raise XsgError(rc, load().xsg_last_error().decode("utf-8", "replace"))


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

    def close(self):
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
    def __init__(self, ctx: Context, d_base: int, capacity: int, chunks: np.ndarray):
        self._lib = ctx._lib
        self.ctx = ctx
        chunks = np.ascontiguousarray(chunks, dtype=CHUNK_DTYPE)
        h = C.c_void_p()
        _check(self._lib.xsg_shard_create(ctx.h, C.c_void_p(d_base), capacity, chunks.ctypes.data, len(chunks),
                                          C.byref(h)))
        self.h = h
        self.nchunks = len(chunks)

    def close(self):
        if self.h:
            self._lib.xsg_shard_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def rebind(self, d_base: int, capacity: int, chunks: np.ndarray):
        chunks = np.ascontiguousarray(chunks, dtype=CHUNK_DTYPE)
        _check(self._lib.xsg_shard_rebind(self.h, C.c_void_p(d_base), capacity, chunks.ctypes.data, len(chunks)))
        self.nchunks = len(chunks)

    def set_line_base(self, base: int):
        _check(self._lib.xsg_shard_set_line_base(self.h, base))

    def count_async(self, mode: int, stream: int, d_counters: int):
        _check(self._lib.xsg_count_async(self.h, mode, C.c_void_p(stream), C.c_void_p(d_counters)))

    def count(self, mode: int) -> np.ndarray:
        out = np.zeros(NUM_COUNTERS, dtype=np.uint64)
        _check(self._lib.xsg_count(self.h, mode, out.ctypes.data_as(_u64p)))
        return out

    def search_u64(self, mode: int) -> np.ndarray:
        n = C.c_uint64(0)
        _check(self._lib.xsg_search(self.h, mode, C.byref(n)))
        out = np.empty(n.value, dtype=np.uint64)
        _check(self._lib.xsg_result_u64(self.h, out.ctypes.data_as(_u64p), n.value))
        return out

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

    def time_scan_kernel(self, mode: int, iters: int) -> float:
        ms = C.c_float(0)
        _check(self._lib.xsg_time_scan_kernel(self.h, mode, iters, C.byref(ms)))
        return ms.value
