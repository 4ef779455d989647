"""ctypes binding of libxsg_diag.so (x-search_amd/csrc/diag/xsg_diag.hip): read-only HBM probes on raw device
buffers.  Diagnostics only -- not part of the product library, not used by any search."""
import ctypes as C
from pathlib import Path

HERE = Path(__file__).resolve().parent
_lib = None


def load():
    global _lib
    if _lib is None:
        import torch  # noqa: F401  (one HIP runtime per process: see xsg.load)
        lib = C.CDLL(str(HERE / "lib" / "libxsg_diag.so"))
        lib.xsg_diag_last_error.restype = C.c_char_p
        lib.xsg_diag_read.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_void_p,
                                      C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
        lib.xsg_diag_read_exp.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_int,
                                          C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
        _lib = lib
    return _lib


def read(d_base: int, nbytes: int, d_sink: int, tile_bytes: int = 16384, variant: int = 1, iters: int = 5):
    """-> (avg ms per launch, bytes per launch); variant 0 plain, 1 non-temporal, 2 XCD-contiguous, 3 wave-interleaved"""
    ms, nb = C.c_float(0), C.c_uint64(0)
    if load().xsg_diag_read(d_base, nbytes, tile_bytes, variant, iters, d_sink, C.byref(ms), C.byref(nb)) != 0:
        raise RuntimeError(load().xsg_diag_last_error().decode())
    return ms.value, nb.value


def read_exp(d_base: int, nbytes: int, d_sink: int, loads: int = 4, block: int = 256, stagger: int = 0, gap: int = 0,
             iters: int = 5):
    ms, nb = C.c_float(0), C.c_uint64(0)
    if load().xsg_diag_read_exp(d_base, nbytes, loads, block, stagger, gap, iters, d_sink, C.byref(ms), C.byref(nb)) != 0:
        raise RuntimeError(load().xsg_diag_last_error().decode())
    return ms.value, nb.value
