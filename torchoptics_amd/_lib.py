"""
ctypes binding of libtltrace.so (C ABI in include/tl_trace.h).

There is NO fallback: if the library is missing or a call fails, a RuntimeError is raised.
The product path never computes the trace any other way.
"""
import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# TORCHOPTICS_AMD_LIB: another build of the same library (an A/B variant written by build.build_library(tag=...))
LIB_PATH = os.environ.get("TORCHOPTICS_AMD_LIB") or os.path.join(_HERE, "libtltrace.so")

TL_ABI_VERSION = 13
TL_NMOM = 10
TL_MAX_SURFACES = 32
TL_MAX_POLY = 4
TL_MAX_HIT_SLOTS = 8
MODE_STRICT, MODE_FAST = 0, 1


class tl_problem(C.Structure):
    _fields_ = [
        ("F", C.c_int32), ("P", C.c_int32), ("W", C.c_int32), ("S", C.c_int32),
        ("device", C.c_int32), ("mode", C.c_int32), ("allow_backward", C.c_int32), ("aggregate", C.c_int32),
        ("x_in", C.c_void_p), ("y_in", C.c_void_p),
        ("xs_f", C.c_int64), ("xs_p", C.c_int64), ("xs_w", C.c_int64),
        ("ys_f", C.c_int64), ("ys_p", C.c_int64), ("ys_w", C.c_int64),
        ("z", C.c_void_p), ("cx", C.c_void_p), ("cy", C.c_void_p),
        ("cx_stride", C.c_int32), ("cy_stride", C.c_int32),
        ("c", C.c_void_p), ("t", C.c_void_p), ("mu", C.c_void_p), ("mask", C.c_void_p),
        ("kappa", C.c_void_p), ("poly", C.c_void_p), ("surf_kind", C.c_void_p), ("n_index", C.c_void_p),
        ("B", C.c_int32), ("cx_stride_b", C.c_int32), ("cy_stride_b", C.c_int32),
        ("xs_b", C.c_int64), ("ys_b", C.c_int64),
        ("asph_hits", C.c_void_p), ("asph_hit_slots", C.c_int32), ("moments_x", C.c_int32), ("cond_flags", C.c_void_p),
    ]


_lock = threading.Lock()
_lib = None

_VP = C.c_void_p
_SIGNATURES = {
    "tl_version": (C.c_int, []),
    "tl_last_error": (C.c_char_p, []),
    "tl_problem_size": (C.c_size_t, []),
    "tl_workspace_bytes": (C.c_size_t, [C.POINTER(tl_problem)]),
    "tl_trace_fwd": (C.c_int, [C.POINTER(tl_problem)] + [_VP] * 9 + [_VP, C.c_size_t, _VP]),
    "tl_trace_bwd": (C.c_int, [C.POINTER(tl_problem)] + [_VP] * 17 + [_VP, C.c_size_t, _VP]),
    "tl_trace_bwd_from_outputs": (C.c_int, [C.POINTER(tl_problem)] + [_VP] * 21 + [_VP, C.c_size_t, _VP]),
    "tl_spot_moments": (C.c_int, [C.c_int32] * 4 + [_VP] * 3 + [C.c_int64] * 3 + [_VP, _VP, C.c_size_t, _VP]),
    "tl_spot_rms": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_double, _VP, _VP, _VP, _VP]),
    "tl_unsup_loss": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_double, _VP, _VP, C.c_double, C.c_float, _VP, _VP, _VP, _VP, _VP]),
    "tl_unsup_loss_bwd": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _VP, _VP, _VP, _VP, C.c_int32, _VP, C.c_double, C.c_float,
                                    _VP, _VP]),
    "tl_aim_fan": (C.c_int, [C.c_int32] * 5 + [_VP] * 9),
    "tl_spot_seed": (C.c_int, [C.c_int32] * 4 + [_VP] * 3 + [C.c_int64] * 3 + [_VP] * 4),
    "tl_pupil_position": (C.c_int, [C.c_int32] * 3 + [_VP] * 8 + [C.c_int32, _VP]),
    "tl_workspace_bytes_f64": (C.c_size_t, [C.POINTER(tl_problem)]),
    "tl_trace_fwd_f64": (C.c_int, [C.POINTER(tl_problem)] + [_VP] * 7 + [_VP, C.c_size_t, _VP]),
    "tl_trace_bwd_f64": (C.c_int, [C.POINTER(tl_problem)] + [_VP] * 15 + [_VP, C.c_size_t, _VP]),
    "tl_selftest_arith": (C.c_int, [C.c_int32, C.c_int32, _VP, _VP, C.c_int64, _VP, _VP, _VP]),
    "tl_ray_aim": (C.c_int, [C.c_int32] * 5 + [_VP] * 12 + [C.c_int32] + [_VP] * 4),
}
EXPORTS = tuple(_SIGNATURES)


def lib():
    """Load (once) and return the CDLL; raises RuntimeError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    f"{LIB_PATH} is missing: the HIP ray-trace library has not been built. "
                    "Run `python -m torchoptics_amd.build` (needs hipcc); there is no CPU fallback.")
            try:
                dll = C.CDLL(LIB_PATH)
            except OSError as e:
                raise RuntimeError(f"cannot load {LIB_PATH}: {e}") from e
            for name, (res, args) in _SIGNATURES.items():
                fn = getattr(dll, name)
                fn.restype, fn.argtypes = res, args
            got = dll.tl_version()
            if got != TL_ABI_VERSION:
                raise RuntimeError(f"libtltrace.so ABI {got} != expected {TL_ABI_VERSION}; rebuild it")
            if dll.tl_problem_size() != C.sizeof(tl_problem):
                raise RuntimeError("tl_problem layout mismatch between _lib.py and libtltrace.so; rebuild it")
            _lib = dll
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().tl_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())
