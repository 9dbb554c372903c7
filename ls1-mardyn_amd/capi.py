"""ctypes binding of libls1hip.so — the C ABI declared in include/ls1hip.h.

The library is the product; there is NO CPU fallback: if the shared object is missing or no gfx950 device is
present, calls fail loudly (``Ls1HipError``).  Build with ``make -C ls1-mardyn_amd`` (or ``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# LS1HIP_LIB: another build of the SAME library (tools/ab_variant.sh: same-box A/B timing of kernel variants); never a fallback
LIB_PATH = os.environ.get("LS1HIP_LIB") or os.path.join(_HERE, "lib", "libls1hip.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_i32p = C.POINTER(C.c_int32)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)

# every symbol include/ls1hip.h declares: (restype, argtypes)
SYMBOLS = {
    "ls1hip_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "ls1hip_destroy": (C.c_int, [C.c_void_p]),
    "ls1hip_last_error": (C.c_char_p, [C.c_void_p]),
    "ls1hip_version": (C.c_char_p, []),
    "ls1hip_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_long]),
    "ls1hip_get_option": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_long)]),
    "ls1hip_set_components": (C.c_int, [C.c_void_p, C.c_int, _ip, _ip, _ip, _ip, _dp, _dp, _dp, _dp, _dp, _dp, _dp,
                                        C.c_double, C.c_double, C.c_double]),
    "ls1hip_set_rot_dof": (C.c_int, [C.c_void_p, C.c_int, _ip]),
    "ls1hip_get_lj_table": (C.c_int, [C.c_void_p, _ip, _dp, _dp, _dp]),
    "ls1hip_set_domain": (C.c_int, [C.c_void_p, _dp, _dp, _dp, C.c_int, _ip]),
    "ls1hip_get_grid": (C.c_int, [C.c_void_p, _ip, _dp, _ip]),
    "ls1hip_upload": (C.c_int, [C.c_void_p, C.c_size_t, _u64p, _i32p, _dp, _dp, _dp, _dp]),
    "ls1hip_upload_begin": (C.c_int, [C.c_void_p, C.c_size_t]),
    "ls1hip_upload_chunk": (C.c_int, [C.c_void_p, C.c_size_t, _u64p, _i32p, _dp, _dp, _dp, _dp]),
    "ls1hip_upload_chunk_device": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p]),
    "ls1hip_upload_records": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]),
    "ls1hip_upload_end": (C.c_int, [C.c_void_p]),
    "ls1hip_download_records": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]),
    "ls1hip_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "ls1hip_download_state": (C.c_int, [C.c_void_p, C.c_size_t, _u64p, _i32p, _dp, _dp, _dp, _dp]),
    "ls1hip_download_forces": (C.c_int, [C.c_void_p, C.c_size_t, _dp, _dp, _dp]),
    "ls1hip_kick_drift": (C.c_int, [C.c_void_p, C.c_double]),
    "ls1hip_scale_kick_drift": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double]),
    "ls1hip_scale_kick_drift_components": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, C.c_double]),
    "ls1hip_kinetic_sums_by_component": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _u64p, _u64p]),
    "ls1hip_kinetic_sums": (C.c_int, [C.c_void_p, _dp, _dp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ls1hip_traversal_mark": (C.c_int, [C.c_void_p]),
    "ls1hip_traversal_sums": (C.c_int, [C.c_void_p, _dp, _dp]),
    "ls1hip_kick_then_kick_drift": (C.c_int, [C.c_void_p, C.c_double]),
    "ls1hip_rebin": (C.c_int, [C.c_void_p]),
    "ls1hip_halo": (C.c_int, [C.c_void_p]),
    "ls1hip_forces": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp]),
    "ls1hip_forces_kick_drift": (C.c_int, [C.c_void_p, C.c_int, C.c_double, _dp, _dp]),
    "ls1hip_kick": (C.c_int, [C.c_void_p, C.c_double, _dp, _dp, _u64p, _u64p]),
    "ls1hip_scale_velocities": (C.c_int, [C.c_void_p, C.c_double, C.c_double]),
    "ls1hip_set_thermostat": (C.c_int, [C.c_void_p, C.c_int, C.c_double]),
    "ls1hip_long_range_homogeneous": (C.c_int, [C.c_void_p, _u64p, C.c_double, _dp, _dp]),
    "ls1hip_set_verlet": (C.c_int, [C.c_void_p, C.c_int, C.c_double]),
    "ls1hip_verlet_build": (C.c_int, [C.c_void_p]),
    "ls1hip_halo_refresh": (C.c_int, [C.c_void_p]),
    "ls1hip_update": (C.c_int, [C.c_void_p, _ip]),
    "ls1hip_forces_list": (C.c_int, [C.c_void_p, C.c_int, C.c_double, _dp, _dp]),
    "ls1hip_forces_list_kick": (C.c_int, [C.c_void_p, C.c_double, _dp, _dp]),
    "ls1hip_verlet_poll": (C.c_int, [C.c_void_p, _ip]),
    "ls1hip_run": (C.c_int, [C.c_void_p, C.c_double, C.c_ulong, _dp]),
    "ls1hip_run_log": (C.c_int, [C.c_void_p, C.c_size_t, _dp, C.POINTER(C.c_size_t)]),
    "ls1hip_export_counts": (C.c_int, [C.c_void_p, C.c_int, _u64p]),
    "ls1hip_export_pack": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
    "ls1hip_export_pack_dirs": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_void_p, C.c_size_t]),
    "ls1hip_import": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    "ls1hip_import_done": (C.c_int, [C.c_void_p, C.c_int]),
    "ls1hip_soa_forces": (C.c_int, [C.c_void_p, _ip, _u32p, C.c_size_t, _dp, _dp, _i32p, _dp, _dp, _dp, _dp, _dp]),
    "ls1hip_timing": (C.c_int, [C.c_void_p, C.c_char_p, _dp, _u64p]),
    "ls1hip_timing_reset": (C.c_int, [C.c_void_p]),
    "ls1hip_timing_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "ls1hip_pair_stats": (C.c_int, [C.c_void_p, _u64p, _u64p]),
}

LEAVING_DOUBLES = 15
HALO_DOUBLES = 9
REFRESH_DOUBLES = 3
FK_AUTO, FK_GENERIC, FK_LDS_LIST, FK_MS_BRICK, FK_MS_SITES, FK_NEIGHBOUR_LIST = 0, 1, 2, 3, 4, 5
REC_ICRVQD, REC_ICRV, REC_IRV = 0, 1, 2
REC_BYTES = {REC_ICRVQD: 116, REC_ICRV: 60, REC_IRV: 56}


class Ls1HipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ls1hip error {code}: {msg}")
        self.code = code


_lib = None


def load():
    """Load libls1hip.so and bind every declared symbol.  Raises if the library was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Ls1HipError(-2, f"{LIB_PATH} not found — build it with `make -C {_HERE}`; there is no CPU fallback")
        # One HIP runtime per process: PyTorch wheels bundle their own libamdhip64 / libhsa-runtime64.  If libls1hip
        # pulls in the system runtime first, a later `import torch` initialises a second HSA instance and finds no GPU
        # ("No HIP GPUs are available").  Loading torch first makes both resolve to the same runtime (same SONAME).
        # torch is optional for the engine itself (it only supplies device buffers / transports to the callers).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the ABI and the header disagree
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def dptr(a):
    return None if a is None else a.ctypes.data_as(_dp)


def iptr(a):
    return None if a is None else a.ctypes.data_as(_ip)


def check(ctx, rc):
    if rc != 0:
        msg = load().ls1hip_last_error(ctx)
        raise Ls1HipError(rc, msg.decode() if msg else "?")


def f64(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a
