"""ctypes binding of libhydrocol.so (C-ABI declared in include/hydrocol.h).

There is no CPU path: if the shared library is missing, or no gfx950 device is visible,
the calls below raise.  Build with ``python -c "import __graft_entry__ as g; g.build()"``.
"""
import ctypes as C
from pathlib import Path

import numpy as np

CSRC = Path(__file__).resolve().parent / "csrc"
LIB_PATH = CSRC / "libhydrocol.so"

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_bp = C.POINTER(C.c_uint8)
_lp = C.POINTER(C.c_int64)


class HcError(RuntimeError):
    """A libhydrocol call returned a negative status."""


class ColumnParams(C.Structure):
    """hc_column_params"""
    _fields_ = ([(k, C.c_int32) for k in ("dim_d", "model", "flag_et", "flag_lf", "flag_hlift",
                                          "n_root_first", "n_root_int", "n_groups")] +
                [(k, C.c_double) for k in ("theta_res", "alpha", "n", "m", "psi_sat", "epsilon",
                                           "lambda_exp", "sigma_noise", "sat_soil", "dz", "ipsi50", "lai",
                                           "surface_evap", "interception", "evap_delta_min")] +
                [("flag_predict", C.c_int32), ("sat_cells", C.c_int32)])


class StepArgs(C.Structure):
    """hc_step_args"""
    _fields_ = [("row_begin", C.c_int64), ("n_rows", C.c_int64), ("spinup", C.c_int32),
                ("accumulate_moments", C.c_int32), ("fresh_noise", _dp), ("wtd_out", _ip),
                ("stats_out", _ip), ("psi_rows_out", _dp), ("diag_out", _dp), ("kernel_ms", C.c_double),
                ("launches", C.c_int64)]


class SpinupArgs(C.Structure):
    """hc_spinup_args"""
    _fields_ = [("forcing_row", C.c_int64), ("max_iterations", C.c_int32), ("zwtd_cm", C.c_double),
                ("z0_cm", C.c_double), ("iterations_out", _ip), ("kernel_ms", C.c_double)]


EXPORTS = {
    "hc_create": ([C.c_int, C.POINTER(C.c_void_p)], C.c_int),
    "hc_destroy": ([C.c_void_p], C.c_int),
    "hc_last_error": ([], C.c_char_p),
    "hc_version": ([], C.c_char_p),
    "hc_set_column": ([C.c_void_p, C.POINTER(ColumnParams), _dp, _dp, _ip], C.c_int),
    "hc_add_point": ([C.c_void_p, C.POINTER(ColumnParams), _dp, _dp], C.c_int),
    "hc_get_point_count": ([C.c_void_p], C.c_int),
    "hc_set_forcing": ([C.c_void_p, C.c_int64, _dp, _dp, _bp, _ip, _bp], C.c_int),
    "hc_set_forcing_row": ([C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_uint8, C.c_int32], C.c_int),
    "hc_set_members": ([C.c_void_p, C.c_int64], C.c_int),
    "hc_set_state": ([C.c_void_p, _dp, C.c_int], C.c_int),
    "hc_get_state": ([C.c_void_p, _dp, C.c_int64, C.c_int64], C.c_int),
    "hc_set_noise_host": ([C.c_void_p, _dp], C.c_int),
    "hc_get_noise_base": ([C.c_void_p, _dp, C.c_int64, C.c_int64], C.c_int),
    "hc_set_noise_philox": ([C.c_void_p, C.c_uint64, C.c_int64], C.c_int),
    "hc_philox_normals": ([C.c_void_p, C.c_int64, C.c_int64, _dp], C.c_int),
    "hc_step_rows": ([C.c_void_p, C.POINTER(StepArgs)], C.c_int),
    "hc_spinup": ([C.c_void_p, C.POINTER(SpinupArgs)], C.c_int),
    "hc_synchronize": ([C.c_void_p], C.c_int),
    "hc_get_counters": ([C.c_void_p, C.POINTER(C.c_uint64)], C.c_int),
    "hc_set_generic_exponents": ([C.c_void_p, C.c_int32], C.c_int),
    "hc_set_rows_per_launch": ([C.c_void_p, C.c_int32], C.c_int),
    "hc_set_iteration_budget": ([C.c_void_p, C.c_int32], C.c_int),
    "hc_set_scipy_152": ([C.c_void_p, C.c_int32], C.c_int),
    "hc_get_moments": ([C.c_void_p, _lp], C.c_int),
    "hc_export_moments": ([C.c_void_p, C.c_void_p], C.c_int),
    "hc_set_moments": ([C.c_void_p, _lp], C.c_int),
    "hc_reset_moments": ([C.c_void_p], C.c_int),
    "hc_allreduce_moments": ([C.POINTER(C.c_void_p), C.c_int], C.c_int),
    "hc_get_noise_scale": ([C.c_void_p, _dp, C.c_int64, C.c_int64], C.c_int),
    "hc_set_noise_scale": ([C.c_void_p, _dp, C.c_int64, C.c_int64], C.c_int),
    "hc_set_point_member_bases": ([C.c_void_p, _lp], C.c_int),
    "hc_get_point_costs": ([C.c_void_p, C.POINTER(C.c_uint64)], C.c_int),
    "hc_rhs": ([C.c_void_p, C.c_int64, C.c_int32, _dp, _dp], C.c_int),
    "hc_model_nodes": ([C.c_void_p, _dp, _dp], C.c_int),
    "hc_plugin_eval": ([C.c_int, C.POINTER(ColumnParams), C.c_int64, C.c_int64, _dp, _dp, _dp, _dp, _dp, _dp, _dp],
                       C.c_int),
}

_lib = None


def load():
    """Load libhydrocol.so; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise HcError(f"{LIB_PATH} is missing: the HIP extension has not been built "
                          f"(run __graft_entry__.build()); there is no CPU fallback")
        lib = C.CDLL(str(LIB_PATH))
        for name, (argtypes, restype) in EXPORTS.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = restype
        _lib = lib
    return _lib


def kernel_hash():
    """The device-code identity the loaded library was built with (hc_version(): "... kernels <hash>")."""
    v = load().hc_version().decode()
    return v.rsplit("kernels ", 1)[1] if "kernels " in v else "unknown"


def check(rc):
    if rc != 0:
        raise HcError(f"libhydrocol status {rc}: {load().hc_last_error().decode()}")


def dptr(a):
    return a.ctypes.data_as(_dp)


def iptr(a):
    return a.ctypes.data_as(_ip)


def bptr(a):
    return a.ctypes.data_as(_bp)


def lptr(a):
    return a.ctypes.data_as(_lp)


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)
