"""ctypes wrapper around oracle/_build/libhydro_oracle.so (the CPU parity oracle).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  hydromodel_amd never imports this module.
"""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

import os

HERE = Path(__file__).resolve().parent
# HYDRO_ORACLE_LIB: another build of the same source (`make -C oracle sanitize` points it at the ASan / UBSan library)
LIB_PATH = Path(os.environ["HYDRO_ORACLE_LIB"]) if os.environ.get("HYDRO_ORACLE_LIB") else HERE / "_build" / "libhydro_oracle.so"

VIEW_NODES, VIEW_TOP, VIEW_FIRST, VIEW_INTERIOR = 0, 1, 2, 3

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_bp = C.POINTER(C.c_uint8)


class HoColumn(C.Structure):
    _fields_ = ([(k, C.c_int32) for k in ("dim_d", "model", "flag_et", "flag_lf", "flag_hlift",
                                          "n_root_first", "n_root_int", "n_groups")] +
                [(k, C.c_double) for k in ("theta_res", "alpha", "n", "m", "psi_sat", "epsilon", "lam",
                                           "sigma_noise", "sat_soil", "dz", "ipsi50", "lai",
                                           "surface_evap", "interception", "evap_delta_min")] +
                [(k, _dp) for k in ("por_node", "meank_node", "noisec_node", "por_mid", "fc_mid",
                                    "wlt_mid", "root_mid", "meank_mid", "noisec_mid")] +
                [("groups", _ip), ("flag_predict", C.c_int32), ("sat_cells", C.c_int32)])


class HoRow(C.Structure):
    _fields_ = [("precip", C.c_double), ("atm", C.c_double), ("daylight", C.c_int32),
                ("wtd_obs", C.c_int32), ("spinup", C.c_int32), ("wet", C.c_int32)]


class HoStats(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ("nfev", "njev", "nlu", "nsteps", "attempts", "success")]


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force or not LIB_PATH.exists() or \
            LIB_PATH.stat().st_mtime < max((HERE / f).stat().st_mtime
                                           for f in ("hydro_oracle.c", "hydro_oracle.h")):
        subprocess.run(["make", "-C", str(HERE)], check=True, capture_output=True)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            build()
        L = C.CDLL(str(LIB_PATH))
        L.ho_model_eval.argtypes = [C.POINTER(HoColumn), C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
        L.ho_model_eval.restype = None
        L.ho_pressure_head.argtypes = [C.POINTER(HoColumn), _dp, _dp, _dp]
        L.ho_pressure_head.restype = None
        L.ho_logn_rnd.argtypes = [C.c_double] * 3
        L.ho_logn_rnd.restype = C.c_double
        L.ho_find_wtd.argtypes = [_bp, C.c_int]
        L.ho_find_wtd.restype = C.c_int
        L.ho_rhs.argtypes = [C.POINTER(HoColumn), C.POINTER(HoRow), _dp, _dp, _dp, _dp]
        L.ho_rhs.restype = None
        L.ho_solve_row.argtypes = [C.POINTER(HoColumn), C.POINTER(HoRow), C.c_double, C.c_double,
                                   _dp, _dp, _dp, C.POINTER(HoStats), _dp, C.c_int]
        L.ho_solve_row.restype = None
        L.ho_run.argtypes = [C.POINTER(HoColumn), C.c_int64, _dp, _dp, _bp, _ip, _bp, C.c_int64,
                             C.c_int64, _dp, _dp, _dp, _ip, _dp, _ip]
        L.ho_run.restype = None
        L.ho_spinup.argtypes = [C.POINTER(HoColumn), C.POINTER(HoRow), C.c_double, _dp, _dp, _dp, C.c_int]
        L.ho_spinup.restype = C.c_int
        L.ho_last_arg_out.argtypes = [_dp]
        L.ho_last_arg_out.restype = None
        L.ho_run_diag.argtypes = [C.POINTER(HoColumn), C.c_int64, _dp, _dp, _bp, _ip, _bp, C.c_int64,
                                  C.c_int64, _dp, _dp, _dp, _ip, _dp]
        L.ho_run_diag.restype = None
        L.ho_set_scipy_152.argtypes = [C.c_int]
        L.ho_set_scipy_152.restype = None
        L.ho_clamp_counts.argtypes = [C.POINTER(C.c_long), C.c_int]
        L.ho_clamp_counts.restype = None
        L.ho_debug_set_jac_reject.argtypes = [C.c_double]
        L.ho_debug_set_jac_reject.restype = None
        L.ho_debug_jac_retry_count.argtypes = []
        L.ho_debug_jac_retry_count.restype = C.c_long
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


class Oracle:
    """One soil column (one parameter point) of the CPU oracle."""

    def __init__(self, cols, surface_evap=0.0, flags=None):
        self.cols = cols
        fl = dict(cols.flags)
        if flags:
            fl.update(flags)
        self._keep = {k: np.ascontiguousarray(getattr(cols, k), dtype=np.float64) for k in
                      ("por_node", "meank_node", "noisec_node", "por_mid", "fc_mid", "wlt_mid",
                       "root_mid", "meank_mid", "noisec_mid")}
        self._groups = np.ascontiguousarray(cols.groups, dtype=np.int32)
        c = HoColumn()
        c.dim_d, c.model = cols.dim_d, cols.model
        c.flag_et, c.flag_lf, c.flag_hlift = int(fl["ET"]), int(fl["LF"]), int(fl["HLIFT"])
        c.n_root_first, c.n_root_int, c.n_groups = cols.n_root_first, cols.n_root_int, cols.n_groups
        c.theta_res, c.alpha, c.n, c.m = cols.theta.res, cols.soil.alpha, cols.soil.n, cols.soil.m
        c.psi_sat, c.epsilon = cols.soil.psi_sat, max(cols.soil.epsilon, 1.0e-8)
        c.lam, c.sigma_noise, c.sat_soil = (cols.k_hc.lambda_exponent, cols.k_hc.sigma_noise,
                                            cols.k_hc.sat_soil)
        c.dz, c.ipsi50, c.lai = cols.dz, cols.ipsi50, cols.lai
        c.surface_evap, c.interception, c.evap_delta_min = surface_evap, cols.interception, cols.evap_delta_min
        for k, v in self._keep.items():
            setattr(c, k, _d(v))
        c.groups = self._groups.ctypes.data_as(_ip)
        # repaired PREDICT mode: an extension, pinned by nothing but its own formula (hydro_oracle.h)
        c.flag_predict, c.sat_cells = int(bool(fl.get("PREDICT"))), int(cols.sat_cells)
        if c.flag_predict and c.sat_cells < 1:
            raise ValueError("PREDICT mode needs sat_cells >= 1 (low_lim = k - (sat_cells - 1) must stay inside the slice)")
        self.c = c
        self.D = cols.dim_d

    # -- pointwise -----------------------------------------------------------------
    def model_eval(self, view, psi, n_rnd):
        psi = np.ascontiguousarray(np.atleast_1d(psi), dtype=np.float64)
        n_rnd = np.ascontiguousarray(n_rnd, dtype=np.float64)
        k = psi.size
        q, K, Cc, kb = (np.empty(k) for _ in range(4))
        qinf = C.c_double(0.0)
        lib().ho_model_eval(C.byref(self.c), view, _d(psi), _d(n_rnd), _d(q), _d(K), _d(Cc), _d(kb),
                            C.byref(qinf))
        return q, K, Cc, kb, qinf.value

    def pressure_head(self, theta):
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        psi, s = np.empty(self.D), np.empty(self.D)
        lib().ho_pressure_head(C.byref(self.c), _d(theta), _d(psi), _d(s))
        return psi, s

    @staticmethod
    def logn_rnd(mx, vx, en):
        return np.array([lib().ho_logn_rnd(float(a), float(b), float(c)) for a, b, c in zip(mx, vx, en)])

    @staticmethod
    def find_wtd(sat):
        sat = np.ascontiguousarray(sat, dtype=np.uint8)
        return lib().ho_find_wtd(sat.ctypes.data_as(_bp), sat.size)

    # -- RHS / solve -----------------------------------------------------------------
    @staticmethod
    def row(precip, atm, daylight, wtd_obs, spinup=False, wet=False):
        return HoRow(float(precip), float(atm), int(daylight), int(wtd_obs), int(spinup), int(wet))

    def rhs(self, row, y, n_rnd, want_aux=False):
        y = np.ascontiguousarray(y, dtype=np.float64)
        n_rnd = np.ascontiguousarray(n_rnd, dtype=np.float64)
        out = np.empty(self.D)
        aux = np.empty(3 * (self.D - 1) + 5) if want_aux else None
        lib().ho_rhs(C.byref(self.c), C.byref(row), _d(y), _d(n_rnd), _d(out),
                     _d(aux) if want_aux else None)
        if not want_aux:
            return out
        M = self.D - 1
        return out, {"c": aux[:M], "s": aux[M:2 * M], "f": aux[2 * M:3 * M], "pL": aux[3 * M],
                     "tr_lf_first": aux[3 * M + 1:3 * M + 3], "tr_lf_int": aux[3 * M + 3:3 * M + 5]}

    @staticmethod
    def set_scipy_152(on):
        """select_initial_step as the reference's pinned scipy==1.5.2 has it (no clamp to the interval); off = scipy >= 1.9."""
        lib().ho_set_scipy_152(int(bool(on)))

    @staticmethod
    def clamp_counts(reset=False):
        """(solves whose h0 exceeded the interval, solves whose min(100 h0, h1) did, all solves) on this thread."""
        out = (C.c_long * 3)()
        lib().ho_clamp_counts(out, int(bool(reset)))
        return int(out[0]), int(out[1]), int(out[2])

    @staticmethod
    def last_arg_out():
        """(transpiration, lateral_flow) left behind by the last RHS evaluation on this thread."""
        out = np.zeros(2)
        lib().ho_last_arg_out(_d(out))
        return out

    def run_diag(self, forcing, psi0, base_noise, fresh, row_begin=1, row_end=None):
        """Row loop recording [T][2] = transpiration, lateral_flow per row (simulation.py:629-630)."""
        T = forcing.dim_t
        row_end = T if row_end is None else row_end
        psi = np.array(psi0, dtype=np.float64)
        base = np.array(base_noise, dtype=np.float64)
        fresh = np.array(fresh, dtype=np.float64).reshape(-1, self.D)
        wtd = np.zeros(T, dtype=np.int32)
        diag = np.zeros((T, 2))
        precip = np.ascontiguousarray(forcing.precip, dtype=np.float64)
        atm = np.ascontiguousarray(forcing.atm, dtype=np.float64)
        day = np.ascontiguousarray(forcing.daylight | (forcing.wet_season << 1), dtype=np.uint8)
        wobs = np.ascontiguousarray(forcing.wtd_obs, dtype=np.int32)
        refr = np.ascontiguousarray(forcing.refresh, dtype=np.uint8)
        lib().ho_run_diag(C.byref(self.c), T, _d(precip), _d(atm), day.ctypes.data_as(_bp),
                          wobs.ctypes.data_as(_ip), refr.ctypes.data_as(_bp), row_begin, row_end, _d(psi),
                          _d(base), _d(fresh) if fresh.size else None, wtd.ctypes.data_as(_ip), _d(diag))
        return {"psi": psi, "wtd_est": wtd, "diag": diag}

    def solve_row(self, row, t0, t1, y0, n_rnd, cap_steps=0):
        """Returns (y1, stats dict, n_rnd after the in-place damping, accepted time points)."""
        y0 = np.ascontiguousarray(y0, dtype=np.float64)
        n_rnd = np.array(n_rnd, dtype=np.float64)
        y1 = np.empty(self.D)
        st = HoStats()
        ts = np.full(max(cap_steps, 1), np.nan)
        lib().ho_solve_row(C.byref(self.c), C.byref(row), float(t0), float(t1), _d(y0), _d(n_rnd), _d(y1),
                           C.byref(st), _d(ts) if cap_steps else None, cap_steps)
        stats = {k: getattr(st, k) for k, _ in HoStats._fields_}
        return y1, stats, n_rnd, ts[:min(cap_steps, stats["nsteps"] + 1)]

    def run(self, forcing, psi0, base_noise, fresh, row_begin=1, row_end=None, want_psi=False,
            want_stats=False):
        """Row loop for ONE member.  fresh: [n_refresh_in_range][D] in row order."""
        T = forcing.dim_t
        row_end = T if row_end is None else row_end
        psi = np.array(psi0, dtype=np.float64)
        base = np.array(base_noise, dtype=np.float64)
        fresh = np.array(fresh, dtype=np.float64).reshape(-1, self.D)
        need = int(forcing.refresh[max(row_begin, 1):row_end].sum())
        if fresh.shape[0] < need:
            raise ValueError(f"need {need} fresh noise vectors, got {fresh.shape[0]}")
        wtd = np.zeros(T, dtype=np.int32)
        psi_out = np.zeros((T, self.D)) if want_psi else None
        per_row = np.zeros((T, 6), dtype=np.int32) if want_stats else None
        precip = np.ascontiguousarray(forcing.precip, dtype=np.float64)
        atm = np.ascontiguousarray(forcing.atm, dtype=np.float64)
        day = np.ascontiguousarray(forcing.daylight | (forcing.wet_season << 1), dtype=np.uint8)
        wobs = np.ascontiguousarray(forcing.wtd_obs, dtype=np.int32)
        refr = np.ascontiguousarray(forcing.refresh, dtype=np.uint8)
        lib().ho_run(C.byref(self.c), T, _d(precip), _d(atm), day.ctypes.data_as(_bp),
                     wobs.ctypes.data_as(_ip), refr.ctypes.data_as(_bp), row_begin, row_end, _d(psi),
                     _d(base), _d(fresh) if fresh.size else None, wtd.ctypes.data_as(_ip),
                     _d(psi_out) if want_psi else None,
                     per_row.ctypes.data_as(_ip) if want_stats else None)
        return {"psi": psi, "wtd_est": wtd, "base_noise": base, "psi_rows": psi_out, "per_row": per_row}

    def spinup(self, row0, zwtd0_cm, psi_start, n_rnd, max_iter=1500):
        psi = np.array(psi_start, dtype=np.float64)
        n_rnd = np.array(n_rnd, dtype=np.float64)
        z = np.ascontiguousarray(self.cols.z, dtype=np.float64)
        row0 = HoRow(row0.precip, row0.atm, row0.daylight, row0.wtd_obs, 1, row0.wet)
        it = lib().ho_spinup(C.byref(self.c), C.byref(row0), float(zwtd0_cm), _d(z), _d(psi), _d(n_rnd),
                             max_iter)
        return psi, it
