"""EnsembleStepper: Python face of one libhydrocol handle (one GPU, one parameter point).

Mirrors what ``Simulation.run`` does per row (``/root/reference/code/src/simulation.py:576-626``)
for N ensemble members at once; see include/hydrocol.h for the entry points.
"""
import ctypes as C

import numpy as np

from . import _lib as L


def column_params(cols, surface_evap, flags=None):
    """Fill hc_column_params from a digest.ColumnTables."""
    fl = dict(cols.flags)
    if flags:
        fl.update(flags)
    p = L.ColumnParams()
    # PREDICT: the reference raises TypeError at richards_pde.py:327-330 (np.linspace with a float count).  The stepper
    # runs the repaired form -- `low_lim` as an int, nothing drains when it is <= 0 -- as a declared extension with
    # no reference oracle; Simulation / the CLI only reach it with "Ensemble": {"repair_predict": true}.
    p.flag_predict = int(bool(fl.get("PREDICT")))
    p.sat_cells = int(cols.sat_cells)
    p.dim_d, p.model = cols.dim_d, cols.model
    p.flag_et, p.flag_lf, p.flag_hlift = int(fl["ET"]), int(fl["LF"]), int(fl["HLIFT"])
    p.n_root_first, p.n_root_int, p.n_groups = cols.n_root_first, cols.n_root_int, cols.n_groups
    p.theta_res, p.alpha, p.n, p.m = cols.theta.res, cols.soil.alpha, cols.soil.n, cols.soil.m
    p.psi_sat, p.epsilon = cols.soil.psi_sat, max(cols.soil.epsilon, 1.0e-8)
    p.lambda_exp, p.sigma_noise, p.sat_soil = (cols.k_hc.lambda_exponent, cols.k_hc.sigma_noise,
                                               cols.k_hc.sat_soil)
    p.dz, p.ipsi50, p.lai = cols.dz, cols.ipsi50, cols.lai
    p.surface_evap, p.interception = surface_evap, cols.interception
    p.evap_delta_min = cols.evap_delta_min
    return p


class EnsembleStepper:
    """N members x D depth nodes on one MI355X.

    ``cols`` is one ``digest.ColumnTables`` or a sequence of them (parameter points of a sweep sharing grid and
    forcing): with P points the N members are point-major, point k owns members [k N/P, (k+1) N/P), and one launch
    advances all of them; ``moments()`` then returns [P][3][T]."""

    def __init__(self, cols, forcing, n_members, device=0, flags=None):
        self.lib = L.load()
        points = list(cols) if isinstance(cols, (list, tuple)) else [cols]
        cols = points[0]
        self.points, self.P = points, len(points)
        self.cols, self.forcing = cols, forcing
        self.D, self.N, self.T = cols.dim_d, int(n_members), forcing.dim_t
        if self.N % self.P:
            raise ValueError(f"{self.N} members do not divide into {self.P} parameter points")
        h = C.c_void_p()
        L.check(self.lib.hc_create(int(device), C.byref(h)))
        self.h = h
        self.params = column_params(cols, forcing.surface_evap, flags)
        node, mid = L.as_f64(cols.node_table()), L.as_f64(cols.mid_table())
        groups = np.ascontiguousarray(cols.groups, dtype=np.int32)
        L.check(self.lib.hc_set_column(self.h, C.byref(self.params), L.dptr(node), L.dptr(mid), L.iptr(groups)))
        for pt in points[1:]:
            if pt.dim_d != self.D or not np.array_equal(pt.z, cols.z):
                raise ValueError("parameter points must share the depth grid")
            pp = column_params(pt, forcing.surface_evap, flags)
            node, mid = L.as_f64(pt.node_table()), L.as_f64(pt.mid_table())
            L.check(self.lib.hc_add_point(self.h, C.byref(pp), L.dptr(node), L.dptr(mid)))
        precip, atm = L.as_f64(forcing.precip), L.as_f64(forcing.atm)
        # bit 0: daylight, bit 1: wet season (read in PREDICT mode only), include/hydrocol.h
        day = np.ascontiguousarray(forcing.daylight | (forcing.wet_season << 1), dtype=np.uint8)
        wobs = np.ascontiguousarray(forcing.wtd_obs, dtype=np.int32)
        refr = np.ascontiguousarray(forcing.refresh, dtype=np.uint8)
        L.check(self.lib.hc_set_forcing(self.h, self.T, L.dptr(precip), L.dptr(atm), L.bptr(day),
                                        L.iptr(wobs), L.bptr(refr)))
        L.check(self.lib.hc_set_members(self.h, self.N))
        self.last_kernel_ms = 0.0
        self.last_launches = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.hc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    # -- state / noise -------------------------------------------------------------
    def set_state(self, psi):
        psi = L.as_f64(psi)
        if psi.shape == (self.D,):
            L.check(self.lib.hc_set_state(self.h, L.dptr(psi), 1))
        elif psi.shape == (self.N, self.D):
            L.check(self.lib.hc_set_state(self.h, L.dptr(psi), 0))
        elif psi.shape == (self.P, self.D):                       # one column per parameter point
            L.check(self.lib.hc_set_state(self.h, L.dptr(psi), 2))
        else:
            raise ValueError(f"state must be [{self.D}], [{self.P}, {self.D}] (per point) or "
                             f"[{self.N}, {self.D}], got {psi.shape}")

    def get_state(self, first=0, count=None):
        count = self.N - first if count is None else count
        out = np.empty((count, self.D))
        L.check(self.lib.hc_get_state(self.h, L.dptr(out), first, count))
        return out

    def set_noise_host(self, base):
        base = L.as_f64(base)
        if base.shape != (self.N, self.D):
            raise ValueError(f"base noise must be [{self.N}, {self.D}]")
        L.check(self.lib.hc_set_noise_host(self.h, L.dptr(base)))

    def get_noise_base(self, first=0, count=None):
        count = self.N - first if count is None else count
        out = np.empty((count, self.D))
        L.check(self.lib.hc_get_noise_base(self.h, L.dptr(out), first, count))
        return out

    def set_noise_philox(self, seed, member_offset=0):
        L.check(self.lib.hc_set_noise_philox(self.h, int(seed), int(member_offset)))

    def philox_normals(self, member, draw):
        out = np.empty(self.D)
        L.check(self.lib.hc_philox_normals(self.h, int(member), int(draw), L.dptr(out)))
        return out

    # -- stepping ------------------------------------------------------------------
    def n_refresh(self, row_begin, n_rows):
        """Noise vectors rows [row_begin, row_begin + n_rows) consume: refresh rows that are actually solved (a row
        whose observation is off the grid is skipped before anything is drawn, simulation.py:582-602)."""
        sl = slice(row_begin, row_begin + n_rows)
        return int((self.forcing.refresh[sl].astype(bool) & (self.forcing.wtd_obs[sl] >= 0)).sum())

    def step_rows(self, row_begin, n_rows, fresh_noise=None, spinup=False, moments=True,
                  want_wtd=False, want_stats=False, want_psi=False, want_diag=False):
        a = L.StepArgs()
        a.row_begin, a.n_rows, a.spinup = int(row_begin), int(n_rows), int(spinup)
        a.accumulate_moments = int(moments and not spinup)     # spin-up solves belong to no forcing row
        keep = []
        if fresh_noise is not None:
            fresh_noise = L.as_f64(fresh_noise)
            need = 0 if spinup else self.n_refresh(row_begin, n_rows)
            if fresh_noise.size != need * self.N * self.D:
                raise ValueError(f"fresh_noise must hold {need} x [{self.N}, {self.D}] values")
            a.fresh_noise = L.dptr(fresh_noise)
            keep.append(fresh_noise)
        out = {}
        if want_wtd:
            out["wtd"] = np.zeros((n_rows, self.N), dtype=np.int32)
            a.wtd_out = L.iptr(out["wtd"])
        if want_stats:
            out["stats"] = np.zeros((n_rows, self.N, 6), dtype=np.int32)
            a.stats_out = L.iptr(out["stats"])
        if want_psi:
            out["psi"] = np.zeros((n_rows, self.N, self.D))
            a.psi_rows_out = L.dptr(out["psi"])
        if want_diag:
            out["diag"] = np.zeros((n_rows, self.N, 2))
            a.diag_out = L.dptr(out["diag"])
        L.check(self.lib.hc_step_rows(self.h, C.byref(a)))
        if want_stats:
            # slot 5 of the C-ABI record = refresh flag | failed attempts << 8 (include/hydrocol.h)
            out["failed"] = out["stats"][:, :, 5] >> 8
            out["stats"][:, :, 5] &= 0xFF
        self.last_kernel_ms, self.last_launches = a.kernel_ms, a.launches
        out["kernel_ms"], out["launches"] = a.kernel_ms, a.launches
        return out

    def counters(self):
        """{'jac_retry': ..., 'failed_attempts': ..., 'guard_trips': ...} since the handle was created."""
        out = (C.c_uint64 * 4)()
        L.check(self.lib.hc_get_counters(self.h, out))
        return {"jac_retry": int(out[0]), "failed_attempts": int(out[1]), "guard_trips": int(out[2]),
                "guard_last_member": int(out[3]) >> 24, "guard_last_row": int(out[3]) & 0xFFFFFF}

    def moments(self):
        """[3][T] (count, sum idx, sum idx^2 per forcing row); [P][3][T] when the handle holds P > 1 points."""
        m = np.zeros((self.P, 3, self.T), dtype=np.int64)
        L.check(self.lib.hc_get_moments(self.h, L.lptr(m)))
        return m[0] if self.P == 1 else m

    def export_moments(self, device_ptr):
        """Copy the moment table device-to-device to `device_ptr` (P * 3 * T int64 on this handle's device)."""
        L.check(self.lib.hc_export_moments(self.h, C.c_void_p(int(device_ptr))))

    def set_moments(self, m):
        m = np.ascontiguousarray(m, dtype=np.int64)
        if m.size != self.P * 3 * self.T:
            raise ValueError(f"moments must hold {self.P} x [3, {self.T}] values")
        L.check(self.lib.hc_set_moments(self.h, L.lptr(m)))

    def noise_scale(self, first=0, count=None):
        """Per-member damping of the Philox base vector, 0.8^(failed attempts on non-refresh rows) (richards_pde.py:522)."""
        count = self.N - first if count is None else count
        out = np.empty(count)
        L.check(self.lib.hc_get_noise_scale(self.h, L.dptr(out), int(first), int(count)))
        return out

    def set_noise_scale(self, scale, first=0):
        scale = L.as_f64(scale)
        L.check(self.lib.hc_set_noise_scale(self.h, L.dptr(scale), int(first), int(scale.size)))

    def set_point_member_bases(self, bases):
        """Global id of each parameter point's first member (Philox key of member j of point k = bases[k] + j)."""
        bases = np.ascontiguousarray(bases, dtype=np.int64)
        if bases.shape != (self.P,):
            raise ValueError(f"need one member base per parameter point ({self.P})")
        L.check(self.lib.hc_set_point_member_bases(self.h, L.lptr(bases)))

    def point_costs(self):
        """RHS evaluations spent on each parameter point's members so far ([P]; zeros for a single point)."""
        out = (C.c_uint64 * self.P)()
        L.check(self.lib.hc_get_point_costs(self.h, out))
        return np.array(list(out), dtype=np.uint64)

    def set_generic_exponents(self, on=True):
        """Pin the generic-exponent cell model (include/hydrocol.h): same bits for a point alone or inside a sweep."""
        L.check(self.lib.hc_set_generic_exponents(self.h, int(bool(on))))

    def set_rows_per_launch(self, rows):
        """Rows per kernel launch; 0 = the library's choice (48 = one simulated day for >= 65 536 members, proportionally
        more for smaller ensembles, at most a year; shorter when per-row outputs are requested -- include/hydrocol.h)."""
        L.check(self.lib.hc_set_rows_per_launch(self.h, int(rows)))

    def set_iteration_budget(self, phase_steps):
        L.check(self.lib.hc_set_iteration_budget(self.h, int(phase_steps)))

    def set_scipy_152(self, on=True):
        """``select_initial_step`` as the reference's pinned scipy==1.5.2 has it (no clamp to the interval); default: scipy >= 1.9."""
        L.check(self.lib.hc_set_scipy_152(self.h, int(bool(on))))

    def reset_moments(self):
        L.check(self.lib.hc_reset_moments(self.h))

    # -- hooks ----------------------------------------------------------------------
    def spinup(self, zwtd_cm, z0_cm, forcing_row=0, max_iterations=1500):
        """Per-member ``Simulation.initial_conditions`` (simulation.py:389-493) in one launch: every member
        iterates from its current state with its own noise vector until its own stop rule holds.
        Returns (iterations[N] -- negative where the cap was reached, kernel_ms)."""
        iters = np.zeros(self.N, dtype=np.int32)
        a = L.SpinupArgs()
        a.forcing_row, a.max_iterations = int(forcing_row), int(max_iterations)
        a.zwtd_cm, a.z0_cm = float(zwtd_cm), float(z0_cm)
        a.iterations_out = L.iptr(iters)
        L.check(self.lib.hc_spinup(self.h, C.byref(a)))
        return iters, a.kernel_ms

    def rhs(self, row, spinup=False, want_aux=False):
        out = np.empty((self.N, self.D))
        M = self.D - 1
        aux = np.empty((self.N, 3 * M + 1)) if want_aux else None
        L.check(self.lib.hc_rhs(self.h, int(row), int(spinup), L.dptr(out), L.dptr(aux) if want_aux else None))
        if not want_aux:
            return out
        return out, {"c": aux[:, :M], "s": aux[:, M:2 * M], "f": aux[:, 2 * M:3 * M], "pL": aux[:, 3 * M]}

    def model_nodes(self):
        out = np.empty((4, self.N, self.D))
        qinf = np.empty(self.N)
        L.check(self.lib.hc_model_nodes(self.h, L.dptr(out), L.dptr(qinf)))
        return {"theta": out[0], "K": out[1], "C": out[2], "K_bkg": out[3], "q_inf_max": qinf}


def moments_to_mean_std(moments, dz, z0=0.0):
    """mu/sigma of the water-table depth [cm] per row from (count, sum idx, sum idx^2): the reference reports
    ``z[wtd_est]`` with ``z[i] = z0 + i dz`` (simulation.py:140,612; ``z0 = well["soil"]``).  Accepts [3][T] or [P][3][T]."""
    moments = np.asarray(moments)
    cnt = moments[..., 0, :].astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        mean_idx = moments[..., 1, :] / cnt
        var_idx = np.maximum(moments[..., 2, :] / cnt - mean_idx ** 2, 0.0)
    return z0 + dz * mean_idx, dz * np.sqrt(var_idx)


def allreduce_handles(steppers):
    """``hc_allreduce_moments``: one process, several devices, one stepper each -- every stepper's moment table becomes
    the sum over all of them (RCCL inside the library, no torch involved)."""
    lib = L.load()
    arr = (C.c_void_p * len(steppers))(*[st.h for st in steppers])
    L.check(lib.hc_allreduce_moments(arr, len(steppers)))

