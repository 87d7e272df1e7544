"""``RichardsPDE``-shaped adapter: the reference's literal call boundary with the arithmetic on the MI355X.

The reference's row loop talks to its PDE object through three things
(``/root/reference/code/src/simulation.py:609,629-630``, ``src/richards_pde.py:82,162,478``):

    y_i = pde_model.solve(t_span, y0, args_i)        # one forcing row, <= 5 BDF attempts, noise x0.8 in place per failure
    dydt = pde_model(t, y, args_i)                   # the method-of-lines right-hand side
    pde_model.arg_out["lateral_flow" | "transpiration"]

This class offers exactly those, so that the reference's UNMODIFIED ``Simulation.run`` / ``initial_conditions`` can keep
their loops and swap only the object (option D).  Every ``solve`` is one ``hc_step_rows`` row for one
member in host-noise mode; the row's forcing values come from ``args_i`` (``hc_set_forcing_row``), the row index from
``t_span`` (the reference integrates row i over ``(i - 1, i)``, and the integrator's arithmetic depends on t).

There is no CPU path here either: without the library or a GPU the constructor raises.
"""
import numpy as np

from .stepper import EnsembleStepper

TOO_SMALL_STEP = "Required step size is less than spacing between numbers."      # scipy's message for the BDF give-up


class _RowForcing:
    """The forcing arrays of a stepper whose rows are filled in one at a time (``hc_set_forcing_row``)."""

    def __init__(self, dim_t, surface_evap):
        self.dim_t = int(dim_t)
        self.surface_evap = float(surface_evap)
        self.precip = np.zeros(self.dim_t)
        self.atm = np.zeros(self.dim_t)
        self.daylight = np.zeros(self.dim_t, dtype=np.uint8)
        self.wet_season = np.zeros(self.dim_t, dtype=np.uint8)
        self.wtd_obs = np.zeros(self.dim_t, dtype=np.int32)
        self.refresh = np.zeros(self.dim_t, dtype=np.uint8)


class RichardsPDE(object):
    """``RichardsPDE(m_data)`` -- richards_pde.py:22-80.

    ``m_data`` is the dictionary ``Simulation.setupModel`` fills (``mData``).  Besides the reference's keys
    (``z_grid``, ``sim_flags``, ``surface_evap``, ``dim_t``) it must carry the column digest under ``"cols"``
    (``digest.ColumnTables``: the static tables of simulation.py:79-387 as the kernels read them); this package's own
    ``Simulation`` stores it there, a reference maintainer adds ``self.mData["cols"] = ColumnTables(params, well)``.
    """

    def __init__(self, m_data=None, device=0):
        if m_data is None:
            raise ValueError(f" {self.__class__.__name__}:"
                             f" No input is given. The model cannot initialize.")
        if "cols" not in m_data:
            raise ValueError(f" {self.__class__.__name__}: m_data carries no column digest ('cols').")
        self.var_arg_out = {"transpiration": 0.0, "lateral_flow": 0.0}
        self.m_data = m_data
        self.h_model = m_data.get("hydro_model")
        self.x_mesh = np.asarray(m_data["z_grid"], dtype=float)
        self.sim_flags = m_data["sim_flags"]                     # (a reference: SPINUP is flipped by the caller, :398,:485)
        self.nx = self.x_mesh.size
        if np.any(np.diff(self.x_mesh, axis=0) <= 0.0):
            raise RuntimeError(f" {self.__class__.__name__}:"
                               f" Space domain is not increasing.")
        self.cols = m_data["cols"]
        if self.cols.dim_d != self.nx:
            raise ValueError(f" {self.__class__.__name__}: the column digest has {self.cols.dim_d} nodes, z_grid {self.nx}.")
        self._forcing = _RowForcing(max(int(m_data.get("dim_t", 2)), 2), m_data.get("surface_evap", 0.0))
        flags = {k: bool(self.sim_flags[k]) for k in ("ET", "LF", "HLIFT", "PREDICT") if k in self.sim_flags}
        self._st = EnsembleStepper(self.cols, self._forcing, 1, device=device, flags=flags)
        self.last_stats = None       # (nfev, njev, nlu, steps, attempts, failed attempts) of the last solve

    def close(self):
        if getattr(self, "_st", None) is not None:
            self._st.close()
            self._st = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    @property
    def arg_out(self):
        """richards_pde.py:162-170."""
        return self.var_arg_out

    # ------------------------------------------------------------------ internals
    def _set_row(self, row, args_i):
        from . import _lib as L
        stamp = args_i["time"]
        daylight = int(6 <= stamp.hour <= 17)                                      # richards_pde.py:230
        wet = int(stamp.month in (10, 11, 12, 1, 2, 3))                           # :315 (PREDICT mode only)
        L.check(self._st.lib.hc_set_forcing_row(self._st.h, int(row), float(args_i["precipitation"]), float(args_i["atm"]),
                                                daylight | (wet << 1), int(args_i["wtd"])))

    def _load(self, y, args_i):
        y = np.ascontiguousarray(y, dtype=float)
        if y.shape != (self.nx,):
            raise ValueError(f" {self.__class__.__name__}: state must be [{self.nx}], got {y.shape}")
        n_rnd = np.ascontiguousarray(args_i["n_rnd"], dtype=float)
        self._st.set_state(y)
        self._st.set_noise_host(n_rnd[None, :])

    # ------------------------------------------------------------------ the reference's surface
    def __call__(self, t, y, *args):
        """dy/dt at (t, y) -- richards_pde.py:82-160 (the RHS does not depend on t; the forcing comes from ``args``)."""
        args_i = args[0]
        self._load(y, args_i)
        self._set_row(1, args_i)                                  # row 1 is scratch: solve() writes its own row
        return self._st.rhs(1, spinup=bool(self.sim_flags.get("SPINUP", False)))[0]

    def solve(self, t_span, y0, *args):
        """richards_pde.py:478-537: integrate ``t_span`` from ``y0``; up to five attempts, ``args[0]["n_rnd"] *= 0.8``
        in place after each failed one; returns ``sol.y[:, -1]`` of the last attempt."""
        args_i = args[0]
        t0, tf = float(t_span[0]), float(t_span[1])
        row = int(round(tf))
        spinup = bool(self.sim_flags.get("SPINUP", False))
        if spinup:
            if (t0, tf) != (0.0, 1.0):
                raise ValueError(f" {self.__class__.__name__}: a SPINUP solve integrates (0, 1), got {t_span}")
            row = 1
        elif tf != row or t0 != row - 1 or row < 1:
            raise ValueError(f" {self.__class__.__name__}: the stepper integrates one forcing row, t_span = (i - 1, i) "
                             f"with integer i >= 1 (simulation.py:606), got {t_span}")
        if row >= self._forcing.dim_t:
            raise ValueError(f" {self.__class__.__name__}: row {row} beyond the {self._forcing.dim_t} rows of m_data['dim_t']")
        self._load(y0, args_i)
        self._set_row(row, args_i)
        out = self._st.step_rows(row, 1, fresh_noise=np.zeros((0, 1, self.nx)), spinup=spinup, moments=False,
                                 want_stats=True, want_diag=True)
        failed = int(out["failed"][0, 0])
        if failed:
            # the library damped its copy by the same successive in-place multiplies (richards_pde.py:522)
            args_i["n_rnd"][...] = self._st.get_noise_base()[0]
        if failed >= 5:
            print(f" {self.__class__.__name__}:"
                  f" The ODE solver failed with message: {TOO_SMALL_STEP}")
        s = out["stats"][0, 0]
        self.last_stats = {"nfev": int(s[0]), "njev": int(s[1]), "nlu": int(s[2]), "steps": int(s[3]),
                           "attempts": int(s[4]), "failed": failed}
        self.var_arg_out["transpiration"] = float(out["diag"][0, 0, 0])
        self.var_arg_out["lateral_flow"] = float(out["diag"][0, 0, 1])
        return self._st.get_state()[0]
