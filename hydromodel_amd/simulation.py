"""Drop-in ``Simulation`` for the reference's single-column workflow, with the time loop on the GPU.

Mirrors ``/root/reference/code/src/simulation.py``: same constructor, ``setupModel(params, data)``,
``initial_conditions()``, ``run()``, ``saveResults()``, the same ``output`` keys (:661-671) and the
same printed messages.  What differs is where the arithmetic runs: every ``RichardsPDE.solve`` call
(:452, :609) and the diagnostics plugin call (:564, :623) execute in libhydrocol on the MI355X.
Randomness follows the reference exactly: ``default_rng(SeedSequence(seed))``, draw #0 for the
spin-up, #1 for the base vector, one more per refresh row.

``lateral_flow`` / ``transpiration`` are whatever the interior ``pde_fun`` call of a row's LAST RHS
evaluation left in ``pde_model.arg_out`` (richards_pde.py:380-391, simulation.py:629-630);
the kernel integrates exactly those two sums on the last evaluation of every row when asked to.
"""
import json
import time
from pathlib import Path

import numpy as np
from numpy.random import SeedSequence, default_rng

from . import hdf5io, models
from .digest import ColumnTables, ForcingDigest, load_site_well
from .ensemble import spinup_on_gpu
from .stepper import EnsembleStepper

ROWS_PER_CALL = 240          # rows solved per library call in run(); bounds host buffers, not results


def rows_noise(base, fresh, refresh, failed):
    """The noise vector ``args_i["n_rnd"]`` holds AFTER the solve of each row of a batch -- what the diagnostics call
    ``h_model(y_i, z, args_i)`` (simulation.py:623) sees.

    ``base`` [D] is the base vector when the batch starts (modified in place, as the reference does), ``fresh``
    [n_refresh][D] the vectors drawn for the batch's refresh rows, ``refresh`` / ``failed`` [n] the rows' refresh flag and
    count of failed BDF attempts.  A non-refresh row shares the base array: every failed attempt scales it by 0.8 in
    place (richards_pde.py:522), for this row and all later ones.  A refresh row owns its vector for that row only
    (simulation.py:599-602); its failures damp that copy.  Successive multiplies, as the in-place ``*=`` does.
    Returns [n][D]."""
    out = np.empty((len(refresh), base.size))
    k = 0
    for i, (is_fresh, n_fail) in enumerate(zip(refresh, failed)):
        if is_fresh:
            v = np.array(fresh[k], dtype=float)
            k += 1
        else:
            v = base
        for _ in range(int(n_fail)):
            v *= 0.8
        out[i] = v
    return out


class Simulation(object):
    """Single soil column (N = 1) -- see the module docstring."""

    __slots__ = ("name", "mData", "rng", "output", "cols", "forcing", "device")

    def __init__(self, name=None, seed=None, device=0):
        self.name = name if name is not None else "ID_None"
        self.mData = {}
        self.rng = default_rng(SeedSequence(seed)) if seed is not None else default_rng()
        self.output = {}
        self.cols = None
        self.forcing = None
        self.device = device

    # ------------------------------------------------------------------ setup
    def setupModel(self, params, data):
        """simulation.py:79-387."""
        self.mData["Well_No"] = params["Well_No"]
        well = load_site_well(params)
        self.cols = ColumnTables(params, well)
        cols = self.cols
        print(" Selected model: Vrettas-Fung" if cols.model == 0 else " Selected model: vanGenuchten")
        self.forcing = ForcingDigest(params, data, cols)
        if cols.flags["PREDICT"] and not (params.get("Ensemble") or {}).get("repair_predict"):
            # richards_pde.py:327-330: np.linspace(..., low_lim) with a float count raises on the first RHS.  The
            # stepper can run the repaired form (int count) on request: "Ensemble": {"repair_predict": true}
            raise TypeError("'numpy.float64' object cannot be interpreted as an integer")
        # the objects the reference keeps in mData (simulation.py:208-231); h_model's call runs on the GPU
        porous = models.Porosity(cols.z, cols.layers, cols.theta, cols.soil,
                                 params["Hydrological_Model"]["Porosity_Profile"])
        self.mData.update({"soil": cols.soil, "theta": cols.theta, "K": cols.k_hc, "porosity": porous,
                           "hydro_model": models.make_model(params["Hydrological_Model"]["Name"], cols.soil, porous,
                                                            cols.k_hc, cols.theta.res, cols.dz, device=self.device)})
        self.mData.update({"layers": cols.layers, "dz": cols.dz, "z_grid": cols.z, "sat_cells": cols.sat_cells,
                           "dim_t": self.forcing.dim_t, "zWtd_cm": self.forcing.zwtd_cm,
                           "precipitation_cm": self.forcing.precip, "atm": self.forcing.atm,
                           "interception": cols.interception, "LAI": cols.lai, "iPsi_50": cols.ipsi50,
                           "surface_evap": self.forcing.surface_evap, "sim_flags": dict(cols.flags),
                           "env_param": params["Environmental"]})
        ic_data_file = params["IC_Filename"]
        if ic_data_file is None:
            self.mData["initial_cond"] = self.initial_conditions()
        else:
            import pandas as pd
            with open(Path(ic_data_file), "r") as input_file:
                init_cond = np.array(pd.read_csv(input_file, names=["IC"]).loc[:, "IC"])
            if cols.z.shape != init_cond.shape:
                raise RuntimeError(f" {self.__class__.__name__}:"
                                   f" IC vector's dimensions do not match the spatial grid.")
            print(f" IC vector was loaded successfully from: {ic_data_file}.")
            self.mData["initial_cond"] = init_cond

    def initial_conditions(self):
        """simulation.py:389-493 (burn-in on the GPU)."""
        well_no = self.mData["Well_No"]
        print(f" [Initial Conditions for Well no. {well_no}] Burn in period started ...")
        n_rnd = self.rng.standard_normal(self.cols.dim_d)
        y0, _, _ = spinup_on_gpu(self.cols, self.forcing, n_rnd, device=self.device, verbose=True, well_no=well_no)
        return y0

    # ------------------------------------------------------------------ run
    def run(self):
        """simulation.py:495-672."""
        if not self.mData:
            raise RuntimeError(f" {self.__class__.__name__}: Simulation data structure 'mData' is empty.")
        cols, forcing = self.cols, self.forcing
        z, D, T = cols.z, cols.dim_d, forcing.dim_t
        y0 = np.asarray(self.mData["initial_cond"], dtype=float).copy()
        psi = np.zeros((T, D))
        theta_vol, k_bkg, k_hrc = np.zeros((T, D)), np.zeros((T, D)), np.zeros((T, D))
        wtd_est = np.zeros(T, dtype=int)
        abs_error = np.zeros(T)
        psi[0] = y0
        transp, lateral = np.zeros(T - 1), np.zeros(T - 1)     # pde_model.arg_out after each solve (:629-630)

        st = EnsembleStepper(cols, forcing, 1, device=self.device)
        try:
            n_rnd = self.rng.standard_normal(D)                      # :561 the base vector
            st.set_state(y0)
            st.set_noise_host(n_rnd[None, :])
            diag = st.model_nodes()                                   # :564
            theta_vol[0], k_hrc[0], k_bkg[0] = diag["theta"][0], diag["K"][0], diag["K_bkg"][0]
            sat0 = y0 >= cols.soil.psi_sat
            unsat = np.nonzero(~sat0)[0]
            wtd_est[0] = min(int(unsat[-1]) + 1, D - 1) if unsat.size else 0      # find_wtd, utilities.py:83-98
            abs_error[0] = np.abs(forcing.zwtd_cm[0] - z[wtd_est[0]])

            time_t0 = time.time()
            row = 1
            while row < T:
                n = min(ROWS_PER_CALL, T - row)
                n_fresh = st.n_refresh(row, n)
                fresh = np.empty((n_fresh, 1, D))
                for k in range(n_fresh):                              # :601 one draw per refresh row, in order
                    fresh[k, 0] = self.rng.standard_normal(D)
                base_before = st.get_noise_base()[0]
                out = st.step_rows(row, n, fresh_noise=fresh, moments=False, want_wtd=True, want_psi=True,
                                   want_diag=True, want_stats=True)
                transp[row - 1:row - 1 + n] = out["diag"][:, 0, 0]
                lateral[row - 1:row - 1 + n] = out["diag"][:, 0, 1]
                psi[row:row + n] = out["psi"][:, 0, :]
                wtd_est[row:row + n] = out["wtd"][:, 0]
                abs_error[row:row + n] = np.abs(forcing.zwtd_cm[row:row + n] - z[wtd_est[row:row + n]])
                # diagnostics h_model(y_i, z, args_i) (:623) with the noise vector exactly as the row's solve left it
                noise = rows_noise(base_before, fresh[:, 0, :], forcing.refresh[row:row + n], out["failed"][:, 0])
                if not np.array_equal(base_before, st.get_noise_base()[0]):
                    raise RuntimeError(f" {self.__class__.__name__}: the base noise vector rebuilt from the per-row"
                                       f" failure counts differs from the library's (rows {row}..{row + n - 1}).")
                self._diagnostics(psi[row:row + n], noise, theta_vol[row:row + n], k_hrc[row:row + n],
                                  k_bkg[row:row + n])
                for i in range(row, row + n):
                    if np.mod(i, 100) == 0:
                        w_side = '-' if forcing.zwtd_cm[i] < z[wtd_est[i]] else '+'
                        print(" [Well No. {0}] {1}: MAE = {2:.2f} cm,"
                              " [{3}]".format(self.mData["Well_No"], i, np.mean(abs_error[0:i]), w_side))
                row += n
            print(" Elapsed time: {0:.2f} seconds.\n".format(time.time() - time_t0))
        finally:
            st.close()

        s_eff = theta_vol / cols.por_raw
        self.output["K_hrc"] = k_hrc
        self.output["K_bkg"] = k_bkg
        self.output["S_eff"] = s_eff
        self.output["psi_press"] = psi
        self.output["theta_vol"] = theta_vol
        self.output["abs_error"] = abs_error
        self.output["wtd_est_cm"] = z[wtd_est]
        self.output["lateral_flow"] = lateral
        self.output["transpiration"] = transp

    def _diagnostics(self, psi_rows, noise_rows, theta_vol, k_hrc, k_bkg):
        """Plugin call at the nodes for a batch of stored rows (each row rides as one 'member')."""
        dst = EnsembleStepper(self.cols, self.forcing, psi_rows.shape[0], device=self.device)
        try:
            dst.set_state(psi_rows)
            dst.set_noise_host(noise_rows)
            diag = dst.model_nodes()
        finally:
            dst.close()
        theta_vol[:], k_hrc[:], k_bkg[:] = diag["theta"], diag["K"], diag["K_bkg"]

    # ------------------------------------------------------------------ results
    def saveResults(self):
        """simulation.py:674-711: one gzip dataset per key in <name>.h5, written through libhdf5
        (hdf5io.py; ``.npz`` only when no HDF5 library can be loaded)."""
        if not self.output:
            print(f" {self.__class__.__name__}: Simulation data structure 'output' is empty.")
            return
        stem = self.name.strip().replace(" ", "_")
        if not hdf5io.available():
            print(f" Saving the results to: {self.name}.npz (no HDF5 library found)")
            np.savez_compressed(Path(stem + ".npz"), **self.output)
            return
        print(f" Saving the results to: {self.name}.h5")
        hdf5io.write(Path(stem + ".h5"), self.output, compression="gzip")


def loadResults(filename=None):
    """simulation.py:716-746."""
    if filename is None:
        raise RuntimeError(" load_data: No input file is given.")
    path = Path(filename)
    if path.suffix == ".npz":
        with np.load(path) as data:
            return {k: np.array(data[k]) for k in data.files}
    return hdf5io.read(path)


def dump_json(obj, path):
    with open(path, "w") as fh:
        json.dump(obj, fh, indent=1)
