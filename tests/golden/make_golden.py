#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference); nothing here runs on the
GPU box.  The reference depends on numba and h5py, which are not installed; both are
replaced by stand-in modules created in a temp dir OUTSIDE the repo
(``numba.njit`` = identity decorator, ``h5py`` = empty module), so ``_local_fast`` and
``find_wtd`` (code/src/utilities.py:4-20,56-99) execute as plain NumPy/Python.
The reference's sources are never copied: only inputs and outputs are stored.

Usage:
    python tests/golden/make_golden.py static     # G1-G4,G6  (fast, ~1 min)
    python tests/golden/make_golden.py traj 101   # G5 trajectory, well 1 (D=101), ~8 min
    python tests/golden/make_golden.py traj 200   # G5 trajectory, synthetic D=200, ~8 min
    python tests/golden/make_golden.py points     # G1-G4 at three non-default (a0, psi_sat, lambda, sigma) points, ~1 min
    python tests/golden/make_golden.py profiles   # G1 tables for the Constant / Linear / Exponential porosity and the other root pdfs
    python tests/golden/make_golden.py points_short  # first 240 rows of year-long runs at two of those points, ~1 min
    python tests/golden/make_golden.py short      # first days of vanGenuchten / HLIFT / ET+LF-off runs, ~1 min
    python tests/golden/make_golden.py spinup     # the reference's spin-up: solves used, first 12 states (wells 1 and 200), ~3 min
    python tests/golden/make_golden.py deep       # G1, G3/G4 and the first 96 rows at the reference's deepest well (no. 14, D = 581)

Vector families:
  G1 static tables / forcing digest     G2 pointwise plugin calls, pressure_head, logN_rnd, find_wtd
  G3 RHS (dydt, pde_fun, bc_fun)        G4 single-row solve + solver statistics
  G5 one-year trajectory                G6 RNG stream
"""
import json
import os
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
REF = Path("/root/reference/code")
SEED = 911


def _install_shims():
    shim = Path(tempfile.mkdtemp(prefix="hm_shim_"))
    (shim / "numba").mkdir()
    (shim / "numba" / "__init__.py").write_text(
        "def njit(*a, **k):\n"
        "    if len(a) == 1 and callable(a[0]) and not k:\n"
        "        return a[0]\n"
        "    return lambda f: f\n")
    (shim / "h5py").mkdir()
    (shim / "h5py" / "__init__.py").write_text("")
    sys.dont_write_bytecode = True
    sys.path.insert(0, str(shim))
    sys.path.insert(0, str(REF))
    sys.path.insert(0, str(REPO))


_install_shims()

import pandas as pd  # noqa: E402
import scipy  # noqa: E402
import src.richards_pde as ref_pde  # noqa: E402
from src.simulation import Simulation  # noqa: E402
from src.utilities import find_wtd, logN_rnd  # noqa: E402
from src.models.vanGenuchten import vanGenuchten  # noqa: E402

from hydromodel_amd.synthetic import (default_parameters, synthetic_forcing_frame,  # noqa: E402
                                      synthetic_well, write_site_information)

VERSIONS = {"numpy": np.__version__, "scipy": scipy.__version__, "pandas": pd.__version__,
            "python": sys.version.split()[0], "numba": "shim(identity)", "seed": SEED}


def _wells():
    with open(REF / "model_parameters" / "site_information.json") as fh:
        site = json.load(fh)
    w1 = site["Well"]["1"]
    w14 = site["Well"]["14"]          # the reference's deepest well: max_depth 2 900 cm -> D = 581 (= synthetic_well(581))
    deep = {k: w14[k] for k in ("soil", "saprolite", "weathered", "max_depth", "sat_depth")}
    assert deep == synthetic_well(581), deep
    w10 = site["Well"]["10"]          # the well input_parameters.json ships with: max_depth 2 000 cm -> D = 401, sat_depth 125
    return {1: {k: w1[k] for k in ("soil", "saprolite", "weathered", "max_depth", "sat_depth")},
            200: synthetic_well(200), 300: synthetic_well(300), 581: deep,
            401: {k: w10[k] for k in ("soil", "saprolite", "weathered", "max_depth", "sat_depth")}}


# Non-default parameter points the reference CAN run (n = 2; see DESIGN.md §8 for why n must be an even integer there):
# BASELINE config 5 sweeps (n, a0, psi_sat); lambda and sigma ride along to pin the generic-exponent kernel.
with open(HERE / "points.json") as _fh:
    POINTS = json.load(_fh)


def _setup(well_no, tmp, n_years=1, model="vrettas_fung", flags=None, quiet=True, overrides=None):
    site = write_site_information(Path(tmp) / "site.json", _wells())
    params = default_parameters()
    params["Site_Information"] = str(site)
    params["Well_No"] = well_no
    params["Hydrological_Model"]["Name"] = model
    if flags:
        params["Simulation_Flags"].update(flags)
    for section, values in (overrides or {}).items():
        params[section].update(values)
    data = synthetic_forcing_frame(n_years)
    sim = Simulation(f"golden_{well_no}", seed=SEED)
    if quiet:
        old = sys.stdout
        sys.stdout = open(os.devnull, "w")
    try:
        sim.setupModel(params, data)
    finally:
        if quiet:
            sys.stdout.close()
            sys.stdout = old
    return sim, params, data


class _SolveRecorder:
    """Wraps scipy's solve_ivp as seen from richards_pde.py:512 to record statistics."""

    def __init__(self):
        self.orig = ref_pde.solve_ivp
        self.calls = []

    def __call__(self, fun, **kw):
        sol = self.orig(fun, **kw)
        self.calls.append((sol.nfev, sol.njev, sol.nlu, len(sol.t) - 1, bool(sol.success),
                           np.array(sol.t)))
        return sol

    def __enter__(self):
        ref_pde.solve_ivp = self
        return self

    def __exit__(self, *a):
        ref_pde.solve_ivp = self.orig


def _timestamp(hour, month=11):
    return pd.Timestamp(year=2008, month=month, day=15, hour=hour)


# --------------------------------------------------------------------------- G1
def g1_tables(sim):
    m = sim.mData
    z = m["z_grid"]
    xm = sim.pde_model.x_mid
    por = m["porosity"]
    pn, fn, wn = por()
    pm, fm, wm = por(xm)
    hm = m["hydro_model"]
    (l0, l1, l2, l3) = por.layers
    out = {"z": z, "x_mid": xm, "layers": np.array(por.layers, float),
           "por_node": pn, "fc_node": fn, "wlt_node": wn,
           "por_mid": pm, "fc_mid": fm, "wlt_mid": wm}

    def mean_k(zz):
        k = np.full(zz.shape, np.nan)
        s = (zz >= l0) & (zz < l1)
        a = (zz >= l1) & (zz < l2)
        w = (zz >= l2) & (zz <= l3)
        k[s] = m["K"].sat_soil
        k[a] = hm.fun_sapr(zz[a])
        k[w] = hm.fun_wbed(zz[w])
        return k
    out["meank_node"] = mean_k(z)
    out["meank_mid"] = mean_k(xm)
    tr = m["tree_roots"]
    r = xm <= tr.max_root_depth
    out["root_mid"] = tr(xm[r])
    out["max_root_depth"] = np.array(tr.max_root_depth)
    out["atm"] = m["atm"]
    out["surface_evap"] = np.array(m["surface_evap"])
    out["iPsi_50"] = np.array(m["iPsi_50"])
    out["zWtd_cm"] = m["zWtd_cm"]
    out["hour"] = np.array([t.hour for t in m["time"]], dtype=np.int8)
    out["month"] = np.array([t.month for t in m["time"]], dtype=np.int8)
    out["precip"] = m["precipitation_cm"]
    out["initial_cond"] = m["initial_cond"]
    out["sat_cells"] = np.array(m["sat_cells"])
    out["xzmp"] = sim.pde_model.xzmp
    out["zxmp"] = sim.pde_model.zxmp
    return out


# --------------------------------------------------------------------------- G2
def g2_pointwise(sim):
    m = sim.mData
    z = m["z_grid"]
    D = z.size
    xm = sim.pde_model.x_mid
    hm = m["hydro_model"]
    vg = vanGenuchten(m["soil"], m["porosity"], m["K"], m["theta"].res, m["dz"])
    rng = np.random.default_rng(4242)
    n_rnd = rng.standard_normal(D)
    psi_sat = float(m["soil"].psi_sat)
    sweep = -np.geomspace(1e-3, 1e5, D)
    sweep[::7] = psi_sat                       # exactly at saturation
    sweep[3::11] = np.linspace(0.0, 50.0, sweep[3::11].size)
    sweep[5::13] = psi_sat * (1.0 + 1e-9)      # just below saturation
    states = {"sweep": sweep, "ic": m["initial_cond"].copy(),
              "moist": -np.abs(rng.standard_normal(D)) * 30.0,
              "dry": -2000.0 - 500.0 * rng.random(D)}
    out = {"n_rnd": n_rnd}
    for name, psi in states.items():
        out[f"psi_{name}"] = psi
        for tag, mdl in (("vf", hm), ("vg", vg)):
            # nodes (diagnostics call, simulation.py:623)
            q, K, C, kb, qi = mdl(psi.copy(), z, {"n_rnd": n_rnd.copy()})
            for k_, v in zip(("q", "K", "C", "kbkg", "qinf"), (q, K, C, kb, qi)):
                out[f"{tag}_{name}_node_{k_}"] = np.asarray(v)
            # interior midpoint slice (richards_pde.py:123-126)
            ym = 0.5 * (psi[1:-1] + psi[2:])
            q, K, C, kb, qi = mdl(ym.copy(), xm[1:], {"n_rnd": n_rnd.copy()})
            for k_, v in zip(("q", "K", "C", "kbkg"), (q, K, C, kb)):
                out[f"{tag}_{name}_mid_{k_}"] = np.asarray(v)
            # first midpoint alone (richards_pde.py:100-103)
            y0m = np.atleast_1d(0.5 * (psi[0] + psi[1]))
            q, K, C, kb, qi = mdl(y0m.copy(), np.atleast_1d(xm[0]), {"n_rnd": n_rnd.copy()})
            out[f"{tag}_{name}_first"] = np.array([q[0], K[0], C[0], kb[0]])
            # top node alone (bc_fun, richards_pde.py:435)
            q, K, C, kb, qi = mdl(np.atleast_1d(psi[0]).copy(), np.atleast_1d(z[0]),
                                  {"n_rnd": n_rnd.copy()})
            out[f"{tag}_{name}_top"] = np.array([q[0], K[0], C[0], kb[0], float(np.ravel(qi)[0])])
            out[f"{tag}_{name}_top_qlen"] = np.array(np.size(q))
    # inverse van Genuchten (hydrological_model.py:43-119)
    por = m["porosity"]()[0]
    thetas = {"porosity": por, "half": 0.5 * por + 0.02, "res": np.full(D, m["theta"].res),
              "rand": m["theta"].res + (por - m["theta"].res) * rng.random(D)}
    for name, th in thetas.items():
        p, s = hm.pressure_head(th.copy(), z)
        out[f"ph_{name}_theta"] = th
        out[f"ph_{name}_psi"] = p
        out[f"ph_{name}_seff"] = s
    # logN_rnd (utilities.py:22-54)
    mx = np.concatenate(([0.0], np.geomspace(1e-3, 10.0, 40)))
    vx = np.concatenate(([0.5], np.linspace(0.0, 2.0, 40)))
    en = rng.standard_normal(41)
    out["logn_mx"], out["logn_vx"], out["logn_en"] = mx, vx, en
    out["logn_out"] = logN_rnd(mx.copy(), vx.copy(), en.copy())
    # find_wtd known answers of code/tests/test_utilities.py:56-86 (+ extras)
    cases = [np.zeros(10, bool), np.ones(10, bool),
             np.array([0, 0, 0, 0, 0, 0, 1, 1, 1, 1], bool),
             np.array([0, 1, 0, 0, 1, 0, 0, 1, 1, 1], bool),
             np.array([1], bool), np.array([0], bool),
             np.array([1, 0, 1, 1, 0], bool)]
    out["wtd_cases"] = np.array([np.pad(c, (0, 10 - c.size)) for c in cases])
    out["wtd_sizes"] = np.array([c.size for c in cases])
    out["wtd_answers"] = np.array([find_wtd(c) for c in cases])
    return out


# ------------------------------------------------------------------------ G3/G4
def _args(m, n_rnd, hour, precip, wtd_idx, atm=None, month=11):
    return {"wtd": int(wtd_idx), "n_rnd": n_rnd, "atm": float(m["atm"][0] if atm is None else atm),
            "time": _timestamp(hour, month), "interception": m["interception"],
            "precipitation": float(precip)}


def g34_states(sim, solve=True):
    m = sim.mData
    pde = sim.pde_model
    z = m["z_grid"]
    D = z.size
    ic = m["initial_cond"].copy()
    rng = np.random.default_rng(777)
    n_rnd = rng.standard_normal(D)
    wtd_idx = int(np.where(z == m["zWtd_cm"][0])[0][0])
    top_sat = ic.copy()
    top_sat[:3] = 1.0
    wet = np.minimum(ic + 150.0, 400.0)
    rough = ic + 5.0 * rng.standard_normal(D)
    cases = [
        ("night_dry", ic, dict(hour=2, precip=0.0), {}),
        ("day_dry", ic, dict(hour=12, precip=0.0), {}),
        ("day_rain", ic, dict(hour=9, precip=0.05), {}),
        ("night_heavy_rain", ic, dict(hour=22, precip=5.0), {}),
        ("top_saturated", top_sat, dict(hour=13, precip=0.3), {}),
        ("lf_active", wet, dict(hour=3, precip=0.0), {}),
        ("lf_active_day", wet, dict(hour=15, precip=0.01), {}),
        ("spinup", ic, dict(hour=12, precip=0.02), {"SPINUP": True}),
        ("hlift_night", ic, dict(hour=1, precip=0.0), {"HLIFT": True}),
        ("dry_profile_day", ic - 3000.0, dict(hour=10, precip=0.0), {}),
        ("rough_day", rough, dict(hour=11, precip=0.6), {}),
        ("rough_night", rough, dict(hour=23, precip=0.0), {}),
        ("no_et_day", ic, dict(hour=12, precip=0.0), {"ET": False}),
        ("no_lf", wet, dict(hour=3, precip=0.0), {"LF": False}),
    ]
    out = {"n_rnd": n_rnd, "wtd_idx": np.array(wtd_idx), "names": np.array([c[0] for c in cases])}
    flags0 = dict(m["sim_flags"])
    for name, y, a, fl in cases:
        m["sim_flags"].update(flags0)
        m["sim_flags"].update(fl)
        args = _args(m, n_rnd.copy(), a["hour"], a["precip"], wtd_idx)
        out[f"{name}_y"] = y
        out[f"{name}_hour"] = np.array(a["hour"])
        out[f"{name}_precip"] = np.array(a["precip"])
        out[f"{name}_atm"] = np.array(args["atm"])
        out[f"{name}_flags"] = np.array([int(bool(m["sim_flags"][k])) for k in
                                         ("SPINUP", "ET", "LF", "HLIFT", "PREDICT")])
        out[f"{name}_dydt"] = pde(0.0, y.copy(), args)
        ym, dym = ref_pde.midpoints(z[:-1], y[:-1], z[1:], y[1:])
        cL, sL, fL = pde.pde_fun(pde.x_mid[0], ym[0], dym[0], args)
        out[f"{name}_first_csf"] = np.array([cL[0], sL[0], fL[0]])
        out[f"{name}_first_tr_lf"] = np.array([pde.arg_out["transpiration"], pde.arg_out["lateral_flow"]])
        cR, sR, fR = pde.pde_fun(pde.x_mid[1:], ym[1:], dym[1:], args)
        out[f"{name}_mid_c"], out[f"{name}_mid_s"], out[f"{name}_mid_f"] = cR, sR, fR
        out[f"{name}_mid_tr_lf"] = np.array([pde.arg_out["transpiration"], pde.arg_out["lateral_flow"]])
        pL, qL, pR, qR = pde.bc_fun(z[0], y[0], z[-1], y[-1], args)
        out[f"{name}_bc"] = np.array([pL[0], qL[0], pR[0], qR[0]])
        if solve:
            args = _args(m, n_rnd.copy(), a["hour"], a["precip"], wtd_idx)
            with _SolveRecorder() as rec:
                y1 = pde.solve((7, 8), y.copy(), args)
            out[f"{name}_solve_y"] = y1
            out[f"{name}_solve_stats"] = np.array([c[:5] for c in rec.calls], dtype=np.int64)
            out[f"{name}_solve_t"] = rec.calls[-1][5]
            out[f"{name}_solve_nrnd_after"] = args["n_rnd"]
    m["sim_flags"].update(flags0)
    return out


# --------------------------------------------------------------------------- G6
def g6_rng():
    rng = np.random.default_rng(np.random.SeedSequence(SEED))
    a = rng.standard_normal(101)
    b = rng.standard_normal(101)
    c = rng.standard_normal(101)
    k1 = np.random.default_rng(np.random.SeedSequence(SEED, spawn_key=(1,))).standard_normal(8)
    return {"draw0": a, "draw1": b, "draw2": c, "member1_first8": k1}


# --------------------------------------------------------------------------- G5
def g5_trajectory(well_no, tmp):
    sim, params, data = _setup(well_no, tmp)
    m = sim.mData
    pde = sim.pde_model
    z = m["z_grid"]
    D = z.size
    T = m["dim_t"]
    rows_full = set(range(1, 130)) | set(range(97, T, 97))
    per_row = np.zeros((T, 6), dtype=np.int32)     # nfev, njev, nlu, nsteps (last attempt), attempts, refresh
    rec_rows, rec_y0, rec_y1, rec_nin, rec_nout, rec_f0, rec_tseq = [], [], [], [], [], [], []
    state = {"i": 0}
    orig_solve = pde.solve.__func__

    # Capture the noise draws: wrap the generator's standard_normal.
    draws = []
    orig_sn = sim.rng.standard_normal

    class _Rng:
        def standard_normal(self, n):
            v = orig_sn(n)
            draws.append(v.copy())
            return v
    sim.rng = _Rng()

    def solve_wrapped(t_span, y0, *args):
        i = int(t_span[1])
        a = args[0]
        n_in = a["n_rnd"].copy()
        y0c = y0.copy()
        f0_rec = None
        if i in rows_full:
            # RHS at the row's start state, evaluated BEFORE the solve so that pde.arg_out (read by
            # Simulation.run after the solve) is left as the solve itself leaves it
            probe = dict(a)
            probe["n_rnd"] = n_in.copy()
            f0_rec = pde(float(t_span[0]), y0c.copy(), probe)
        with _SolveRecorder() as rec:
            y1 = orig_solve(pde, t_span, y0, *args)
        last = rec.calls[-1]
        per_row[i, 0] = sum(c[0] for c in rec.calls)
        per_row[i, 1] = sum(c[1] for c in rec.calls)
        per_row[i, 2] = sum(c[2] for c in rec.calls)
        per_row[i, 3] = last[3]
        per_row[i, 4] = len(rec.calls)
        failed = len(rec.calls) > 1
        if i in rows_full or (failed and len(rec_rows) < 400):
            rec_rows.append(i)
            rec_y0.append(y0c)
            rec_y1.append(np.array(y1))
            rec_nin.append(n_in)
            rec_nout.append(a["n_rnd"].copy())
            rec_f0.append(f0_rec if f0_rec is not None else np.full(y0c.shape, np.nan))
            ts = last[5]
            pad = np.full(64, np.nan)
            pad[:min(64, ts.size)] = ts[:64]
            rec_tseq.append(pad)
        return y1

    # RichardsPDE uses __slots__: patch on the class for the duration of the run.
    cls = type(pde)
    cls.solve = lambda self, t_span, y0, *args: solve_wrapped(t_span, y0, *args)
    t0 = time.time()
    try:
        old = sys.stdout
        sys.stdout = open(os.devnull, "w")
        sim.run()
    finally:
        sys.stdout.close()
        sys.stdout = old
        cls.solve = orig_solve
    wall = time.time() - t0
    precip = m["precipitation_cm"]
    per_row[1:, 5] = ((precip[1:] > 0.5) | (np.arange(1, T) % 48 == 0)).astype(np.int32)
    o = sim.output
    keep = np.arange(0, T, 48)
    out = {"wtd_est_cm": o["wtd_est_cm"], "abs_error": o["abs_error"],
           "psi_daily": o["psi_press"][keep], "theta_daily": o["theta_vol"][keep],
           "khrc_daily": o["K_hrc"][keep], "kbkg_daily": o["K_bkg"][keep],
           "psi_last": o["psi_press"][-1], "daily_rows": keep,
           "lateral_flow": o["lateral_flow"], "transpiration": o["transpiration"],
           "per_row_stats": per_row, "n_draws": np.array(len(draws)),
           "draw_first": np.array(draws[:3]), "draw_last": draws[-1],
           "initial_cond": m["initial_cond"],
           "rec_rows": np.array(rec_rows), "rec_y0": np.array(rec_y0), "rec_y1": np.array(rec_y1),
           "rec_nrnd_in": np.array(rec_nin), "rec_nrnd_out": np.array(rec_nout),
           "rec_f0": np.array(rec_f0), "rec_tseq": np.array(rec_tseq),
           "reference_wall_s": np.array(wall), "reference_column_days_per_s": np.array((T - 1) / 48.0 / wall)}
    return out


class _StopRun(Exception):
    pass


def g5_short(well_no, tmp, n_rows, model="vrettas_fung", flags=None, overrides=None):
    """First n_rows rows of the year-long run (whole-year forcing file, run() interrupted): per-row input/output."""
    sim, params, data = _setup(well_no, tmp, model=model, flags=flags, overrides=overrides)
    m = sim.mData
    pde = sim.pde_model
    cls = type(pde)
    orig_solve = cls.solve
    rec = {"rows": [], "y0": [], "y1": [], "nin": [], "nout": [], "stats": []}

    def solve_wrapped(self, t_span, y0, *args):
        i = int(t_span[1])
        if i > n_rows:
            raise _StopRun()
        a = args[0]
        n_in = a["n_rnd"].copy()
        y0c = y0.copy()
        with _SolveRecorder() as r:
            y1 = orig_solve(self, t_span, y0, *args)
        rec["rows"].append(i)
        rec["y0"].append(y0c)
        rec["y1"].append(np.array(y1))
        rec["nin"].append(n_in)
        rec["nout"].append(a["n_rnd"].copy())
        rec["stats"].append([sum(c[0] for c in r.calls), sum(c[1] for c in r.calls), sum(c[2] for c in r.calls),
                             r.calls[-1][3], len(r.calls)])
        return y1

    cls.solve = solve_wrapped
    old = sys.stdout
    sys.stdout = open(os.devnull, "w")
    try:
        sim.run()
    except _StopRun:
        pass
    finally:
        sys.stdout.close()
        sys.stdout = old
        cls.solve = orig_solve
    return {"rows": np.array(rec["rows"]), "y0": np.array(rec["y0"]), "y1": np.array(rec["y1"]),
            "nrnd_in": np.array(rec["nin"]), "nrnd_out": np.array(rec["nout"]),
            "stats": np.array(rec["stats"], dtype=np.int32), "initial_cond": m["initial_cond"],
            "flags": np.array([int(bool(m["sim_flags"][k])) for k in ("SPINUP", "ET", "LF", "HLIFT", "PREDICT")])}


def g1_spinup(well_no, tmp, keep=12):
    """Simulation.initial_conditions (simulation.py:389-493) as the reference runs it inside setupModel: how many solves
    it took, the start state, the state after each of the first `keep` solves and the solver statistics of those."""
    states, starts = [], []
    orig = ref_pde.RichardsPDE.solve

    def recording_solve(self, t_span, y0, *args):
        starts.append(np.array(y0, dtype=float))
        y = orig(self, t_span, y0, *args)
        states.append(np.array(y, dtype=float))
        return y

    ref_pde.RichardsPDE.solve = recording_solve
    try:
        with _SolveRecorder() as rec:
            sim, _, _ = _setup(well_no, tmp)
    finally:
        ref_pde.RichardsPDE.solve = orig
    assert np.array_equal(states[-1], sim.mData["initial_cond"])
    stats = np.array([c[:4] for c in rec.calls[:keep]], dtype=np.int64)
    return {"iterations": np.array(len(states)), "attempts_total": np.array(len(rec.calls)), "y_start": starts[0],
            "y_first": np.array(states[:keep]), "stats_first": stats, "initial_cond": sim.mData["initial_cond"]}


def _save(name, arrays):
    meta = json.dumps(VERSIONS)
    np.savez_compressed(HERE / name, _meta=np.array(meta), **arrays)
    print(f"wrote {name}: {len(arrays)} arrays, {(HERE / name).stat().st_size / 1024:.0f} KiB")


def main(argv):
    mode = argv[1] if len(argv) > 1 else "static"
    with tempfile.TemporaryDirectory() as tmp:
        if mode == "static":
            for well in (1, 200, 300):
                sim, _, _ = _setup(well, tmp)
                _save(f"g1_tables_{well}.npz", g1_tables(sim))
                _save(f"g2_pointwise_{well}.npz", g2_pointwise(sim))
                _save(f"g34_states_{well}.npz", g34_states(sim))
            _save("g6_rng.npz", g6_rng())
        elif mode == "points":
            # G1-G4 at non-default parameter points (n = 2), synthetic well D=200
            for tag, ov in POINTS.items():
                sim, _, _ = _setup(200, tmp, overrides=ov)
                _save(f"g1p_tables_{tag}.npz", g1_tables(sim))
                _save(f"g2p_pointwise_{tag}.npz", g2_pointwise(sim))
                _save(f"g34p_states_{tag}.npz", g34_states(sim))
        elif mode == "profiles":
            # G1 for the other porosity / root-density profile types (static tables only; well D=200)
            for por, root in (("Constant", "Uniform"), ("Linear", "Gamma_pdf"), ("Exponential", "Mixture")):
                ov = {"Hydrological_Model": {"Porosity_Profile": por}, "Trees": {"Root_Pdf_Profile": root}}
                sim, _, _ = _setup(200, tmp, overrides=ov)
                g = g1_tables(sim)
                keep = ("por_node", "fc_node", "wlt_node", "por_mid", "fc_mid", "wlt_mid", "root_mid", "max_root_depth",
                        "meank_node", "meank_mid", "initial_cond", "iPsi_50")
                _save(f"g1q_tables_{por.lower()}_{root.lower()}.npz", {k: g[k] for k in keep})
        elif mode == "points_short":
            # first 240 rows of year-long reference runs at two non-default points (lambda != 1: generic-exponent path)
            for tag in ("a03l13", "s07l08"):
                _save(f"g5sp_{tag}_200.npz", g5_short(200, tmp, 240, overrides=POINTS[tag]))
            _save("g5sp_vg_a003_200.npz", g5_short(200, tmp, 240, model="vanGenuchten", overrides=POINTS["a003"]))
        elif mode == "traj":
            well = int(argv[2])
            _save(f"g5_traj_{well}.npz", g5_trajectory(well, tmp))
        elif mode == "deep":
            # the reference's deepest well (no. 14, D = 581): tables, RHS, one-row solves and the first two days of its run --
            # the depth the split-column kernel serves
            sim, _, _ = _setup(581, tmp)
            _save("g1_tables_581.npz", g1_tables(sim))
            _save("g34_states_581.npz", g34_states(sim))
            _save("g5s_deep_581.npz", g5_short(581, tmp, 96))
            # ... and the well the reference's input_parameters.json selects (no. 10, D = 401: 7 cells per lane)
            sim, _, _ = _setup(401, tmp)
            _save("g1_tables_401.npz", g1_tables(sim))
            _save("g34_states_401.npz", g34_states(sim))
            _save("g5s_default_well_401.npz", g5_short(401, tmp, 96))
        elif mode == "spinup":
            for well in (1, 200):
                _save(f"g1s_spinup_{well}.npz", g1_spinup(well, tmp))
        elif mode == "short":
            _save("g5s_vangenuchten_200.npz", g5_short(200, tmp, 480, model="vanGenuchten"))
            _save("g5s_hlift_200.npz", g5_short(200, tmp, 240, flags={"HLIFT": True}))
            _save("g5s_noet_nolf_300.npz", g5_short(300, tmp, 240, flags={"ET": False, "LF": False}))
        else:
            raise SystemExit(__doc__)


if __name__ == "__main__":
    main(sys.argv)
