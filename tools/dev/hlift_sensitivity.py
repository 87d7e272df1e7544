"""How well-conditioned is one row with hydraulic lift on?  CPU only: the oracle integrates each row of a fuzz configuration
(D = 413, ET + HLIFT, rows 30 .. 39) from its own start state and from that state perturbed by 1e-13 (relative), three
times.  Night rows (lift active): 70 .. 1 700 RHS evaluations and answers that move by 1e-2 .. 1e-1; daylight rows: 1e-10 ..
1e-3.  That spread, not an implementation difference, is what the GPU-vs-oracle figures of such rows show.
    python tools/dev/hlift_sensitivity.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hydromodel_amd.digest import ColumnTables, ForcingDigest
from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
from oracle.oracle import Oracle
D, roots, sat, wtd_m, first, rows, seed = 413, 1595.0, 400.0, -3.0, 30, 10, 3
params = default_parameters()
params["Simulation_Flags"].update({"ET": True, "LF": False, "HLIFT": True, "PREDICT": False})
params["Trees"]["Max_Root_Depth_cm"] = roots
well = synthetic_well(D); well["sat_depth"] = sat
fr = synthetic_forcing_frame(1).copy(); fr["WTD_m"] = wtd_m
cols = ColumnTables(params, well); forcing = ForcingDigest(params, fr, cols)
rng = np.random.default_rng(seed)
y0 = cols.z - abs(wtd_m) * 100.0 + 0.3 * rng.standard_normal(D)
base = rng.standard_normal(D)
nf = int(forcing.refresh[first:first + rows].sum())
fresh = rng.standard_normal((nf, D))
o = Oracle(cols, forcing.surface_evap)
r = o.run(forcing, y0, base, fresh, first, first + rows, want_psi=True, want_stats=True)
states = np.vstack([y0[None, :], r["psi_rows"][first:first + rows]])
prng = np.random.default_rng(99)
for k in range(rows):
    row = first + k
    if forcing.refresh[row]: continue
    a = o.run(forcing, states[k], base, np.zeros((0, D)), row, row + 1, want_psi=True, want_stats=True)
    errs = []
    for t in range(3):
        yp = states[k] * (1.0 + 1e-13 * prng.standard_normal(D))
        b = o.run(forcing, yp, base, np.zeros((0, D)), row, row + 1, want_psi=True, want_stats=True)
        e = np.max(np.abs(a["psi_rows"][row] - b["psi_rows"][row]) / (1 + np.abs(a["psi_rows"][row])))
        errs.append((e, b["per_row"][row, :5].tolist()))
    print(f"row {row} day={int(forcing.daylight[row])}: oracle stats {a['per_row'][row, :5].tolist()}; 1e-13-perturbed start: " +
          "; ".join(f"{e:.1e} {s}" for e, s in errs), flush=True)
