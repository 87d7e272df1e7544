"""Follow-up to fuzz_vs_oracle.py: one configuration, row by row FROM THE ORACLE'S OWN STATES (no chaining on the GPU side):
a large chained error with small per-row errors is sensitivity, a large per-row error is a defect.
    python tools/dev/fuzz_replay.py D model n lam flags(ELHP or -) roots sat wtd first rows [seed]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
from hydromodel_amd.digest import ColumnTables, ForcingDigest
from hydromodel_amd.stepper import EnsembleStepper
from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
from oracle.oracle import Oracle
D, model, n, lam, fl, roots, sat, wtd_m, first, rows = sys.argv[1:11]
D, n, lam, roots, sat, wtd_m, first, rows = int(D), float(n), float(lam), float(roots), float(sat), float(wtd_m), int(first), int(rows)
seed = int(sys.argv[11]) if len(sys.argv) > 11 else 3
params = default_parameters()
params["Hydrological_Model"]["Name"] = model
params["Soil_Properties"]["n"] = n
params["Hydraulic_Conductivity"]["Lambda_Exponent"] = lam
params["Simulation_Flags"].update({"ET": "E" in fl, "LF": "L" in fl, "HLIFT": "H" in fl, "PREDICT": "P" in fl})
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
st = EnsembleStepper(cols, forcing, 1)
seen = 0
for k in range(rows):
    row = first + k
    refresh = bool(forcing.refresh[row])
    # noise the oracle used on this row: base vector as damped so far is not tracked here -> only rows before the first retry are exact
    st.set_state(states[k][None, :]); st.set_noise_host(base[None, :])
    fz = fresh[seen][None, None, :] if refresh else np.zeros((0,))
    out = st.step_rows(row, 1, fresh_noise=fz, want_stats=True, want_wtd=True)
    seen += int(refresh)
    y1 = st.get_state()[0]
    e = np.max(np.abs(y1 - states[k + 1]) / (1 + np.abs(states[k + 1])))
    # RHS at the row's start state
    st.set_state(states[k][None, :])
    dydt_gpu = st.rhs(row)[0]
    dydt_o = o.rhs(Oracle.row(forcing.precip[row], forcing.atm[row], forcing.daylight[row], forcing.wtd_obs[row], wet=int(forcing.wet_season[row])), states[k], fresh[seen - 1] if refresh else base)
    er = np.max(np.abs(dydt_gpu - dydt_o) / np.maximum(1.0, np.abs(dydt_o)))
    print(f"row {row} day={int(forcing.daylight[row])}: one-row error {e:.1e}, RHS error {er:.1e}, GPU stats {out['stats'][0, 0, :5].tolist()} "
          f"oracle {r['per_row'][row, :5].tolist()} wtd {int(out['wtd'][0, 0])}/{int(r['wtd_est'][row])}")
st.close()
