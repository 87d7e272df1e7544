"""Replay the saved guard case (tools/guard_hunt.py) with explicit host noise on GPU and oracle.
python tools/guard_replay.py [member] [row]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
from hydromodel_amd.digest import ColumnTables, ForcingDigest
from hydromodel_amd.ensemble import EnsembleSimulation
from hydromodel_amd.stepper import EnsembleStepper
from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
from oracle.oracle import Oracle
k = int(sys.argv[1]) if len(sys.argv) > 1 else 165062
row = int(sys.argv[2]) if len(sys.argv) > 2 else 2140
params = default_parameters()
cols = ColumnTables(params, synthetic_well(300))
forcing = ForcingDigest(params, synthetic_forcing_frame(1), cols)
psi0 = EnsembleSimulation(cols, forcing, 1, seed=17).psi0
one = EnsembleStepper(cols, forcing, 1)
one.set_state(psi0); one.set_noise_philox(17, k)
out = one.step_rows(1, row - 1, want_stats=True)
st = out["stats"][:, 0, :]
fails = np.maximum(st[:, 4] - 1, 0) * (st[:, 5] == 0)
nscale = 0.8 ** int(fails.sum())
print("rows before:", row - 1, "failed attempts on non-refresh rows:", int(fails.sum()), "nscale", nscale,
      "max nfev before", st[:, 0].max())
y_before = one.get_state()[0].copy()
refresh = bool(forcing.refresh[row])
z = one.philox_normals(k, int(np.cumsum(forcing.refresh)[row])) if refresh else one.philox_normals(k, 0) * nscale
one.close()
h = EnsembleStepper(cols, forcing, 1)
h.set_state(y_before); h.set_noise_host(z[None, :] if not refresh else np.zeros((1, cols.dim_d)))
t0 = time.perf_counter()
o2 = h.step_rows(row, 1, fresh_noise=(z[None, None, :] if refresh else np.zeros((0,))), want_stats=True)
print("GPU host-noise replay: stats", o2["stats"][0, 0].tolist(), f"{time.perf_counter() - t0:.2f} s", h.counters())
yg = h.get_state()[0]
o = Oracle(cols, forcing.surface_evap)
r = Oracle.row(forcing.precip[row], forcing.atm[row], forcing.daylight[row], forcing.wtd_obs[row])
yo, so, _, ts = o.solve_row(r, row - 1, row, y_before, z.copy())
print("oracle:", so, "max diff", np.max(np.abs(yo - yg)))
np.savez(os.path.join(R, "gpurun_out", "guard_case2.npz"), y_before=y_before, z=z, row=row, member=k, y_gpu=yg, y_oracle=yo,
         gpu_stats=o2["stats"][0, 0], refresh=refresh)
h.close()
