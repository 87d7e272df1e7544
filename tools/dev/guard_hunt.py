"""Find attempts that exhaust the kernel's iteration budget at scale and replay one on the CPU oracle.
python tools/guard_hunt.py [N] [days]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
from hydromodel_amd.digest import ColumnTables, ForcingDigest
from hydromodel_amd.ensemble import EnsembleSimulation
from hydromodel_amd.stepper import EnsembleStepper
from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
N = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
days = int(sys.argv[2]) if len(sys.argv) > 2 else 73
params = default_parameters()
cols = ColumnTables(params, synthetic_well(300))
forcing = ForcingDigest(params, synthetic_forcing_frame(1), cols)
sim = EnsembleSimulation(cols, forcing, N, seed=17)
psi0 = sim.psi0
found = None
for d in range(days):
    t0 = time.perf_counter()
    sim.advance(48)
    c = sim.stepper.counters()
    print(f"day {d + 1}: {time.perf_counter() - t0:.1f} s, counters {c}", flush=True)
    if c["guard_trips"]:
        found = (c["guard_last_member"], c["guard_last_row"], c["guard_trips"])
        break
sim.close()
if not found:
    print("no guard trips"); sys.exit(0)
k, row, trips = found
print(f"replaying member {k} up to row {row} ({trips} trips so far)")
one = EnsembleStepper(cols, forcing, 1)
one.set_state(psi0); one.set_noise_philox(17, k)
if row > 1:
    one.step_rows(1, row - 1)
y_before = one.get_state()[0].copy()
c0 = one.counters()
out = one.step_rows(row, 1, want_stats=True)
c1 = one.counters()
print("single-member replay: stats", out["stats"][0, 0].tolist(), "guard trips in this row", c1["guard_trips"] - c0["guard_trips"],
      "kernel_ms", out["kernel_ms"])
np.savez(os.path.join(R, "gpurun_out", "guard_case.npz"), y_before=y_before, member=k, row=row, y_after=one.get_state()[0],
         base=one.philox_normals(k, 0), draw_idx=np.cumsum(forcing.refresh)[row], refresh=forcing.refresh[row],
         fresh=one.philox_normals(k, int(np.cumsum(forcing.refresh)[row])), stats=out["stats"][0, 0])
# oracle on the same row
sys.path.insert(0, os.path.join(R, "tests"))
from oracle.oracle import Oracle
o = Oracle(cols, forcing.surface_evap)
r = Oracle.row(forcing.precip[row], forcing.atm[row], forcing.daylight[row], forcing.wtd_obs[row])
nz = one.philox_normals(k, int(np.cumsum(forcing.refresh)[row])) if forcing.refresh[row] else one.philox_normals(k, 0)
t0 = time.perf_counter()
yo, so, _, _ = o.solve_row(r, row - 1, row, y_before, nz.copy())
print(f"oracle: {so} in {time.perf_counter() - t0:.2f} s; max |y_gpu - y_oracle| = {np.max(np.abs(yo - one.get_state()[0])):.3e}")
print("state range before the row:", y_before.min(), y_before.max(), "precip", forcing.precip[row], "daylight", forcing.daylight[row])
one.close()
