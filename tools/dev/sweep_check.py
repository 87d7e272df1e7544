"""Robustness sweep over the BASELINE config-5 parameter grid (reduced): every (n, a0, psi_sat) point gets its own
tables, spin-up and a short ensemble run through the generic-exponent kernel; reports solver health per point.
python tools/sweep_check.py [points_per_axis] [members] [days] [depth]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import copy
import numpy as np
from hydromodel_amd.digest import ColumnTables, ForcingDigest
from hydromodel_amd.ensemble import EnsembleSimulation
from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
days = int(sys.argv[3]) if len(sys.argv) > 3 else 2
D = int(sys.argv[4]) if len(sys.argv) > 4 else 200
params = default_parameters()
data = synthetic_forcing_frame(1)
well = synthetic_well(D)
grid = [(n, a0, ps) for n in np.linspace(1.5, 3.0, K) for a0 in np.geomspace(0.003, 0.03, K)
        for ps in -np.geomspace(1e-3, 1.0, K)]
t0 = time.perf_counter()
bad, tot_fail, tot_retry, worst = [], 0, 0, 0.0
for k, (n, a0, ps) in enumerate(grid):
    p = copy.deepcopy(params)
    p["Soil_Properties"].update({"n": float(n), "a0": float(a0), "psi_sat": float(ps)})
    try:
        cols = ColumnTables(p, well)
        forcing = ForcingDigest(p, data, cols)
        sim = EnsembleSimulation(cols, forcing, N, seed=100 + k)
        sim.advance(48 * days)
        y = sim.stepper.get_state()
        c = sim.stepper.counters()
        m = sim.moments()
        ok = np.isfinite(y).all() and c["guard_trips"] == 0 and (m[0, 1:1 + 48 * days] == N).all()
        tot_fail += c["failed_attempts"]; tot_retry += c["jac_retry"]
        rate = N * days / (sim.kernel_ms * 1e-3)
        worst = max(worst, sim.kernel_ms)
        if not ok:
            bad.append((k, n, a0, ps, "non-finite or guard"))
        if k % max(1, len(grid) // 16) == 0:
            print(f"point {k:3d} n={n:.2f} a0={a0:.4f} psi_sat={ps:.4f} spin-up {sim.spinup_iters:4d} its, "
                  f"{rate:9.0f} column-days/s, failed attempts {c['failed_attempts']}", flush=True)
        sim.close()
    except Exception as exc:                              # report, keep sweeping
        bad.append((k, n, a0, ps, repr(exc)[:120]))
print(f"{len(grid)} points x {N} members x {days} days (D={D}) in {time.perf_counter() - t0:.1f} s; "
      f"failed attempts {tot_fail}, jac retry passes {tot_retry}, slowest point {worst:.0f} ms of kernel time")
print("problem points:", bad if bad else "none")
