"""Which parameter points of the config-5 grid fail BDF attempts, and does the CPU oracle agree?"""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
from hydromodel_amd.digest import ColumnTables, ForcingDigest
from hydromodel_amd.ensemble import SweepSimulation, merge_parameters
from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
M, D, days = 4, 300, int(sys.argv[1]) if len(sys.argv) > 1 else 26
params = default_parameters()
grid = [(n, a0, ps) for n in np.linspace(1.5, 3.0, 8) for a0 in np.geomspace(0.003, 0.03, 8)
        for ps in -np.geomspace(1e-3, 1.0, 8)]
pts = [{"Soil_Properties": {"n": float(n), "a0": float(a0), "psi_sat": float(ps)}} for n, a0, ps in grid]
well = synthetic_well(D)
cols_all = [ColumnTables(merge_parameters(params, p), well) for p in pts]
forcing = ForcingDigest(params, synthetic_forcing_frame(1), cols_all[0])
sim = SweepSimulation(cols_all, forcing, M, seed=11)
failed = np.zeros(len(pts), dtype=np.int64)
worst = None
for d in range(days):
    psi_before = sim.stepper.get_state()
    o = sim.advance(48, want_stats=True)
    f = o["failed"].reshape(48, len(pts), M)
    failed += f.sum(axis=(0, 2))
    if worst is None and f.max() >= 5:
        r, p, k = np.unravel_index(np.argmax(f), f.shape)
        worst = dict(day=d, row=int(sim.next_row - 48 + r), point=int(p), member=int(k), stats=o["stats"][r, p * M + k].tolist())
order = np.argsort(-failed)[:12]
print("failed attempts per point (top 12 of", len(pts), "), total", int(failed.sum()))
for p in order:
    print(f"  point {p}: n={grid[p][0]:.3f} a0={grid[p][1]:.4f} psi_sat={grid[p][2]:.4f}: {int(failed[p])} failed attempts")
print("points with any failure:", int((failed > 0).sum()), "first row with 5 failures:", worst)
by_n = failed.reshape(8, 64).sum(axis=1)
print("by n:", dict(zip([round(float(x), 3) for x in np.linspace(1.5, 3.0, 8)], by_n.tolist())))
by_ps = failed.reshape(8, 8, 8).sum(axis=(0, 1))
print("by psi_sat:", dict(zip([round(float(x), 4) for x in -np.geomspace(1e-3, 1.0, 8)], by_ps.tolist())))
sim.close()
