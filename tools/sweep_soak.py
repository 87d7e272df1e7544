"""BASELINE config 5's full grid on one GPU as a health check: 512 points (8 n x 8 a0 x 8 psi_sat) x M members,
`days` days in one handle; reports spin-ups, finiteness, failed attempts and iteration-budget trips per point.
    python tools/sweep_soak.py [M=512] [D=300] [days=30] [report_every_days=5]"""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
from hydromodel_amd.digest import ColumnTables, ForcingDigest
from hydromodel_amd.ensemble import SweepSimulation, merge_parameters
from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
D = int(sys.argv[2]) if len(sys.argv) > 2 else 300
days = int(sys.argv[3]) if len(sys.argv) > 3 else 30
report = int(sys.argv[4]) if len(sys.argv) > 4 else 5        # days between health checks / progress lines
params = default_parameters()
grid = [(n, a0, ps) for n in np.linspace(1.5, 3.0, 8) for a0 in np.geomspace(0.003, 0.03, 8)
        for ps in -np.geomspace(1e-3, 1.0, 8)]
pts = [{"Soil_Properties": {"n": float(n), "a0": float(a0), "psi_sat": float(ps)}} for n, a0, ps in grid]
well = synthetic_well(D)
cols_all = [ColumnTables(merge_parameters(params, p), well) for p in pts]
forcing = ForcingDigest(params, synthetic_forcing_frame(1), cols_all[0])
t0 = time.perf_counter()
sim = SweepSimulation(cols_all, forcing, M, seed=11)
t_spin = time.perf_counter() - t0
it = np.asarray(sim.spinup_iters)
print(f"{len(pts)} points x {M} members, D={D}: spin-ups {t_spin:.1f} s, iterations min/median/max {it.min()}/{int(np.median(np.abs(it)))}/{it.max()}, "
      f"capped {(it < 0).sum()}", flush=True)
done = 0
cost_prev, ms_prev = sim.stepper.point_costs().astype(float), 0.0
while done < days:
    n = min(report, days - done)
    sim.advance(48 * n)
    done += n
    y = sim.stepper.get_state()
    assert np.isfinite(y).all(), f"non-finite state after day {done}"
    cost = sim.stepper.point_costs().astype(float)
    d_cost, d_ms = cost - cost_prev, sim.kernel_ms - ms_prev
    cost_prev, ms_prev = cost, sim.kernel_ms
    # RHS evaluations per second of kernel time: constant while the launch keeps every wavefront busy
    print(f"day {done}: psi range [{y.min():.1f}, {y.max():.1f}], counters {sim.stepper.counters()}; these {n} days: "
          f"{len(pts) * M * n / (d_ms * 1e-3):.0f} column-days/s, {d_cost.sum() / (d_ms * 1e-3) / 1e9:.2f} G RHS evaluations/s, "
          f"costliest point {d_cost.max() / (M * n * 48):.1f} evaluations per column-step (median {np.median(d_cost) / (M * n * 48):.1f})",
          flush=True)
m = sim.moments()
mean_idx = m[:, 1, 1:48 * days + 1] / m[:, 0, 1:48 * days + 1]
print(json.dumps({"points": len(pts), "members": M, "days": days, "column_days_per_s": len(pts) * M * days / (sim.kernel_ms * 1e-3),
                  "kernel_s": sim.kernel_ms * 1e-3, "counters": sim.stepper.counters(),
                  "wtd_mean_cm_final_min_max": [float(5.0 * mean_idx[:, -1].min()), float(5.0 * mean_idx[:, -1].max())],
                  "spinup_iterations_max": int(it.max()), "spinup_capped": int((it < 0).sum())}))
sim.close()
