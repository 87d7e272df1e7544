"""What the iteration budget of one BDF attempt (MAX_PHASE_ITERATIONS, hc_step.h: a deviation from SciPy, which has no
exit from a chattering attempt) changes at sweep scale.

BASELINE config 5's grid (512 points) x M members x `days` days of the 1-year forcing, TWICE in the same process from the same
spin-ups and Philox streams: with the shipped budget (20 000 phase-loop trips per attempt) and with 10 x that.  A budget trip
abandons the attempt and applies the reference's own failure rule (noise x 0.8, retry: richards_pde.py:509-533); with the
larger budget the same attempt runs on.  Reported: trips in both runs, members whose FINAL state differs at all / beyond the
integrator's accuracy class 1e-2 (1 + |psi|), the same for the water-table index of the last row, and the
change of every point's final mean water table.

    python tools/budget_effect.py [M=512] [D=300] [days=30] [factor=10]
"""
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
factor = int(sys.argv[4]) if len(sys.argv) > 4 else 10
BUDGET = 20000
params = default_parameters()
grid = [(n, a0, ps) for n in np.linspace(1.5, 3.0, 8) for a0 in np.geomspace(0.003, 0.03, 8)
        for ps in -np.geomspace(1e-3, 1.0, 8)]
pts = [{"Soil_Properties": {"n": float(n), "a0": float(a0), "psi_sat": float(ps)}} for n, a0, ps in grid]
well = synthetic_well(D)
cols_all = [ColumnTables(merge_parameters(params, p), well) for p in pts]
forcing = ForcingDigest(params, synthetic_forcing_frame(1), cols_all[0])
psi0 = None
res = {}
for name, budget in (("shipped", BUDGET), (f"x{factor}", BUDGET * factor)):
    t0 = time.perf_counter()
    sim = SweepSimulation(cols_all, forcing, M, seed=11, psi0=psi0)
    psi0 = sim.psi0                                   # the second run starts from the first run's spin-ups
    sim.stepper.set_iteration_budget(budget)
    done = 0
    while done < days:
        n = min(5, days - done)
        sim.advance(48 * n)
        done += n
        print(f"[{name}] day {done}: counters {sim.stepper.counters()}, {time.perf_counter() - t0:.0f} s", flush=True)
    y = sim.stepper.get_state()
    m = sim.moments()
    res[name] = dict(y=y, m=m, counters=sim.stepper.counters(), kernel_s=sim.kernel_ms * 1e-3)
    sim.close()
a, b = res["shipped"], res[f"x{factor}"]
P = len(pts)
d = np.abs(a["y"] - b["y"])
rel = (d / (1.0 + np.abs(b["y"]))).max(axis=1)                      # per member
differ = rel > 0
beyond = rel > 1e-2
last = 48 * days
by_point = beyond.reshape(P, M).sum(axis=1)
mean_a = 5.0 * a["m"][:, 1, last] / a["m"][:, 0, last]
mean_b = 5.0 * b["m"][:, 1, last] / b["m"][:, 0, last]
sd = lambda m: 5.0 * np.sqrt(np.maximum(m[:, 2, last] / m[:, 0, last] - (m[:, 1, last] / m[:, 0, last]) ** 2, 0.0))
out = {"points": P, "members_per_point": M, "days": days, "budget": BUDGET, "factor": factor,
       "trips_shipped": a["counters"]["guard_trips"], f"trips_x{factor}": b["counters"]["guard_trips"],
       "failed_attempts_shipped": a["counters"]["failed_attempts"], f"failed_attempts_x{factor}": b["counters"]["failed_attempts"],
       "members": P * M, "members_with_any_difference": int(differ.sum()),
       "members_beyond_1e-2_tier": int(beyond.sum()), "points_with_members_beyond_tier": int((by_point > 0).sum()),
       "max_members_beyond_tier_in_one_point": int(by_point.max()),
       "moments_identical_points": int(sum(np.array_equal(a["m"][k], b["m"][k]) for k in range(P))),
       "final_mean_wtd_cm_max_abs_change": float(np.max(np.abs(mean_a - mean_b))),
       "final_sigma_wtd_cm_max_abs_change": float(np.max(np.abs(sd(a["m"]) - sd(b["m"])))),
       "kernel_s_shipped": a["kernel_s"], f"kernel_s_x{factor}": b["kernel_s"]}
print(json.dumps(out))
