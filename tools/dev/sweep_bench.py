"""BASELINE config 5 on one GPU: P parameter points x M members in ONE handle against the same points run one
handle at a time and against one point at full occupancy.

    python tools/sweep_bench.py [P=64] [M=4096] [D=300] [days=2] [--no-alone]
"""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
from hydromodel_amd.digest import ColumnTables, ForcingDigest
from hydromodel_amd.ensemble import SweepSimulation, merge_parameters
from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well

args = [a for a in sys.argv[1:] if not a.startswith("--")]
P = int(args[0]) if len(args) > 0 else 64
M = int(args[1]) if len(args) > 1 else 4096
D = int(args[2]) if len(args) > 2 else 300
days = int(args[3]) if len(args) > 3 else 2
k = round(P ** (1 / 3))
assert k ** 3 == P, "P must be a cube (n x a0 x psi_sat grid)"
params = default_parameters()
grid = [(n, a0, ps) for n in np.linspace(1.5, 3.0, k) for a0 in np.geomspace(0.003, 0.03, k)
        for ps in -np.geomspace(1e-3, 1.0, k)]
pts = [{"Soil_Properties": {"n": float(n), "a0": float(a0), "psi_sat": float(ps)}} for n, a0, ps in grid]
well = synthetic_well(D)
cols_all = [ColumnTables(merge_parameters(params, p), well) for p in pts]
forcing = ForcingDigest(params, synthetic_forcing_frame(1), cols_all[0])
out = {"points": P, "members_per_point": M, "depth_nodes": D, "days": days}

t0 = time.perf_counter()
sim = SweepSimulation(cols_all, forcing, M, seed=11)
out["spinup_s"] = time.perf_counter() - t0
out["spinup_iterations"] = [int(x) for x in sim.spinup_iters]
sim.advance(48)
sim.kernel_ms = 0.0
sim.advance(48 * days)
one_ms = sim.kernel_ms
m_one = sim.moments()
cnt = sim.stepper.counters()
out["one_launch"] = {"kernel_ms": one_ms, "column_days_per_s": P * M * days / (one_ms * 1e-3),
                     "failed_attempts": cnt["failed_attempts"], "guard_trips": cnt["guard_trips"]}
psi0 = sim.psi0.copy()
sim.close()
print(json.dumps(out), flush=True)

if "--no-alone" not in sys.argv:
    tot_ms, same = 0.0, True
    per_point = []
    for j, c in enumerate(cols_all):
        s1 = SweepSimulation([c], forcing, M, seed=11, first_point=j, psi0=psi0[j])
        s1.advance(48)
        s1.kernel_ms = 0.0
        s1.advance(48 * days)
        tot_ms += s1.kernel_ms
        per_point.append(M * days / (s1.kernel_ms * 1e-3))
        same = same and np.array_equal(s1.moments()[0], m_one[j])
        s1.close()
    out["per_point_handles"] = {"kernel_ms": tot_ms, "column_days_per_s": P * M * days / (tot_ms * 1e-3),
                                "bit_identical_moments": bool(same),
                                "slowest_point_cds": min(per_point), "fastest_point_cds": max(per_point)}
    # the same kernel at full occupancy on ONE point (the grid's middle)
    mid = cols_all[len(cols_all) // 2]
    s2 = SweepSimulation([mid], forcing, P * M, seed=11, psi0=psi0[len(cols_all) // 2])
    s2.advance(48)
    s2.kernel_ms = 0.0
    s2.advance(48 * days)
    out["one_point_full_occupancy"] = {"kernel_ms": s2.kernel_ms,
                                       "column_days_per_s": P * M * days / (s2.kernel_ms * 1e-3),
                                       "point": pts[len(cols_all) // 2]}
    s2.close()
    out["one_launch_vs_per_point"] = out["per_point_handles"]["kernel_ms"] / one_ms
print(json.dumps(out))
