"""Throughput of a parameter sweep at a deep column, split column vs one-wave kernels:
    python tools/dev/deep_sweep_bench.py [D=581] [points=8] [members=2048] [days=2]"""
import os, sys, time, hashlib
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
from hydromodel_amd.digest import ColumnTables, ForcingDigest
from hydromodel_amd.ensemble import SweepSimulation, check_sweep_points
from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
D = int(sys.argv[1]) if len(sys.argv) > 1 else 581
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
M = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
days = int(sys.argv[4]) if len(sys.argv) > 4 else 2
params = default_parameters()
pts = [{"Soil_Properties": {"n": float(n), "a0": float(a0)}} for n in np.linspace(1.6, 2.4, P // 2) for a0 in (0.006, 0.012)][:P]
cols_all = [ColumnTables(mp, synthetic_well(D)) for mp in check_sweep_points(params, pts)]
forcing = ForcingDigest(params, synthetic_forcing_frame(1), cols_all[0])
psi0 = None
for mode in ("default", "one-wave"):
    if mode == "one-wave":
        os.environ["HYDROCOL_SPLIT_COLUMN"] = "0"
    sim = SweepSimulation(cols_all, forcing, M, seed=3, psi0=psi0)
    psi0 = sim.psi0
    sim.advance(48)
    sim.kernel_ms = 0.0
    sim.advance(48 * days)
    print(f"D={D} {len(pts)} points x {M} members, {mode}: {len(pts) * M * days / (sim.kernel_ms * 1e-3):.0f} column-days/s, "
          f"counters {sim.stepper.counters()['failed_attempts']} failed, sha {hashlib.sha1(sim.stepper.get_state().tobytes()).hexdigest()[:10]}", flush=True)
    sim.close()
