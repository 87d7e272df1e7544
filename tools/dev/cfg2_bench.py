"""BASELINE config 2 (4 096 members x D=200, the whole 1-year forcing) at several launch lengths.
    python tools/cfg2_bench.py [rows_per_launch ...]   (default: 48 480 4380 17472)"""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
from hydromodel_amd.digest import ColumnTables, ForcingDigest
from hydromodel_amd.ensemble import EnsembleSimulation
from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
params = default_parameters()
cols = ColumnTables(params, synthetic_well(200))
forcing = ForcingDigest(params, synthetic_forcing_frame(1), cols)
psi0 = None
ref = None
for rpl in [int(a) for a in sys.argv[1:]] or [48, 480, 4380, 17472]:
    sim = EnsembleSimulation(cols, forcing, 4096, seed=2024, psi0=psi0)
    psi0 = sim.psi0
    sim.stepper.set_rows_per_launch(rpl)
    sim.advance(48)                                   # day 1: warm-up
    sim.kernel_ms = 0.0
    t0 = time.perf_counter()
    sim.advance(forcing.dim_t - 1 - 48)               # days 2..365 (17 471 rows)
    wall = time.perf_counter() - t0
    days = (forcing.dim_t - 1 - 48) / 48.0
    m = sim.moments()
    same = True if ref is None else bool(np.array_equal(m, ref))
    ref = m if ref is None else ref
    print(json.dumps({"rows_per_launch": rpl, "launches": sim.launches, "column_days_per_s": 4096 * days / wall,
                      "kernel_s": sim.kernel_ms * 1e-3, "wall_s": wall, "moments_identical_to_first": same,
                      "counters": sim.stepper.counters()}), flush=True)
    sim.close()
