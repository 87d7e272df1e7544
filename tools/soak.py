"""Soak run: N members x the whole 10-year synthetic forcing (175 199 rows), Philox noise.
Checks every launch for finite states and reports solver-health counters.  python tools/soak.py [N] [depth] [years]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
from hydromodel_amd.digest import ColumnTables, ForcingDigest
from hydromodel_amd.ensemble import EnsembleSimulation
from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
D = int(sys.argv[2]) if len(sys.argv) > 2 else 300
years = int(sys.argv[3]) if len(sys.argv) > 3 else 10
params = default_parameters()
cols = ColumnTables(params, synthetic_well(D))
forcing = ForcingDigest(params, synthetic_forcing_frame(years), cols)
sim = EnsembleSimulation(cols, forcing, N, seed=17)
T = forcing.dim_t
t0 = time.perf_counter()
done, last = 0, time.perf_counter()
while done < T - 1:
    n = min(48 * 73, T - 1 - done)
    sim.advance(n)
    done += n
    y = sim.stepper.get_state()
    assert np.isfinite(y).all(), f"non-finite state after row {done}"
    if time.perf_counter() - last > 20:
        print(f"rows {done}/{T - 1}  psi range [{y.min():.1f}, {y.max():.1f}]", flush=True); last = time.perf_counter()
wall = time.perf_counter() - t0
m = sim.moments()
mean_cm, std_cm = sim.wtd_mean_std(m)
c = sim.stepper.counters()
assert (m[0, 1:] == N).all()
print(f"N={N} D={cols.dim_d} rows={T - 1} wall={wall:.1f}s kernel={sim.kernel_ms / 1e3:.1f}s "
      f"column-days/s={N * (T - 1) / 48 / (sim.kernel_ms / 1e3):.0f}")
print("counters: jac_retry_passes", c["jac_retry"], "failed_attempts", c["failed_attempts"], "attempts abandoned by the iteration budget",
      c["guard_trips"], f"(failed attempts per member-year: {c['failed_attempts'] / N / years:.2f})")
print(f"wtd mean over the run: {np.nanmean(mean_cm[1:]):.1f} cm, ensemble sigma mean {np.nanmean(std_cm[1:]):.2f} cm, "
      f"max sigma {np.nanmax(std_cm[1:]):.2f} cm; final-row mean {mean_cm[-1]:.1f} sigma {std_cm[-1]:.2f}")
sim.close()
