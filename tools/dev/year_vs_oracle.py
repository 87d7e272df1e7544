"""dev: one Philox member for a whole year on the GPU vs the CPU oracle fed the same normals."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from hydromodel_amd.digest import ColumnTables, ForcingDigest
from hydromodel_amd.ensemble import EnsembleSimulation
from hydromodel_amd.stepper import EnsembleStepper
from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
from oracle.oracle import Oracle
D = int(sys.argv[1]) if len(sys.argv) > 1 else 300
members = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 5, 1000, 4000]
params = default_parameters()
cols = ColumnTables(params, synthetic_well(D))
forcing = ForcingDigest(params, synthetic_forcing_frame(1), cols)
T = forcing.dim_t
psi0 = EnsembleSimulation(cols, forcing, 1, seed=17).psi0
draws = np.cumsum(forcing.refresh)
ref_rows = np.flatnonzero(forcing.refresh[1:]) + 1

def one(k):
    st = EnsembleStepper(cols, forcing, 1)
    st.set_state(psi0); st.set_noise_philox(17, k)
    base = st.philox_normals(k, 0)
    fresh = np.array([st.philox_normals(k, int(draws[r])) for r in ref_rows])
    out = st.step_rows(1, T - 1, want_wtd=True, want_stats=True)
    c = st.counters(); st.close()
    o = Oracle(cols, forcing.surface_evap)
    t0 = time.perf_counter()
    r = o.run(forcing, psi0, base, fresh, 1, T, want_stats=True)
    wg, wo = out["wtd"][:, 0], r["wtd_est"][1:]
    return (k, wg.mean() * cols.dz, wo.mean() * cols.dz, (wg == wo).mean(), np.abs(wg - wo).max(),
            int((out["stats"][:, 0, 4] > 1).sum()), int((r["per_row"][1:, 4] > 1).sum()), c["guard_trips"], time.perf_counter() - t0)

with ThreadPoolExecutor(max_workers=len(members)) as ex:
    for res in ex.map(one, members):
        print("member %d: mean wtd GPU %.1f cm oracle %.1f cm, index equal on %.1f %% of rows, max |diff| %d cells; rows with retries GPU %d oracle %d; budget trips %d; oracle %.0f s"
              % (res[0], res[1], res[2], 100 * res[3], res[4], res[5], res[6], res[7], res[8]), flush=True)
