"""How often does the PREDICT-mode solver run into the iteration budget from a rough start?  python predict_cost.py D [budget]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
from hydromodel_amd.digest import ColumnTables, ForcingDigest
from hydromodel_amd.stepper import EnsembleStepper
from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
D = int(sys.argv[1]); budget = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
for predict in (False, True):
    params = default_parameters()
    params["Simulation_Flags"]["PREDICT"] = predict
    cols = ColumnTables(params, synthetic_well(D))
    forcing = ForcingDigest(params, synthetic_forcing_frame(1), cols)
    N = 6
    rng = np.random.default_rng(D)
    y0 = np.tile(cols.z - 300.0, (N, 1)) + rng.standard_normal((N, D))
    st = EnsembleStepper(cols, forcing, N)
    st.set_iteration_budget(budget)
    st.set_state(y0); st.set_noise_philox(77, 3)
    for r in range(6):
        t0 = time.perf_counter()
        o = st.step_rows(1 + r, 1, want_stats=True)
        print(f"D={D} predict={predict} row {1 + r}: {time.perf_counter() - t0:.3f} s, nfev {o['stats'][0, :, 0].tolist()}, "
              f"attempts {o['stats'][0, :, 4].tolist()}, counters {st.counters()['guard_trips']}", flush=True)
    st.close()
