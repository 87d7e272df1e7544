"""Does a build give the same bits for R rows in one launch and in R launches (Philox noise)?
    python partition_check.py <lib.so> D [special|generic|predict] [rows]"""
import os, sys, pathlib, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
from hydromodel_amd import _lib
_lib.LIB_PATH = pathlib.Path(sys.argv[1]).resolve()
from hydromodel_amd.digest import ColumnTables, ForcingDigest
from hydromodel_amd.stepper import EnsembleStepper
from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
D = int(sys.argv[2]); build = sys.argv[3] if len(sys.argv) > 3 else "special"; rows = int(sys.argv[4]) if len(sys.argv) > 4 else 50
params = default_parameters()
params["Simulation_Flags"]["PREDICT"] = build == "predict"
cols = ColumnTables(params, synthetic_well(D))
forcing = ForcingDigest(params, synthetic_forcing_frame(1), cols)
N = 6
rng = np.random.default_rng(D)
y0 = np.tile(cols.z - 300.0, (N, 1)) + rng.standard_normal((N, D))
res = []
for step in (1, rows):
    st = EnsembleStepper(cols, forcing, N)
    st.set_iteration_budget(3000)          # bounds a run that goes wrong
    if build == "generic":
        st.set_generic_exponents(True)
    st.set_state(y0)
    if os.environ.get("PC_HOST_NOISE"):
        st.set_noise_host(np.random.default_rng(5).standard_normal((N, D)))
    else:
        st.set_noise_philox(77, 3)
    done = 0; t0 = time.perf_counter()
    while done < rows:
        n = min(step, rows - done)
        nf = st.n_refresh(1 + done, n)
        st.step_rows(1 + done, n, fresh_noise=np.random.default_rng(9 + done).standard_normal((nf, N, D)) if os.environ.get("PC_HOST_NOISE") else None)
        done += n
    print(f"  step {step}: {time.perf_counter() - t0:.2f} s, counters {st.counters()}", flush=True)
    res.append(st.get_state()); st.close()
print(pathlib.Path(sys.argv[1]).name, D, build, "identical" if np.array_equal(*res) else f"DIFFERENT max {np.abs(res[0]-res[1]).max():.2e}")
