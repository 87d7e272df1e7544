"""Diagnostic build (python tools/build_dev.py prof -DHC_PROFILE --cpl 5): time R back-to-back RHS evaluations per member at the
step kernel's occupancy.  python tools/prof_rhs_loop.py <lib.so> [N] [reps] [row]"""
import os, sys, pathlib, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
from hydromodel_amd import _lib
_lib.LIB_PATH = pathlib.Path(sys.argv[1]).resolve()
import numpy as np
from helpers import digest, golden
from hydromodel_amd.stepper import EnsembleStepper
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
rows = [int(sys.argv[4])] if len(sys.argv) > 4 else [2, 24]
_, cols, forcing = digest(300)
g = golden("g1_tables_300.npz")
st = EnsembleStepper(cols, forcing, N)
st.set_state(g["initial_cond"]); st.set_noise_host(np.random.default_rng(0).standard_normal((N, cols.dim_d)))
for row in rows:
    os.environ["HYDROCOL_RHS_REPEAT"] = "1"
    st.rhs(row)
    t0 = time.perf_counter(); st.rhs(row); t1 = time.perf_counter()
    os.environ["HYDROCOL_RHS_REPEAT"] = str(reps)
    t2 = time.perf_counter(); f = st.rhs(row); t3 = time.perf_counter()
    dt = (t3 - t2) - (t1 - t0)
    per_eval = dt / (reps - 1) / (N / 1024)          # seconds per evaluation per wave slot (1024 SIMDs)
    print(sys.argv[1], "row", row, "daylight", int(forcing.daylight[row]), "cycles/eval @2.4GHz", round(per_eval * 2.4e9),
          "quads", round(per_eval * 2.4e9 / 4), "sum f", float(np.abs(f).sum()))
st.close()
