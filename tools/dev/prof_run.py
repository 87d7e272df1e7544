"""Small fixed workload for rocprofv3 counter passes: N members x D, one launch of R rows."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np
from helpers import digest, golden
from hydromodel_amd.stepper import EnsembleStepper
well = int(sys.argv[1]) if len(sys.argv) > 1 else 300
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 48
_, cols, forcing = digest(well)
g = golden(f"g1_tables_{well}.npz")
st = EnsembleStepper(cols, forcing, N)
st.set_state(g["initial_cond"]); st.set_noise_philox(42, 0)
out = st.step_rows(1, rows, want_stats=False)
print("kernel_ms", out["kernel_ms"], "col-days/s", N * rows / 48 / (out["kernel_ms"] * 1e-3))
st.close()
