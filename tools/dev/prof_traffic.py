"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE target.

Dispatch A (calibration): one SKIPPED row (wtd_obs < 0) -> the step kernel only loads and stores
psi: exactly N*D*8 bytes each way in the kernel's real access pattern.
Dispatch B: the benchmark launch shape, 48 rows.
"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np
from helpers import digest, golden
from hydromodel_amd.stepper import EnsembleStepper
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
_, cols, forcing = digest(300)
g = golden("g1_tables_300.npz")
import copy
fc = copy.copy(forcing)
fc.wtd_obs = forcing.wtd_obs.copy(); fc.wtd_obs[1] = -1
st = EnsembleStepper(cols, fc, N)
st.set_state(g["initial_cond"]); st.set_noise_philox(42, 0)
a = st.step_rows(1, 1)          # dispatch A: skipped row
b = st.step_rows(2, 48)         # dispatch B
print("N", N, "D", cols.dim_d, "state bytes", N * cols.dim_d * 8, "ms A", a["kernel_ms"], "ms B", b["kernel_ms"])
st.close()
