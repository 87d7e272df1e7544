"""rocprofv3 target: one RHS evaluation per member (night row then day row)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
from helpers import digest, golden
from hydromodel_amd.stepper import EnsembleStepper
well = int(sys.argv[1]) if len(sys.argv) > 1 else 300
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
_, cols, forcing = digest(well)
g = golden(f"g1_tables_{well}.npz")
st = EnsembleStepper(cols, forcing, N)
st.set_state(g["initial_cond"]); import numpy as np; st.set_noise_host(np.random.default_rng(0).standard_normal((N, cols.dim_d)))
st.rhs(2); st.rhs(24)
st.close()
