"""dev: dump RHS outputs of a development build -> npz.  python tools/dev/rhs_dump.py <lib.so> <out.npz>"""
import os, sys, pathlib
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
from hydromodel_amd import _lib
_lib.LIB_PATH = pathlib.Path(sys.argv[1]).resolve()
import numpy as np
from helpers import digest, golden
from hydromodel_amd.stepper import EnsembleStepper
_, cols, forcing = digest(300)
g = golden("g1_tables_300.npz")
N = 8
rng = np.random.default_rng(0)
st = EnsembleStepper(cols, forcing, N)
psi = g["initial_cond"][None, :] + 0.3 * rng.standard_normal((N, cols.dim_d))
st.set_state(psi); st.set_noise_host(rng.standard_normal((N, cols.dim_d)))
out = {}
for row in (2, 24):
    f, aux = st.rhs(row, want_aux=True)
    out[f"f{row}"] = f
    for k, v in aux.items():
        out[f"{k}{row}"] = v
np.savez(sys.argv[2], **out)
st.close()
