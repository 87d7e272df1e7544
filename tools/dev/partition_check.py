"""Does a build give the same bits for 50 rows in one launch and in 50 launches (Philox noise)?  python partition_check.py <lib.so> D"""
import os, sys, pathlib
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
from hydromodel_amd import _lib
_lib.LIB_PATH = pathlib.Path(sys.argv[1]).resolve()
from hydromodel_amd.digest import ColumnTables, ForcingDigest
from hydromodel_amd.stepper import EnsembleStepper
from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
D = int(sys.argv[2])
params = default_parameters()
cols = ColumnTables(params, synthetic_well(D))
forcing = ForcingDigest(params, synthetic_forcing_frame(1), cols)
N, rows = 6, 50
rng = np.random.default_rng(D)
y0 = np.tile(cols.z - 300.0, (N, 1)) + rng.standard_normal((N, D))
res = []
for step in (rows, 1):
    st = EnsembleStepper(cols, forcing, N)
    st.set_state(y0); st.set_noise_philox(77, 3)
    done = 0
    while done < rows:
        n = min(step, rows - done)
        st.step_rows(1 + done, n)
        done += n
    res.append(st.get_state()); st.close()
print(pathlib.Path(sys.argv[1]).name, D, "identical" if np.array_equal(*res) else f"DIFFERENT max {np.abs(res[0]-res[1]).max():.2e}")
