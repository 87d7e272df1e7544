import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import pathlib
from hydromodel_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = pathlib.Path(sys.argv[1]).resolve()
from hydromodel_amd.digest import ColumnTables, ForcingDigest
from hydromodel_amd.ensemble import pressure_head
from hydromodel_amd.stepper import EnsembleStepper
from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
for D in (581,):
    params = default_parameters()
    cols = ColumnTables(params, synthetic_well(D))
    forcing = ForcingDigest(params, synthetic_forcing_frame(1), cols)
    y0, _ = pressure_head(cols, cols.por_raw)
    N = 3
    base = np.random.default_rng(4).standard_normal((N, D))
    st = EnsembleStepper(cols, forcing, N)
    st.set_iteration_budget(3)
    st.set_state(y0); st.set_noise_host(base)
    out = st.step_rows(1, 3, fresh_noise=np.zeros((0,)), want_stats=True)
    print(D, "attempts", out["stats"][:, :, 4].tolist(), "failed", out["failed"].tolist(), st.counters())
    print(np.abs(st.get_noise_base() / base).mean(axis=1), 0.8 ** 15)
    st.close()
