"""dev: per-member spin-up (hc_spinup) on a deep column (no noise vector in LDS) vs the host-driven loop."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
from hydromodel_amd.digest import ColumnTables, ForcingDigest
from hydromodel_amd.ensemble import spinup_members_on_gpu, spinup_on_gpu
from hydromodel_amd.stepper import EnsembleStepper
from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
for D in (401, 581):
    params = default_parameters()
    cols = ColumnTables(params, synthetic_well(D))
    forcing = ForcingDigest(params, synthetic_forcing_frame(1), cols)
    noise = np.random.default_rng(D).standard_normal((3, D))
    st = EnsembleStepper(cols, forcing, 3)
    st.set_noise_host(noise)
    psi0, iters = spinup_members_on_gpu(st, cols, forcing)
    st.close()
    ic, it, early = spinup_on_gpu(cols, forcing, noise[1])
    print(D, "iters", iters.tolist(), "host loop", it, early, "max diff", np.max(np.abs(ic - psi0[1])))
    ph = EnsembleStepper(cols, forcing, 4)
    ph.set_noise_philox(9, 0)
    p0, i0 = spinup_members_on_gpu(ph, cols, forcing)
    print(D, "philox iters", i0.tolist(), "finite", np.isfinite(p0).all(), ph.counters())
    ph.close()
