"""One fixed workload per kernel variant, for rocprofv3 passes (kernel trace, PMC): N members x D nodes, ONE launch of
`rows` rows from a cached spun-up initial condition, so that the profiled process launches the step kernel exactly once
(plus, with --calibrate, one skipped row before it: the launch that only loads and stores psi, for the FETCH/WRITE_SIZE
correction of MI355X_MICROARCH.md).

    python tools/prof_kernel.py D N [rows=48] [--generic] [--n 1.7 --lam 1.0] [--model vanGenuchten] [--calibrate]
                                [--ic gpurun_out/ic_cache.npz]
"""
import argparse, copy, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
from hydromodel_amd import _lib
if os.environ.get("HC_LIB"):          # a development build (tools/build_dev.py) instead of the shipped library
    import pathlib
    _lib.LIB_PATH = pathlib.Path(os.environ["HC_LIB"]).resolve()
    import ctypes
    _have = ctypes.CDLL(str(_lib.LIB_PATH))          # an older build lacks the newer entry points: bind what it has
    _lib.EXPORTS = {k: v for k, v in _lib.EXPORTS.items() if hasattr(_have, k)}
from hydromodel_amd.digest import ColumnTables, ForcingDigest
from hydromodel_amd.ensemble import PHILOX_DRAW_SPINUP, spinup_on_gpu
from hydromodel_amd.stepper import EnsembleStepper
from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
ap = argparse.ArgumentParser()
ap.add_argument("D", type=int); ap.add_argument("N", type=int); ap.add_argument("rows", type=int, nargs="?", default=48)
ap.add_argument("--generic", action="store_true"); ap.add_argument("--n", type=float, default=2.0)
ap.add_argument("--lam", type=float, default=1.0); ap.add_argument("--model", default="vrettas_fung")
ap.add_argument("--calibrate", action="store_true"); ap.add_argument("--ic", default="")
a = ap.parse_args()
params = default_parameters()
params["Soil_Properties"]["n"] = a.n
params["Hydraulic_Conductivity"]["Lambda_Exponent"] = a.lam
params["Hydrological_Model"]["Name"] = a.model
cols = ColumnTables(params, synthetic_well(a.D))
forcing = ForcingDigest(params, synthetic_forcing_frame(1), cols)
if a.calibrate:
    forcing = copy.copy(forcing)
    forcing.wtd_obs = forcing.wtd_obs.copy(); forcing.wtd_obs[1] = -1      # row 1 is skipped: load + store only
key = f"D{a.D}_n{a.n}_l{a.lam}_{a.model}"
ic = None
if a.ic and os.path.exists(a.ic):
    z = np.load(a.ic)
    ic = z[key] if key in z.files else None
if ic is None:
    probe = EnsembleStepper(cols, forcing, 1); probe.set_noise_philox(1, 0)
    n0 = probe.philox_normals(0, PHILOX_DRAW_SPINUP); probe.close()
    ic, _, _ = spinup_on_gpu(cols, forcing, n0)
    if a.ic:
        old = dict(np.load(a.ic)) if os.path.exists(a.ic) else {}
        old[key] = ic
        os.makedirs(os.path.dirname(a.ic) or ".", exist_ok=True)
        np.savez(a.ic, **old)
        print("initial condition cached; run again under the profiler"); sys.exit(0)
st = EnsembleStepper(cols, forcing, a.N)
if a.generic:
    st.set_generic_exponents(True)
st.set_state(ic); st.set_noise_philox(42, 0)
first = 1
if a.calibrate:
    c = st.step_rows(1, 1); first = 2
    print("calibration launch (skipped row):", round(c["kernel_ms"], 3), "ms; state bytes", a.N * cols.dim_d * 8)
out = st.step_rows(first, a.rows)
print(f"D={a.D} N={a.N} rows={a.rows} generic={a.generic} n={a.n} lam={a.lam} {a.model}: kernel_ms {out['kernel_ms']:.2f} "
      f"launches {out['launches']} col-days/s {a.N * a.rows / 48 / (out['kernel_ms'] * 1e-3):.0f} "
      f"algorithmic GB/s {a.N * a.rows * (16 * cols.dim_d + 16) / (out['kernel_ms'] * 1e-3) / 1e9:.2f}")
st.close()
