import os, sys, pathlib
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
from hydromodel_amd import _lib
_lib.LIB_PATH = pathlib.Path(sys.argv[1]).resolve()
import ctypes
_have = ctypes.CDLL(str(_lib.LIB_PATH))
_lib.EXPORTS = {k: v for k, v in _lib.EXPORTS.items() if hasattr(_have, k)}
OLD = "hc_add_point" not in _lib.EXPORTS
from hydromodel_amd.digest import ColumnTables, ForcingDigest
from hydromodel_amd.ensemble import pressure_head
from hydromodel_amd.stepper import EnsembleStepper
from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
from oracle.oracle import Oracle
D = int(sys.argv[2]); rows = int(sys.argv[3])
params = default_parameters()
cols = ColumnTables(params, synthetic_well(D))
forcing = ForcingDigest(params, synthetic_forcing_frame(1), cols)
if OLD:
    forcing.wet_season = forcing.wet_season * 0
rng = np.random.default_rng(D)
N = 3
y0 = np.tile(cols.z - 300.0, (N, 1)) + rng.standard_normal((N, D))
base = rng.standard_normal((N, D))
o = Oracle(cols, forcing.surface_evap)
for chunked in (False, True):
    st = EnsembleStepper(cols, forcing, N)
    st.set_state(y0); st.set_noise_host(base)
    nf = st.n_refresh(1, rows)
    fresh = rng.standard_normal((nf, N, D))
    if chunked:
        outs = [st.step_rows(1 + r, 1, fresh_noise=fresh[:0], want_psi=True)["psi"] for r in range(rows)]
        psi = np.concatenate(outs)
    else:
        psi = st.step_rows(1, rows, fresh_noise=fresh, want_psi=True)["psi"]
    st.close()
    for k in range(N):
        ref = o.run(forcing, y0[k], base[k], fresh[:, k, :], 1, 1 + rows, want_psi=True)["psi_rows"][1:1 + rows]
        e = np.max(np.abs(psi[:, k] - ref) / (1 + np.abs(ref)), axis=1)
        print(pathlib.Path(sys.argv[1]).name, "chunked" if chunked else "one launch", "member", k, " ".join(f"{x:.1e}" for x in e))
# ---- Philox mode: the oracle fed with the normals the kernel generates
st = EnsembleStepper(cols, forcing, N)
st.set_state(y0); st.set_noise_philox(77, 0)
base_p = np.stack([st.philox_normals(k, 0) for k in range(N)])
ref_rows = [i for i in range(1, 1 + rows) if forcing.refresh[i]]
fresh_p = np.stack([[st.philox_normals(k, j + 1) for k in range(N)] for j in range(int(forcing.refresh[1:ref_rows[-1] + 1].sum()))]) if ref_rows else np.zeros((0, N, D))
psi = st.step_rows(1, rows, want_psi=True)["psi"]
st.close()
for k in range(N):
    ref = o.run(forcing, y0[k], base_p[k], fresh_p[:, k, :], 1, 1 + rows, want_psi=True)["psi_rows"][1:1 + rows]
    e = np.max(np.abs(psi[:, k] - ref) / (1 + np.abs(ref)), axis=1)
    print(pathlib.Path(sys.argv[1]).name, "philox one launch", "member", k, " ".join(f"{x:.1e}" for x in e))
