"""Split-column kernel (two waves per member) against the one-wave kernel of the same library and against the oracle.
    python tools/dev/pair_check.py [D=581] [rows=6] [N=6]
The one-wave run sets HYDROCOL_SPLIT_COLUMN=0 (read at hc_create)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
from hydromodel_amd.digest import ColumnTables, ForcingDigest
from hydromodel_amd.stepper import EnsembleStepper
from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
from oracle.oracle import Oracle
D = int(sys.argv[1]) if len(sys.argv) > 1 else 581
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 6
N = int(sys.argv[3]) if len(sys.argv) > 3 else 6
params = default_parameters()
cols = ColumnTables(params, synthetic_well(D))
forcing = ForcingDigest(params, synthetic_forcing_frame(1), cols)
rng = np.random.default_rng(D)
y0 = np.tile(cols.z - 300.0, (N, 1)) + rng.standard_normal((N, cols.dim_d))
base = rng.standard_normal((N, D))
first = 20                                              # daylight rows from 13 on
nf = int(forcing.refresh[first:first + rows].sum())
fresh = rng.standard_normal((nf, N, D))
res = {}
for mode in ("split", "one-wave"):
    if mode == "one-wave":
        os.environ["HYDROCOL_SPLIT_COLUMN"] = "0"
    else:
        os.environ.pop("HYDROCOL_SPLIT_COLUMN", None)
    st = EnsembleStepper(cols, forcing, N)
    st.set_state(y0); st.set_noise_host(base)
    t0 = time.time()
    out = st.step_rows(first, rows, fresh_noise=fresh, want_wtd=True, want_stats=True, want_psi=True)
    res[mode] = (out["psi"], out["wtd"], out["stats"], st.get_noise_base())
    print(mode, "kernel_ms", round(out["kernel_ms"], 2), "wall", round(time.time() - t0, 2), flush=True)
    st.close()
a, b = res["split"], res["one-wave"]
e = np.max(np.abs(a[0] - b[0]) / (1 + np.abs(b[0])), axis=2)           # [rows][members]
print("split vs one-wave: max rel diff per row", np.array2string(e.max(axis=1), precision=2),
      "wtd equal", bool(np.array_equal(a[1], b[1])), "stats equal rows", int((a[2] == b[2]).all(axis=(1, 2)).sum()), "of", rows)
o = Oracle(cols, forcing.surface_evap)
worst = 0.0
for k in range(min(N, 3)):
    r = o.run(forcing, y0[k], base[k], fresh[:, k, :], first, first + rows, want_psi=True, want_stats=True)
    want = r["psi_rows"][first:first + rows]
    eo = np.max(np.abs(a[0][:, k, :] - want) / (1 + np.abs(want)), axis=1)
    same = (a[2][:, k, :5] == r["per_row"][first:first + rows, :5]).all(axis=1).sum()
    print(f"member {k}: split vs oracle max rel diff per row {np.array2string(eo, precision=2)}; rows with the oracle's statistics {same}/{rows}")
    worst = max(worst, eo.max())
print("OK" if (e.max() < 1e-6 and worst < 1e-6) else "MISMATCH")
