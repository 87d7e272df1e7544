"""dev: phase trace of the saved guard case on a diagnostic build vs the oracle's accepted steps."""
import os, sys, pathlib, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
os.environ["HYDROCOL_DEBUG_TRACE"] = "1"
from hydromodel_amd import _lib
_lib.LIB_PATH = pathlib.Path(sys.argv[1]).resolve()
import numpy as np
from hydromodel_amd.digest import ColumnTables, ForcingDigest
from hydromodel_amd.stepper import EnsembleStepper
from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
from oracle.oracle import Oracle
g = np.load(os.path.join(R, "tools", "dev", "guard_case2.npz"))
row = int(g["row"])
params = default_parameters()
cols = ColumnTables(params, synthetic_well(300))
forcing = ForcingDigest(params, synthetic_forcing_frame(1), cols)
h = EnsembleStepper(cols, forcing, 1)
h.set_state(g["y_before"]); h.set_noise_host(g["z"][None, :])
out = h.step_rows(row, 1, fresh_noise=np.zeros((0,)), want_stats=True)
print("stats", out["stats"][0, 0].tolist())
n = 1 + 6 * 20000
buf = np.zeros(n)
h.lib.hc_debug_trace.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int64]
assert h.lib.hc_debug_trace(h.h, buf.ctypes.data_as(C.POINTER(C.c_double)), n) == 0
k = int(buf[0]); tr = buf[1:1 + 6 * k].reshape(k, 6)
names = ["F0", "F1", "JAC", "JAC_REDO", "NEWTON", "JAC_FIN", "STEP_BEGIN", "STEP_TRY", "NEWTON_BEGIN", "NEWTON_FAIL", "ERR_TEST",
         "ACCEPT", "SUCCESS", "FAIL"]
o = Oracle(cols, forcing.surface_evap)
r = Oracle.row(forcing.precip[row], forcing.atm[row], forcing.daylight[row], forcing.wtd_obs[row])
yo, so, _, ts = o.solve_row(r, row - 1, row, g["y_before"], g["z"].copy())
print("oracle", so, "accepted t:", np.array2string(ts[:70] - (row - 1), precision=6, max_line_width=200))
print("trace entries", k)
# accepted times on the GPU: t changes between consecutive entries
tt = tr[:, 1] - (row - 1)
chg = np.flatnonzero(np.diff(tt) != 0)
print("GPU accepted t (first 80):", np.array2string(tt[chg + 1][:80], precision=6, max_line_width=200))
lim = int(sys.argv[2]) if len(sys.argv) > 2 else 400
for i in range(min(k, lim)):
    p, t, ha, od, fl, en = tr[i]
    print(f"{i:5d} {names[int(p)]:12s} t={t - (row - 1):.9f} h={ha:.6e} order={int(od)} n_eq={int(fl) % 100} cj={int(fl) // 100 % 10} lu={int(fl) // 1000 % 10} k={int(fl) // 10000} norm={en:.4e}")
h.close()
