import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np
from helpers import digest, golden
from hydromodel_amd.stepper import EnsembleStepper
from oracle.oracle import Oracle
well = int(sys.argv[1]) if len(sys.argv) > 1 else 1
_, cols, forcing = digest(well)
g = golden(f"g5_traj_{well}.npz")
D, T = cols.dim_d, forcing.dim_t
rng = np.random.default_rng(np.random.SeedSequence(911)); rng.standard_normal(D)
base = rng.standard_normal(D)
nref = int(forcing.refresh.sum()); fresh = np.array([rng.standard_normal(D) for _ in range(nref)])
o = Oracle(cols, forcing.surface_evap)
r = o.run(forcing, g["initial_cond"], base, fresh, 1, T, want_stats=True)
st = EnsembleStepper(cols, forcing, 1)
st.set_state(g["initial_cond"]); st.set_noise_host(base[None])
out = st.step_rows(1, T - 1, fresh_noise=fresh[:, None, :], want_wtd=True, want_stats=True)
wg = out["wtd"][:, 0]; wo = r["wtd_est"][1:]; wr = np.rint(g["wtd_est_cm"] / 5).astype(int)[1:]
print("gpu==oracle", (wg == wo).mean(), "maxoff", np.abs(wg - wo).max())
print("gpu==ref   ", (wg == wr).mean(), "maxoff", np.abs(wg - wr).max())
print("oracle==ref", (wo == wr).mean(), "maxoff", np.abs(wo - wr).max())
sg = out["stats"][:, 0]; so = r["per_row"][1:]; sr = g["per_row_stats"][1:]
print("failed rows gpu", (sg[:, 4] > 1).sum(), "oracle", (so[:, 4] > 1).sum(), "ref", (sr[:, 4] > 1).sum())
print("mean nfev gpu", sg[:, 0].mean(), "oracle", so[:, 0].mean(), "ref", sr[:, 0].mean())
first = np.argmax(sg[:, 0] != so[:, 0]); print("first nfev mismatch gpu/oracle at row", first + 1)
print("mean wtd idx gpu", wg.mean(), "oracle", wo.mean(), "ref", wr.mean())
print("fail rows gpu", (np.where(sg[:, 4] > 1)[0] + 1).tolist(), [int(x) for x in sg[sg[:, 4] > 1, 4]])
print("fail rows ora", (np.where(so[:, 4] > 1)[0] + 1).tolist(), [int(x) for x in so[so[:, 4] > 1, 4]])
print("fail rows ref", (np.where(sr[:, 4] > 1)[0] + 1).tolist(), [int(x) for x in sr[sr[:, 4] > 1, 4]])
for lo in range(0, T - 1, 1460):
    hi = min(T - 1, lo + 1460)
    print(lo, "gpu==ora %.3f ora==ref %.3f  mean idx gpu %.2f ora %.2f ref %.2f" % ((wg[lo:hi] == wo[lo:hi]).mean(), (wo[lo:hi] == wr[lo:hi]).mean(), wg[lo:hi].mean(), wo[lo:hi].mean(), wr[lo:hi].mean()))
print("base noise norm gpu", np.linalg.norm(st.get_noise_base()[0]), "oracle", np.linalg.norm(r["base_noise"]), "initial", np.linalg.norm(base))
