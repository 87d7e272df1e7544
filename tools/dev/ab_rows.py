"""Row-by-row state and solver statistics of one build, for A/B comparison of two libraries:
    python tools/dev/ab_rows.py <lib.so> D N rows out.npz        (one process per library)
    python tools/dev/ab_rows.py --cmp a.npz b.npz                (first row / member where they part)
"""
import os, sys, pathlib
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
if sys.argv[1] == "--cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for r in range(a["psi"].shape[0]):
        ds = np.argwhere(a["stats"][r] != b["stats"][r])
        dp = np.abs(a["psi"][r] - b["psi"][r])
        if len(ds) or dp.max() > 0:
            who = np.flatnonzero(dp.max(axis=1) > 0)
            print(f"row {r + 1}: members {who[:40].tolist()}{' ...' if len(who) > 40 else ''}")
            if "--all" in sys.argv and r < a["psi"].shape[0] - 1:
                continue
            m = int(np.argmax(dp.max(axis=1)))
            print(f"row {r + 1}: {np.count_nonzero(dp.max(axis=1) > 0)} members differ, worst member {m} max |d psi| {dp.max():.3e} "
                  f"at node {int(np.argmax(dp[m]))}; stats a {a['stats'][r][m]} b {b['stats'][r][m]}")
            break
    else:
        print("identical states and statistics over", a["psi"].shape[0], "rows")
    sys.exit(0)
from hydromodel_amd import _lib
_lib.LIB_PATH = pathlib.Path(sys.argv[1]).resolve()
from hydromodel_amd.digest import ColumnTables, ForcingDigest
from hydromodel_amd.ensemble import EnsembleSimulation
from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
D, N, rows, out = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
params = default_parameters()
cols = ColumnTables(params, synthetic_well(D))
forcing = ForcingDigest(params, synthetic_forcing_frame(1), cols)
sim = EnsembleSimulation(cols, forcing, N, seed=1, member_offset=int(os.environ.get("AB_OFFSET", "0")))
st = sim.stepper
psi, stats = [], []
for r in range(1, rows + 1):
    o = st.step_rows(r, 1, want_stats=True)
    psi.append(st.get_state()); stats.append(o["stats"][0])
np.savez(out, psi=np.array(psi), stats=np.array(stats))
print(sys.argv[1], "done", np.array(psi).shape)
sim.close()
