#!/usr/bin/env python3
"""Why the CPU oracle and the reference disagree on the solver statistics of a handful of recorded rows.

Runs only in the build container (imports /root/reference through make_golden.py's stand-ins for numba / h5py).

Of the rows `g5_traj_{1,200}.npz` recorded with full input state, the oracle reproduces the reference's (nfev, njev, nlu,
attempts) on all but a few.  DESIGN.md attributed those to the linear solver: SciPy's BDF factors `I - c J` with SuperLU
(`scipy/integrate/_ivp/bdf.py`: `splu(A)`, column order COLAMD, threshold pivoting), the oracle by Gaussian elimination of
the tridiagonal matrix in natural order with partial pivoting.  Both are backward-stable; their solutions differ in the last
bits; Newton's convergence test and the step controller compare norms against thresholds, and on a stiff row one of a few
hundred such tests can fall the other way.

This script tests that attribution with the REFERENCE ITSELF: it replays the disagreeing rows (and a control sample of
agreeing ones) through `RichardsPDE.solve` twice -- once as shipped, once with `bdf.splu` replaced by
`splu(A, permc_spec="NATURAL", diag_pivot_thresh=1.0)` (natural column order, classical partial pivoting: the oracle's
elimination) -- and compares the statistics with the recorded ones and with the oracle's.

    python tests/golden/superlu_order_check.py [--wells 1 200] [--control 60]

Finding (2026-10, scipy 1.15.3): see tests/golden/README.md, "SuperLU".
"""
import argparse
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import make_golden as mg  # noqa: E402  (installs the stand-ins and puts the reference on sys.path)
import scipy.integrate._ivp.bdf as bdf  # noqa: E402
from scipy.sparse.linalg import splu  # noqa: E402

sys.path.insert(0, str(mg.REPO / "tests"))
from helpers import digest  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402


def natural_splu(A):
    return splu(A, permc_spec="NATURAL", diag_pivot_thresh=1.0)


def reference_stats(sim, i, y0, n_rnd, natural):
    """(nfev, njev, nlu, attempts), state of the reference's own solve of row i."""
    m = sim.mData
    z = m["z_grid"]
    wtd_i = np.where(z == m["zWtd_cm"][i])[0][0]
    args_i = {"wtd": wtd_i, "n_rnd": n_rnd.copy(), "atm": m["atm"][i], "time": m["time"][i],
              "interception": m["interception"], "precipitation": m["precipitation_cm"][i]}
    old = bdf.splu
    if natural:
        bdf.splu = natural_splu
    try:
        with mg._SolveRecorder() as rec:
            out = sys.stdout
            sys.stdout = open(os.devnull, "w")
            try:
                y1 = sim.pde_model.solve((i - 1, i), y0.copy(), args_i)
            finally:
                sys.stdout.close()
                sys.stdout = out
    finally:
        bdf.splu = old
    c = rec.calls
    return (sum(x[0] for x in c), sum(x[1] for x in c), sum(x[2] for x in c), len(c)), np.array(y1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--wells", type=int, nargs="*", default=[1, 200])
    ap.add_argument("--control", type=int, default=60, help="agreeing rows replayed as a control, per well")
    a = ap.parse_args()
    tmp = tempfile.mkdtemp(prefix="hm_slu_")
    total = dict(rows=0, differ=0, explained=0, control=0, control_changed=0, shipped_reproduces=0)
    for well in a.wells:
        g = np.load(HERE / f"g5_traj_{well}.npz")
        _, cols, forcing = digest(well)
        orc = Oracle(cols, forcing.surface_evap)
        sim, _, _ = mg._setup(well, tmp)
        rows, stats = g["rec_rows"], g["per_row_stats"]
        differ, agree = [], []
        for k, i in enumerate(rows):
            r = Oracle.row(forcing.precip[i], forcing.atm[i], forcing.daylight[i], forcing.wtd_obs[i])
            _, st, _, _ = orc.solve_row(r, i - 1, i, g["rec_y0"][k], g["rec_nrnd_in"][k])
            mine = (st["nfev"], st["njev"], st["nlu"], st["attempts"])
            ref = tuple(int(v) for v in stats[i, [0, 1, 2, 4]])
            (agree if mine == ref else differ).append((k, int(i), mine, ref))
        total["rows"] += len(rows)
        total["differ"] += len(differ)
        print(f"well {well}: {len(rows)} recorded rows, oracle statistics differ from the reference's on {len(differ)}")
        for k, i, mine, ref in differ:
            shipped, y_s = reference_stats(sim, i, g["rec_y0"][k], g["rec_nrnd_in"][k], natural=False)
            nat, y_n = reference_stats(sim, i, g["rec_y0"][k], g["rec_nrnd_in"][k], natural=True)
            total["shipped_reproduces"] += shipped == ref
            total["explained"] += nat == mine
            dy = float(np.max(np.abs(y_n - y_s) / (1.0 + np.abs(y_s))))
            print(f"  row {i:6d}: recorded {ref}  reference now {shipped}  reference with natural-order LU {nat}  oracle {mine}"
                  f"  -> {'EXPLAINED' if nat == mine else 'not the ordering alone'}; states of the two reference solves differ by {dy:.1e}")
        rng = np.random.default_rng(5)
        pick = rng.choice(len(agree), size=min(a.control, len(agree)), replace=False)
        changed = 0
        for j in pick:
            k, i, mine, ref = agree[j]
            nat, _ = reference_stats(sim, i, g["rec_y0"][k], g["rec_nrnd_in"][k], natural=True)
            changed += nat != ref
        total["control"] += len(pick)
        total["control_changed"] += changed
        print(f"  control: {len(pick)} agreeing rows replayed with the natural-order LU: statistics changed on {changed}")
    print("SUMMARY", total)


if __name__ == "__main__":
    main()
