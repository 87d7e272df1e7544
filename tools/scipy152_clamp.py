"""How often do the select_initial_step clamps of scipy >= 1.9 bind on this path, and what does the reference's pinned
scipy==1.5.2 (no clamps) do differently?   python tools/scipy152_clamp.py [> profiles/r05_scipy152_clamp.txt]

CPU only (the C oracle, test infrastructure).  (i) every recorded row of the G5 fixtures (the reference's own year-long
runs on wells 1 and 200: start state, noise vector, end state): clamp counters, and the 1.5.2 form against the >= 1.9 form
and against the reference's recorded end state (made with scipy 1.15.3).  (ii) one oracle year on well 200 both ways.
"""
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
from helpers import digest, golden  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402


def row_of(forcing, i):
    return Oracle.row(forcing.precip[i], forcing.atm[i], forcing.daylight[i], forcing.wtd_obs[i])


def main():
    print("# select_initial_step: scipy 1.5.2 (pinned by the reference) has no clamp to the interval; scipy >= 1.9 (1.15.3 made the")
    print("# golden vectors; the kernel and the oracle follow it) clamps h0 and the returned step to |t_bound - t0| = 1.")
    for well in (1, 200):
        _, cols, forcing = digest(well)
        g = golden(f"g5_traj_{well}.npz")
        orc = Oracle(cols, forcing.surface_evap)
        rows = [(k, int(i)) for k, i in enumerate(g["rec_rows"]) if i >= 1]
        res = {}
        for mode in (0, 1):
            Oracle.set_scipy_152(mode)
            Oracle.clamp_counts(reset=True)
            ys, st = [], []
            for k, i in rows:
                y, s, _, _ = orc.solve_row(row_of(forcing, i), i - 1, i, g["rec_y0"][k], g["rec_nrnd_in"][k].copy())
                ys.append(y)
                st.append((s["nfev"], s["njev"], s["nlu"], s["nsteps"], s["attempts"]))
            res[mode] = (np.array(ys), st, Oracle.clamp_counts(reset=True))
        Oracle.set_scipy_152(0)
        ref = np.array([g["rec_y1"][k] for k, _ in rows])
        c0, c1 = res[0][2], res[1][2]
        rel = lambda a, b: np.max(np.abs(a - b) / (1 + np.abs(b)), axis=1)
        d_forms = rel(res[1][0], res[0][0])
        e_new, e_old = rel(res[0][0], ref), rel(res[1][0], ref)
        same_stats = sum(a == b for a, b in zip(res[0][1], res[1][1]))
        print(f"\nwell {well} (D = {cols.dim_d}): {len(rows)} recorded rows of the reference's year, {c0[2]} solves (attempts)")
        print(f"  h0 > interval (first clamp would bind):          {c0[0]} solves")
        print(f"  min(100 h0, h1) > interval (second clamp binds): {c0[1]} solves = {100.0 * c0[1] / max(c0[2], 1):.1f} %")
        print(f"  1.5.2 form against >= 1.9 form: identical solver statistics on {same_stats} of {len(rows)} rows; "
              f"max rel. state difference {d_forms.max():.2e}, median {np.median(d_forms):.2e}, rows above 1e-9: {(d_forms > 1e-9).sum()}")
        print(f"  against the reference's recorded end states: >= 1.9 form median {np.median(e_new):.2e} / max {e_new.max():.2e}; "
              f"1.5.2 form median {np.median(e_old):.2e} / max {e_old.max():.2e}")
    # (ii) an oracle year both ways
    _, cols, forcing = digest(200)
    g = golden("g5_traj_200.npz")
    rng = np.random.default_rng(11)
    base = rng.standard_normal(cols.dim_d)
    fresh = rng.standard_normal((int(forcing.refresh.sum()), cols.dim_d))
    out = {}
    for mode in (0, 1):
        Oracle.set_scipy_152(mode)
        Oracle.clamp_counts(reset=True)
        orc = Oracle(cols, forcing.surface_evap)
        r = orc.run(forcing, g["initial_cond"], base.copy(), fresh.copy(), 1, forcing.dim_t)
        out[mode] = (r["wtd_est"], Oracle.clamp_counts(reset=True))
    Oracle.set_scipy_152(0)
    w0, w1 = out[0][0][1:], out[1][0][1:]
    c = out[0][1]
    print(f"\none oracle year (well 200, {forcing.dim_t - 1} rows, same noise both ways): {c[2]} solves, h0 clamp {c[0]}, "
          f"step clamp {c[1]} ({100.0 * c[1] / max(c[2], 1):.1f} %)")
    print(f"  water-table index of the two forms equal on {100.0 * (w0 == w1).mean():.2f} % of the rows, never more than "
          f"{int(np.abs(w0 - w1).max())} cell(s) apart; mean depth {5.0 * w0.mean():.2f} / {5.0 * w1.mean():.2f} cm")


if __name__ == "__main__":
    main()
