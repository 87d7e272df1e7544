"""Round 5: the integrator's ``select_initial_step`` in the form of the reference's PINNED scipy (1.5.2: no clamp of h0 / the
returned step to the interval, /root/reference/requirements.txt:4) next to the default (scipy >= 1.9, the version that made
the pinning vectors).  ``hc_set_scipy_152`` against the oracle's ``ho_set_scipy_152`` on the rows the reference recorded
inside its own year-long runs; profiles/r05_scipy152_clamp.txt says how often the clamps bind (h0: 45 % of the solves)."""
import numpy as np
import pytest

from helpers import digest, golden, rel_err

pytestmark = pytest.mark.gpu


def _row(forcing, i):
    from oracle.oracle import Oracle
    return Oracle.row(forcing.precip[i], forcing.atm[i], forcing.daylight[i], forcing.wtd_obs[i])


@pytest.mark.parametrize("well", [1, 200])
def test_scipy_152_initial_step_switch_matches_the_oracle_switch(well):
    from hydromodel_amd.stepper import EnsembleStepper
    from oracle.oracle import Oracle
    _, cols, forcing = digest(well)
    g = golden(f"g5_traj_{well}.npz")
    orc = Oracle(cols, forcing.surface_evap)
    st_new, st_old = EnsembleStepper(cols, forcing, 1), EnsembleStepper(cols, forcing, 1)
    st_old.set_scipy_152(True)
    rows = [(k, int(i)) for k, i in enumerate(g["rec_rows"][:110]) if i >= 1]
    same_stats = differ = 0
    errs_old, errs_new = [], []
    try:
        for k, i in rows:
            y0, nin = g["rec_y0"][k], g["rec_nrnd_in"][k]
            fresh = nin[None, None, :] if forcing.refresh[i] else np.zeros((0,))
            out = {}
            for name, st in (("new", st_new), ("old", st_old)):
                st.set_state(y0[None, :])
                st.set_noise_host(nin[None, :])
                o = st.step_rows(i, 1, fresh_noise=fresh, want_stats=True)
                out[name] = (st.get_state()[0], o["stats"][0, 0, :5].tolist())
            Oracle.set_scipy_152(True)
            y_o, s_o, _, _ = orc.solve_row(_row(forcing, i), i - 1, i, y0, nin.copy())
            Oracle.set_scipy_152(False)
            y_n, s_n, _, _ = orc.solve_row(_row(forcing, i), i - 1, i, y0, nin.copy())
            same_stats += out["old"][1] == [s_o[q] for q in ("nfev", "njev", "nlu", "nsteps", "attempts")]
            errs_old.append(rel_err(out["old"][0], y_o))
            errs_new.append(rel_err(out["new"][0], y_n))
            differ += rel_err(out["old"][0], out["new"][0]) > 1e-9
    finally:
        Oracle.set_scipy_152(False)
        st_new.close()
        st_old.close()
    errs_old, errs_new = np.array(errs_old), np.array(errs_new)
    print(f"[well {well}] {len(rows)} recorded rows: 1.5.2 form -- the oracle's statistics on {same_stats}, median / max difference to "
          f"the oracle {np.median(errs_old):.1e} / {errs_old.max():.1e} (default form: {np.median(errs_new):.1e} / {errs_new.max():.1e}); "
          f"the two forms differ on {differ} rows")
    assert same_stats >= 0.95 * len(rows)
    assert np.median(errs_old) < 1e-8 and np.quantile(errs_old, 0.95) < 1e-5 and errs_old.max() < 5e-2      # the one-row tiers
    assert differ >= 5                                                # the switch does something: the h0 clamp binds often


def test_automatic_launch_length_follows_the_noise_mode_and_leaves_the_results_alone():
    """Late round 5 (include/hydrocol.h, hc_set_rows_per_launch): with the in-kernel noise a request is cut into launches of ~1 M
    member-days (the tail behind a launch's slowest wavefront: profiles/r05_launch_length_policy.txt), with the caller's noise --
    every refreshed row of a launch stages members x D doubles on the device -- into ~64 k member-days as before.  The per-row
    moments and the end state do not depend on the cut."""
    from hydromodel_amd.stepper import EnsembleStepper
    _, cols, forcing = digest(1)
    ic = golden("g1_tables_1.npz")["initial_cond"]
    N, rows = 16384, 48 * 6                                   # 6 days: 1 M / 16 384 = 64 days, 64 k / 16 384 = 4 days per launch
    res = []
    for rpl in (0, 48):
        st = EnsembleStepper(cols, forcing, N)
        if rpl:
            st.set_rows_per_launch(rpl)
        st.set_state(ic)
        st.set_noise_philox(11, 0)
        out = st.step_rows(1, rows)
        res.append((out["launches"], st.moments()[:, :rows + 1].copy(), st.get_state(0, 16)))
        st.close()
    assert res[0][0] == 1 and res[1][0] == 6
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    # the caller's noise: 4 days per launch at this size -> 2 launches for the 6 days
    n_fresh = int(forcing.refresh[1:rows + 1].sum())
    rng = np.random.default_rng(5)
    M = 2048                                                  # (64 k / 2 048 = 32 days per launch: one launch; 16 384 members would
    st = EnsembleStepper(cols, forcing, M)                    #  need 16 384 x D x n_fresh doubles of host noise for nothing more)
    st.set_state(ic)
    st.set_noise_host(rng.standard_normal((M, cols.dim_d)))
    out = st.step_rows(1, 48 * 40, fresh_noise=rng.standard_normal((int(forcing.refresh[1:48 * 40 + 1].sum()), M, cols.dim_d)))
    st.close()
    assert out["launches"] == 2 and n_fresh >= 1              # 40 days at 32 days per launch (in-kernel noise: 365 days -> 1)
