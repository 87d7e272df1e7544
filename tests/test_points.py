"""Non-default parameter points (BASELINE config 5): digest and oracle pinned by the reference where it can run.

The reference runs any (a0, psi_sat, lambda, sigma, K) at n = 2; fixtures g1p/g2p/g34p (tests/golden/points.json,
``make_golden.py points``) hold its tables, plugin calls, RHS and one-row solves at three such points.  For n that is
not an even integer the reference cannot run at all -- ``porosity.py:172-181`` raises the signed ``alpha * psi`` to the
power n, the profiles go complex and the first RHS evaluation raises at ``richards_pde.py:119`` -- so that axis is an
extension with declared semantics (retention curve on |psi|, the form ``vrettas_fung.py:115`` uses) and no oracle.
"""
import warnings

import numpy as np
import pytest

from helpers import WELLS, digest_point, golden, points, rel_err
from hydromodel_amd import digest as dg
from hydromodel_amd.synthetic import default_parameters
from oracle.oracle import Oracle, VIEW_FIRST, VIEW_INTERIOR, VIEW_NODES, VIEW_TOP

TAGS = sorted(points())
POINT_TOL = 1e-11


@pytest.mark.parametrize("tag", TAGS)
def test_tables_at_the_point_match_the_reference_bit_for_bit(tag):
    _, cols, forcing = digest_point(tag)
    g = golden(f"g1p_tables_{tag}.npz")
    for mine, ref in ((cols.por_raw, "por_node"), (cols.fc_raw, "fc_node"), (cols.wlt_raw, "wlt_node"),
                      (cols.por_mid, "por_mid"), (cols.fc_mid, "fc_mid"), (cols.wlt_mid, "wlt_mid"),
                      (cols.meank_node, "meank_node"), (cols.meank_mid, "meank_mid")):
        assert np.array_equal(mine, g[ref]), ref
    assert cols.ipsi50 == float(g["iPsi_50"])            # depends on lambda and a0 (simulation.py:336-339)
    assert np.array_equal(forcing.atm, g["atm"])
    assert forcing.surface_evap == float(g["surface_evap"])
    # the points really differ from the default one
    d = golden("g1_tables_200.npz")
    assert not np.array_equal(g["wlt_node"], d["wlt_node"]) or not np.array_equal(g["meank_node"], d["meank_node"])


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("model,key", [("vrettas_fung", "vf"), ("vanGenuchten", "vg")])
def test_plugin_call_at_the_point_matches_the_reference(tag, model, key):
    _, cols, forcing = digest_point(tag, model)
    o = Oracle(cols, forcing.surface_evap)
    g = golden(f"g2p_pointwise_{tag}.npz")
    nr = g["n_rnd"]
    for name in ("sweep", "ic", "moist", "dry"):
        psi = g[f"psi_{name}"]
        q, K, C, kb, qi = o.model_eval(VIEW_NODES, psi, nr)
        for k_, v in zip(("q", "K", "C", "kbkg"), (q, K, C, kb)):
            assert rel_err(v, g[f"{key}_{name}_node_{k_}"]) < POINT_TOL, (name, k_)
        assert abs(qi - float(g[f"{key}_{name}_node_qinf"])) < POINT_TOL
        ym = 0.5 * (psi[1:-1] + psi[2:])
        q, K, C, kb, _ = o.model_eval(VIEW_INTERIOR, ym, nr)
        for k_, v in zip(("q", "K", "C", "kbkg"), (q, K, C, kb)):
            assert rel_err(v, g[f"{key}_{name}_mid_{k_}"]) < POINT_TOL, (name, k_)
        q, K, C, kb, _ = o.model_eval(VIEW_FIRST, [0.5 * (psi[0] + psi[1])], nr)
        assert rel_err([q[0], K[0], C[0], kb[0]], g[f"{key}_{name}_first"]) < POINT_TOL
        q, K, C, kb, qi = o.model_eval(VIEW_TOP, [psi[0]], nr)
        assert rel_err([q[0], K[0], C[0], kb[0], qi], g[f"{key}_{name}_top"]) < POINT_TOL
    for name in ("porosity", "half", "res", "rand"):
        psi, s = o.pressure_head(g[f"ph_{name}_theta"])
        assert rel_err(psi, g[f"ph_{name}_psi"]) < 1e-12
        assert rel_err(s, g[f"ph_{name}_seff"]) < 1e-14


def _case(g, name, tag):
    fl = g[f"{name}_flags"]
    _, cols, forcing = digest_point(tag)
    o = Oracle(cols, forcing.surface_evap, flags={"ET": bool(fl[1]), "LF": bool(fl[2]), "HLIFT": bool(fl[3])})
    hour = int(g[f"{name}_hour"])
    row = Oracle.row(g[f"{name}_precip"], g[f"{name}_atm"], 6 <= hour <= 17, int(g["wtd_idx"]), spinup=bool(fl[0]))
    return o, row


@pytest.mark.parametrize("tag", TAGS)
def test_rhs_at_the_point_matches_the_reference(tag):
    g = golden(f"g34p_states_{tag}.npz")
    for name in g["names"]:
        o, row = _case(g, name, tag)
        dydt, aux = o.rhs(row, g[f"{name}_y"], g["n_rnd"], want_aux=True)
        assert rel_err(dydt, g[f"{name}_dydt"]) < 1e-11, name
        assert rel_err(aux["c"][1:], g[f"{name}_mid_c"], 1e-7) < 1e-11, name
        assert rel_err(aux["s"][1:], g[f"{name}_mid_s"]) < 1e-12, name
        assert rel_err(aux["f"][1:], g[f"{name}_mid_f"]) < 1e-11, name
        assert rel_err([aux["c"][0], aux["s"][0], aux["f"][0]], g[f"{name}_first_csf"]) < 1e-11, name
        assert abs(aux["pL"] - g[f"{name}_bc"][0]) < 1e-12, name


@pytest.mark.parametrize("tag", TAGS)
def test_single_row_solve_at_the_point_matches_the_reference(tag):
    g = golden(f"g34p_states_{tag}.npz")
    same = total = loose = 0
    for name in g["names"]:
        ref_stats = g[f"{name}_solve_stats"]
        if ref_stats.shape[0] != 1 or ref_stats[0, 0] > 300:    # very long solves (hlift, one saturated-top state):
            continue                                            # chaotic at rtol = 1e-3, see DESIGN.md §3
        o, row = _case(g, name, tag)
        y1, st, n_after, ts = o.solve_row(row, 7, 8, g[f"{name}_y"], g["n_rnd"], cap_steps=512)
        ry = g[f"{name}_solve_y"]
        err = np.max(np.abs(y1 - ry) / (1.0 + np.abs(ry)))
        # every re-evaluation of the FD Jacobian amplifies last-bit differences (h ~ 1.5e-8 |y|): rows that keep the
        # first Jacobian are held to 1e-9, the others to 2e-4 (measured: <= 1.4e-4 with identical statistics)
        regular = ref_stats[0, 1] <= 1
        total += 1
        same += [st["nfev"], st["njev"], st["nlu"], st["nsteps"]] == ref_stats[0, :4].tolist()
        loose += not regular
        assert err < (1e-9 if regular else 2e-4), (name, err, st, ref_stats)
        assert np.array_equal(n_after, g[f"{name}_solve_nrnd_after"])
    print(f"[{tag}] one-row solves: {same}/{total} with the reference's nfev/njev/nlu/steps, "
          f"{loose} with a refreshed Jacobian (2e-4 tier)")
    assert total >= 12 and same == total, (same, total)


@pytest.mark.parametrize("tag,model,fname", [("a03l13", "vrettas_fung", "g5sp_a03l13_200.npz"),
                                             ("s07l08", "vrettas_fung", "g5sp_s07l08_200.npz"),
                                             ("a003", "vanGenuchten", "g5sp_vg_a003_200.npz")])
def test_first_days_of_the_reference_run_at_the_point_replay(tag, model, fname):
    """First 240 rows of the reference's own year-long run at the point (lambda != 1), replayed row by row."""
    _, cols, forcing = digest_point(tag, model)
    o = Oracle(cols, forcing.surface_evap)
    g = golden(fname)
    assert g["rows"].tolist() == list(range(1, 241))
    errs, same = [], 0
    for k, i in enumerate(g["rows"]):
        row = Oracle.row(forcing.precip[i], forcing.atm[i], forcing.daylight[i], forcing.wtd_obs[i])
        y1, st, n_out, _ = o.solve_row(row, i - 1, i, g["y0"][k], g["nrnd_in"][k])
        ref = g["y1"][k]
        errs.append(np.max(np.abs(y1 - ref) / (1.0 + np.abs(ref))))
        same += [st["nfev"], st["njev"], st["nlu"], st["nsteps"], st["attempts"]] == g["stats"][k].tolist()
        assert np.array_equal(n_out, g["nrnd_out"][k])
    errs = np.array(errs)
    print(f"[{tag}] 240 reference rows: {same} with identical nfev/njev/nlu/steps/attempts, median error {np.median(errs):.1e}, "
          f"max {errs.max():.1e}")
    assert same >= 0.9 * len(errs), (same, len(errs))
    assert np.median(errs) < 1e-10 and np.quantile(errs, 0.9) < 1e-5 and errs.max() < 5e-2


# ------------------------------------------------------------------------------- the n axis
@pytest.mark.parametrize("n", [1.5, 1.7, 2.5, 3.0])
def test_exponents_the_reference_cannot_run_use_the_modulus_form(n):
    """fc / wilting are real, finite, ordered, and equal the retention curve evaluated on |psi|."""
    params = default_parameters()
    params["Soil_Properties"]["n"] = n
    with warnings.catch_warnings():
        warnings.simplefilter("error")               # a ComplexWarning (complex -> real cast) would fail here
        cols = dg.ColumnTables(params, WELLS[200])
    soil, theta = cols.soil, cols.theta
    for prof, psi in ((cols.fc_raw, theta.flc), (cols.wlt_raw, theta.wlt)):
        assert prof.dtype == np.float64 and np.all(np.isfinite(prof))
    m = 1.0 - 1.0 / n
    expect_w = theta.res + (cols.por_raw - theta.res) * (1.0 + (soil.alpha * abs(theta.wlt)) ** n) ** (-m)
    expect_f = theta.res + (cols.por_raw - theta.res) * (1.0 + (soil.alpha * abs(theta.flc)) ** n) ** (-m)
    assert np.allclose(cols.wlt_raw, np.minimum(expect_w, np.maximum(expect_f, theta.res)), rtol=1e-14, atol=0)
    assert np.allclose(cols.fc_raw, np.maximum(expect_f, theta.res), rtol=1e-14, atol=0)
    assert np.all(cols.wlt_raw >= theta.res) and np.all(cols.wlt_raw <= cols.fc_raw)
    assert np.all(cols.fc_raw <= cols.por_raw)


def test_even_integer_exponents_keep_the_reference_expression():
    """n = 2 (pinned bit for bit by G1) and n = 4 go through the signed power the reference writes."""
    params = default_parameters()
    params["Soil_Properties"]["n"] = 4.0
    cols = dg.ColumnTables(params, WELLS[200])
    soil, theta = cols.soil, cols.theta
    w = theta.res + (cols.por_raw - theta.res) / (1.0 + (soil.alpha * theta.wlt) ** soil.n) ** soil.m
    assert np.array_equal(cols.wlt_raw, np.minimum(w, cols.fc_raw))


def test_complex_profiles_are_an_error_not_a_cast():
    class Soil:
        n, alpha, m = 2.0, 0.009, 0.5
    theta = dg.WaterContent()
    theta.wlt = complex(0.0, 1.0)                    # whatever makes the curve complex must not be silently dropped
    z = np.arange(0.0, 100.0, 5.0)
    with pytest.raises((ValueError, TypeError)):
        dg.porosity_profiles(z, (0.0, 50.0, 80.0, 95.0), theta, Soil, "Stratified")


def test_a_sweep_point_may_not_change_what_every_point_shares():
    """The forcing digest (ET series, surface evaporation) and the PREDICT gate are taken once from the base parameters:
    a point that overrides them is refused instead of silently running with the base values."""
    from hydromodel_amd.ensemble import check_sweep_points
    params = default_parameters()
    ok = check_sweep_points(params, [{"Soil_Properties": {"a0": 0.012}},
                                     {"Hydraulic_Conductivity": {"Sigma_Noise": 1.0},
                                      "Environmental": {"Interception_pct": 0.1}}])      # interception is per point
    assert ok[0]["Soil_Properties"]["a0"] == 0.012 and ok[1]["Environmental"]["Interception_pct"] == 0.1
    assert params["Soil_Properties"]["a0"] != 0.012                                      # the base is not mutated
    for bad in ({"Environmental": {"Atmospheric_Demand": 2.0}}, {"Environmental": {"Evaporation_pct": 0.5}},
                {"Environmental": {"Wet_Season_pct": 0.9}}, {"Simulation_Flags": {"PREDICT": True}},
                {"Well_No": 3}):
        with pytest.raises(ValueError, match="Sweep"):
            check_sweep_points(params, [{"Soil_Properties": {"a0": 0.012}}, bad])
