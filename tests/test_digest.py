"""Host digest (hydromodel_amd/digest.py) against golden G1 captured from the reference."""
import numpy as np
import pytest

from helpers import WELLS, digest, golden
from hydromodel_amd import digest as dg
from hydromodel_amd.synthetic import default_parameters, synthetic_forcing, synthetic_well


@pytest.mark.parametrize("well", [1, 200, 300, 401, 581])
def test_static_tables_match_reference_bit_for_bit(well):
    _, cols, forcing = digest(well)
    g = golden(f"g1_tables_{well}.npz")
    assert np.array_equal(cols.z, g["z"]) and np.array_equal(cols.x_mid, g["x_mid"])
    for mine, ref in ((cols.por_raw, "por_node"), (cols.fc_raw, "fc_node"), (cols.wlt_raw, "wlt_node"),
                      (cols.por_mid, "por_mid"), (cols.fc_mid, "fc_mid"), (cols.wlt_mid, "wlt_mid"),
                      (cols.meank_node, "meank_node"), (cols.meank_mid, "meank_mid")):
        assert np.array_equal(mine, g[ref]), ref
    nr = cols.n_root_first + cols.n_root_int
    assert nr == g["root_mid"].size
    assert np.array_equal(cols.root_mid[:nr], g["root_mid"])
    assert np.all(cols.root_mid[nr:] == 0.0)
    assert cols.ipsi50 == float(g["iPsi_50"])
    assert cols.max_root_depth == float(g["max_root_depth"])
    assert cols.sat_cells == float(g["sat_cells"])


@pytest.mark.parametrize("well", [1, 200, 300, 401, 581])
def test_forcing_digest_matches_reference(well):
    _, cols, forcing = digest(well)
    g = golden(f"g1_tables_{well}.npz")
    assert forcing.dim_t == 17520
    assert np.array_equal(forcing.atm, g["atm"])
    assert np.array_equal(forcing.precip, g["precip"])
    assert np.array_equal(forcing.zwtd_cm, g["zWtd_cm"])
    assert np.array_equal(forcing.hour, g["hour"])
    assert forcing.surface_evap == float(g["surface_evap"])
    assert np.array_equal(forcing.daylight, (g["hour"] >= 6) & (g["hour"] <= 17))
    assert np.all(cols.z[forcing.wtd_obs] == forcing.zwtd_cm)
    i = np.arange(forcing.dim_t)
    expect = ((forcing.precip > 0.5) | (i % 48 == 0))
    expect[0] = False
    assert np.array_equal(forcing.refresh.astype(bool), expect)


@pytest.mark.parametrize("n", [4, 10, 101, 200, 300, 401, 581])
def test_column_groups_equal_scipy(n):
    from scipy import sparse
    from scipy.optimize._numdiff import group_columns
    A = sparse.diags((np.ones(n - 1), np.ones(n), np.ones(n - 1)), offsets=(-1, 0, 1))
    assert np.array_equal(dg.group_columns_tridiagonal(n), group_columns(sparse.csc_matrix(A)))


def test_interp_linear_equals_scipy_interp1d():
    from scipy.interpolate import interp1d
    rng = np.random.default_rng(0)
    x = np.sort(rng.random(40)) * 100
    y = rng.standard_normal((3, 40))
    xn = np.concatenate((x, x[0] + (x[-1] - x[0]) * rng.random(200)))
    assert np.array_equal(dg.interp_linear(x, y, xn), interp1d(x, y)(xn))
    with pytest.raises(ValueError):
        dg.interp_linear(x, y, [x[-1] + 1.0])


@pytest.mark.parametrize("kind", ["Uniform", "Negative_Exp", "Gamma_pdf", "Mixture"])
def test_root_profiles_integrate_to_one(kind):
    # code/tests/test_tree_roots.py:60-94: every pdf integrates to one (rel_tol 1e-5)
    knots, pdf = dg.root_profile(200, 5.0, kind)
    assert abs(np.sum(pdf) * 5.0 - 1.0) < 1e-5
    assert knots[0] == 0.0 and knots[-1] == 1000.0
    with pytest.raises(ValueError):
        dg.root_profile(200, 5.0, "no_such_profile")


def test_gamma_pdf_matches_scipy():
    from scipy.stats import gamma
    x = np.linspace(1, 100, 200)
    assert np.allclose(dg._gamma_pdf(x, 2.5, 5.0), gamma.pdf(x, a=2.5, scale=5.0), rtol=1e-13, atol=0)


@pytest.mark.parametrize("kind", ["Constant", "Linear", "Exponential", "Stratified"])
def test_porosity_profiles_bounded(kind):
    # code/tests/test_porosity.py:50-213: shapes and min <= profile <= max
    p = default_parameters()
    cols = dg.ColumnTables(p, WELLS[200])
    por, fc, wlt = dg.porosity_profiles(cols.z, cols.layers, cols.theta, cols.soil, kind)
    assert por.shape == fc.shape == wlt.shape == (200,)
    assert np.all(por >= cols.theta.min) and np.all(por <= cols.theta.max)
    assert np.all(wlt <= fc)


def test_porosity_profile_errors():
    p = default_parameters()
    cols = dg.ColumnTables(p, WELLS[200])
    with pytest.raises(ValueError):
        dg.porosity_profiles(cols.z, cols.layers, cols.theta, cols.soil, "bogus")
    with pytest.raises(ValueError):     # unseeded RNG in the reference: no oracle, refused loudly
        dg.porosity_profiles(cols.z, cols.layers, cols.theta, cols.soil, "Noisy")
    with pytest.raises(RuntimeError):
        dg.porosity_profiles(cols.z[::-1], cols.layers, cols.theta, cols.soil, "Constant")


def test_holder_validation_mirrors_reference():
    # code/tests/test_soil_properties.py, test_water_content.py, test_hydraulic_conductivity.py
    with pytest.raises(ValueError):
        dg.SoilProperties(n=1.0)
    with pytest.raises(ValueError):
        dg.SoilProperties(alpha=0.0)
    assert dg.SoilProperties(psi_sat=5.0).psi_sat == 0.0
    assert dg.SoilProperties(epsilon=0.0).epsilon == 1.0e-8
    with pytest.raises(ValueError):
        dg.WaterContent(minimum=0.4, maximum=0.3)
    with pytest.raises(ValueError):
        dg.WaterContent(residual=0.1, minimum=0.05)
    assert dg.WaterContent().mid == 0.5 * (0.30 + 0.08)
    for bad in ({"sat_soil": 0.0}, {"sat_saprolite": -1.0}, {"sat_fresh_bedrock": 0.0}):
        with pytest.raises(ValueError):
            dg.HydraulicConductivity(**bad)
    assert dg.HydraulicConductivity(sigma_noise=-3.0).sigma_noise == 0.0


def test_bad_holder_falls_back_to_defaults(capsys):
    # simulation.py:146-196
    p = default_parameters()
    p["Soil_Properties"]["n"] = 0.5
    cols = dg.ColumnTables(p, WELLS[200])
    assert cols.soil.n == 2.0 and cols.soil.psi_sat == -100.0
    assert "SoilProperties failed to initialize" in capsys.readouterr().out


def test_fully_saturated_well_is_refused():
    with pytest.raises(RuntimeError):
        dg.ColumnTables(default_parameters(), {"soil": 0.0, "saprolite": 50.0, "weathered": 200.0,
                                               "max_depth": 100.0, "sat_depth": 100.0})


def test_synthetic_forcing_shape():
    ids, datenum, precip, wtd = synthetic_forcing(2)
    assert ids.size == 2 * 17520 and datenum[0] == 733682.0
    daily = precip.reshape(-1, 48)
    assert np.all(daily == daily[:, :1])                 # uniform within a day
    assert 80.0 < precip.sum() / 2 < 170.0               # ~125 cm / yr
    assert np.all(wtd == -3.0)
    assert synthetic_well(300)["max_depth"] == 1495.0


def test_wet_season_bit_and_skipped_rows():
    """richards_pde.py:315: Oct-Mar of the rounded stamp is the wet season (PREDICT mode reads it); a row whose
    observation is off the grid is skipped before anything is drawn (simulation.py:582-602), so it never refreshes."""
    _, cols, forcing = digest(200)
    g = golden("g1_tables_200.npz")
    assert np.array_equal(forcing.wet_season.astype(bool), np.isin(g["month"], [10, 11, 12, 1, 2, 3]))
    assert forcing.wet_season[0] == 1 and forcing.wet_season[-1] == 0          # the record runs Oct 1 -> Sep 30
    assert set(np.unique(forcing.daylight | (forcing.wet_season << 1))) <= {0, 1, 2, 3}
    assert not np.any(forcing.refresh[forcing.wtd_obs < 0])
    from hydromodel_amd.stepper import column_params
    p = column_params(cols, forcing.surface_evap)
    assert p.flag_predict == 0 and p.sat_cells == int(cols.sat_cells) == 20
    assert column_params(cols, forcing.surface_evap, {"PREDICT": True}).flag_predict == 1


@pytest.mark.parametrize("por,root", [("Constant", "Uniform"), ("Linear", "Gamma_pdf"), ("Exponential", "Mixture")])
def test_other_profile_types_match_reference_bit_for_bit(por, root):
    """porosity.py:80-100 and tree_roots.py:57-123: every profile type the ensemble stepper accepts, against tables
    the reference built itself (g1q_*, `make_golden.py profiles`)."""
    from helpers import WELLS
    params = default_parameters()
    params["Hydrological_Model"]["Porosity_Profile"] = por
    params["Trees"]["Root_Pdf_Profile"] = root
    cols = dg.ColumnTables(params, WELLS[200])
    g = golden(f"g1q_tables_{por.lower()}_{root.lower()}.npz")
    for mine, ref in ((cols.por_raw, "por_node"), (cols.fc_raw, "fc_node"), (cols.wlt_raw, "wlt_node"),
                      (cols.por_mid, "por_mid"), (cols.fc_mid, "fc_mid"), (cols.wlt_mid, "wlt_mid"),
                      (cols.meank_node, "meank_node"), (cols.meank_mid, "meank_mid")):
        assert np.array_equal(mine, g[ref]), ref
    nr = cols.n_root_first + cols.n_root_int
    if root == "Uniform":
        assert np.array_equal(cols.root_mid[:nr], g["root_mid"])
    else:
        # gamma pdf: this repo evaluates exp((a-1) log x - x - lgamma(a)) where scipy.stats goes through xlogy / gammaln
        assert np.allclose(cols.root_mid[:nr], g["root_mid"], rtol=1e-13, atol=0)
    assert cols.max_root_depth == float(g["max_root_depth"]) and cols.ipsi50 == float(g["iPsi_50"])
