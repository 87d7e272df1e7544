"""Parameter holders and static-profile objects: the acceptance checks of the reference's own unit tests, restated.

What the reference's suite asks (code/tests/test_soil_properties.py, test_hydraulic_conductivity.py,
test_water_content.py, test_porosity.py, test_tree_roots.py): out-of-range values raise ValueError at construction and
on assignment and leave the object valid; profile lookups keep their shapes and stay inside [theta_min, theta_max];
root densities integrate to one; unknown profile names raise.  Written against this repo's classes, not copied.
"""
import numpy as np
import pytest

from hydromodel_amd import digest as dg
from hydromodel_amd import models


def test_soil_properties_ranges():
    sp = dg.SoilProperties()
    assert (sp.n, sp.alpha, sp.psi_sat, sp.epsilon) == (2.0, 0.009, -100.0, 1.0e-7) and sp.m == 0.5
    for bad in (1.0, 0.3, -2.0):
        with pytest.raises(ValueError):
            sp.n = bad
        with pytest.raises(ValueError):
            dg.SoilProperties(n=bad)
    for bad in (0.0, -0.009):
        with pytest.raises(ValueError):
            sp.alpha = bad
        with pytest.raises(ValueError):
            dg.SoilProperties(alpha=bad)
    assert (sp.n, sp.alpha) == (2.0, 0.009)              # refused values leave the object as it was
    sp.psi_sat = 25.0                                    # suction at saturation cannot be positive: clamped, no error
    assert sp.psi_sat == 0.0
    sp.epsilon = 0.0
    assert sp.epsilon == 1.0e-8
    sp.n = 3.0
    assert abs(sp.m - (1.0 - 1.0 / 3.0)) < 1e-16


def test_hydraulic_conductivity_ranges():
    k = dg.HydraulicConductivity()
    for name in ("sat_soil", "sat_saprolite", "sat_fresh_bedrock"):
        for bad in (0.0, -1.0):
            with pytest.raises(ValueError):
                setattr(k, name, bad)
            with pytest.raises(ValueError):
                dg.HydraulicConductivity(**{name: bad})
        assert getattr(k, name) > 0.0
    for name in ("sigma_noise", "lambda_exponent"):
        with pytest.raises(ValueError):
            setattr(k, name, -0.5)
        setattr(k, name, 0.0)                            # zero is allowed
        assert getattr(k, name) == 0.0
    # the constructor clamps the last two instead of raising (hydraulic_conductivity.py:56-57)
    k2 = dg.HydraulicConductivity(sigma_noise=-3.0, lambda_exponent=-1.0)
    assert k2.sigma_noise == 0.0 and k2.lambda_exponent == 0.0


def test_water_content_ordering():
    w = dg.WaterContent()
    assert (w.min, w.max, w.res) == (0.08, 0.30, 0.05) and abs(w.mid - 0.19) < 1e-16
    with pytest.raises(ValueError):
        w.max = 0.01                                     # below min: refused
    with pytest.raises(ValueError):
        w.min -= 1.0                                     # the reference's own test_min (tests/test_water_content.py:50-58)
    with pytest.raises(ValueError):
        w.res = 0.5                                      # above min
    # setters do NOT clip (water_content.py:88-190; only the constructor does, :40-42): the reference's own test_max
    # (tests/test_water_content.py:39-48) -- `max += 1.0` raises and the old value stays
    with pytest.raises(ValueError):
        w.max += 1.0
    with pytest.raises(ValueError):
        w.max = 5.0
    with pytest.raises(ValueError):
        w.res = -0.01
    assert (w.min, w.max, w.res) == (0.08, 0.30, 0.05)   # every refused assignment left the old values behind
    w.max = 1.0                                          # the closed upper end is allowed
    assert w.max == 1.0
    assert dg.WaterContent(maximum=5.0).max == 1.0       # the constructor clips (inRange)
    for bad in (dict(minimum=0.4, maximum=0.3), dict(residual=0.2), dict(minimum=0.0, residual=0.0)):
        with pytest.raises(ValueError):
            dg.WaterContent(**bad)
    w.wlt, w.flc = -2000.0, 100.0                        # pressure heads: any sign
    assert (w.wlt, w.flc) == (-2000.0, 100.0)


@pytest.mark.parametrize("kind", ["Constant", "Linear", "Exponential", "Stratified"])
def test_porosity_profiles_shapes_and_bounds(kind):
    z = np.arange(0.0, 1505.0, 5.0)
    layers = (0.0, 50.0, 200.0, 1500.0)
    theta, soil = dg.WaterContent(), dg.SoilProperties()
    por = models.Porosity(z, layers, theta, soil, kind)
    full = por()
    assert all(a.shape == z.shape for a in full)
    assert np.all(full[0] >= theta.min) and np.all(full[0] <= theta.max)
    assert np.all(full[2] <= full[1]) and np.all(full[1] <= full[0] + 1e-15) and np.all(full[2] >= theta.res)
    one = por(np.array([137.5]))
    many = por(np.array([2.5, 7.5, 12.5, 747.5]))
    assert all(a.shape == (1,) for a in one) and all(a.shape == (4,) for a in many)
    assert np.all(many[0] >= theta.min) and np.all(many[0] <= theta.max)
    assert por.layers == layers and kind in str(por)


def test_porosity_rejects_what_it_cannot_reproduce():
    z = np.arange(0.0, 105.0, 5.0)
    args = ((0.0, 20.0, 50.0, 100.0), dg.WaterContent(), dg.SoilProperties())
    with pytest.raises(ValueError):
        models.Porosity(z, *args, "no_such_profile")
    with pytest.raises(ValueError, match="Noisy"):
        models.Porosity(z, *args, "Noisy")               # unseeded RNG in the reference (porosity.py:124): out of scope
    with pytest.raises(RuntimeError):
        models.Porosity(z[::-1], *args, "Constant")      # the grid must increase
    with pytest.raises(ValueError):
        models.Porosity(np.array([]), *args, "Constant")


@pytest.mark.parametrize("kind", ["Uniform", "Negative_Exp", "Gamma_pdf", "Mixture"])
def test_root_density_lookup_integrates_to_one(kind):
    ln, dz = 200, 5.0
    knots, pdf = dg.root_profile(ln, dz, kind)
    assert knots.shape == pdf.shape == (ln,) and np.all(pdf > 0.0)
    assert abs(np.sum(pdf) * dz - 1.0) < 1e-5
    z_mid = knots[:-1] + 0.5 * np.diff(knots)
    inter = dg.interp_linear(knots, pdf, z_mid)
    assert abs(np.sum(inter) * (knots[1] - knots[0]) - 1.0) < 2e-2      # the lookup between knots keeps the mass


def test_root_water_uptake_efficiency_properties():
    """tree_roots.py:179-292 through the oracle's restatement: no water above the wilting point -> no uptake at all;
    any wetter profile -> a density that integrates to one over the root zone."""
    from helpers import digest
    from oracle.oracle import Oracle
    _, cols, forcing = digest(200)
    o = Oracle(cols, forcing.surface_evap, flags={"LF": False})
    day = int(np.argmax((forcing.daylight == 1) & (forcing.precip == 0.0)))
    row = Oracle.row(0.0, forcing.atm[day], 1, forcing.wtd_obs[day])
    dry = np.full(cols.dim_d, -1.0e6)                    # theta -> theta_res < wilting point everywhere
    _, aux = o.rhs(row, dry, np.zeros(cols.dim_d), want_aux=True)
    assert np.all(aux["s"] == 0.0) and aux["tr_lf_int"][0] == 0.0
    rng = np.random.default_rng(2)
    wet = -np.abs(rng.standard_normal(cols.dim_d)) * 40.0
    _, aux = o.rhs(row, wet, np.zeros(cols.dim_d), want_aux=True)
    uptake = -aux["s"][1:1 + cols.n_root_int]
    assert np.all(uptake >= 0.0) and uptake.sum() > 0.0
    # sum(uptake) dz = min(atm, water_k): the normalised density carried the whole demand
    assert abs(uptake.sum() * cols.dz - aux["tr_lf_int"][0]) < 1e-15
    assert aux["tr_lf_int"][0] <= forcing.atm[day] * (1 + 1e-12)
