"""Host-side digest: JSON parameters + forcing table -> flat arrays for the HIP stepper.

Everything that is *static* during a run (depth profiles, layer tables, root pdf,
forcing series, FD-Jacobian column groups) is evaluated once here, exactly the way
the reference evaluates it inside its per-call Python objects, and handed to the
device as plain arrays.  Reference anchors:

* grid / wells / ET series / evaporation: ``code/src/simulation.py:79-352``
* porosity, field capacity, wilting point: ``code/src/porosity.py:22-208``
* root pdf: ``code/src/tree_roots.py:26-177``
* layer mean-K tables: ``code/src/models/vrettas_fung.py:22-49``
* linear ``interp1d`` semantics (scipy, ``_call_linear``): slope*(x-x_lo)+y_lo with
  ``searchsorted(side='left')`` clipped to [1, n-1]
* FD-Jacobian column groups: ``scipy.optimize._numdiff.group_columns`` (greedy
  Curtis-Powell-Reid on a ``RandomState(0)`` column permutation), called from
  ``scipy/integrate/_ivp/bdf.py`` (``_validate_jac``) for ``jac_sparsity``
  (``code/src/richards_pde.py:499-503``).
"""
import json
import math
from functools import lru_cache
from pathlib import Path

import numpy as np

DZ_CM = 5.0  # simulation.py:101 (hard-coded)

MODEL_VRETTAS_FUNG = 0
MODEL_VAN_GENUCHTEN = 1


# --------------------------------------------------------------------------- helpers
def interp_linear(xk, yk, x_new):
    """scipy.interpolate.interp1d(kind='linear') restated (bounds_error=True)."""
    xk = np.asarray(xk, dtype=float)
    yk = np.asarray(yk, dtype=float)
    x_new = np.atleast_1d(np.asarray(x_new, dtype=float))
    if x_new.size and (x_new.min() < xk[0] or x_new.max() > xk[-1]):
        raise ValueError("A value in x_new is outside the interpolation range.")
    idx = np.searchsorted(xk, x_new).clip(1, xk.size - 1).astype(int)
    lo, hi = idx - 1, idx
    # a zero-width knot interval (TreeRoots' linspace(0, ln dz, ln) has one when ln = 1: tree_roots.py:135) has no slope;
    # scipy divides 0 / 0 there and multiplies the NaN by a zero offset only where nobody looks.  The value at such a
    # knot is the knot's: slope 0.
    dx = xk[hi] - xk[lo]
    flat = dx == 0.0
    slope = (yk[..., hi] - yk[..., lo]) / np.where(flat, 1.0, dx)
    slope = np.where(flat, 0.0, slope)
    return slope * (x_new - xk[lo]) + yk[..., lo]


def group_columns_tridiagonal(n):
    """Column groups scipy uses for a tridiagonal ``jac_sparsity`` of size n (cached by n: a sweep builds one
    ``ColumnTables`` per parameter point on the same grid).

    Restates ``group_columns(A, order=0)``: columns are visited in the order
    ``RandomState(0).permutation(n)``; a column joins the current group when it shares
    no row with the group's union (column j of a tridiagonal pattern has rows j-1..j+1).
    """
    return _group_columns_tridiagonal(int(n)).copy()


@lru_cache(maxsize=16)
def _group_columns_tridiagonal(n):
    order = np.random.RandomState(0).permutation(n)
    groups_p = -np.ones(n, dtype=np.int64)
    rows = [np.arange(max(c - 1, 0), min(c + 2, n)) for c in order]
    current = 0
    for i in range(n):
        if groups_p[i] >= 0:
            continue
        groups_p[i] = current
        union = np.zeros(n, dtype=bool)
        union[rows[i]] = True
        all_grouped = True
        for j in range(n):
            if groups_p[j] >= 0:
                continue
            all_grouped = False
            if not union[rows[j]].any():
                union[rows[j]] = True
                groups_p[j] = current
        if all_grouped:
            break
        current += 1
    groups = np.empty(n, dtype=np.int32)
    groups[order] = groups_p
    return groups


def _gamma_pdf(x, shape, scale):
    """scipy.stats.gamma.pdf(x, a=shape, scale=scale) for x > 0."""
    xs = np.asarray(x, dtype=float) / scale
    return np.exp((shape - 1.0) * np.log(xs) - xs - math.lgamma(shape)) / scale


# --------------------------------------------------------------------------- parameter holders
class _Checked:
    """Data descriptor: an attribute whose every assignment goes through ``rule(owner, value)``, which returns the
    value to store or raises ValueError -- the reference's holders do the same with one property setter per field."""

    def __init__(self, rule):
        self.rule = rule

    def __set_name__(self, owner, name):
        self.slot = "_" + name
        self.name = name

    def __get__(self, obj, owner=None):
        return self if obj is None else getattr(obj, self.slot)

    def __set__(self, obj, value):
        setattr(obj, self.slot, self.rule(obj, self.name, value))


def _require(cond_text, cond):
    def rule(obj, name, value):
        if not cond(value):
            raise ValueError(f" {type(obj).__name__}: '{name}' = {value} {cond_text}.")
        return value
    return rule


class SoilProperties:
    """code/src/soil_properties.py:13-160: n > 1, alpha > 0 (ValueError otherwise, at construction and on assignment);
    psi_sat is clamped to <= 0, epsilon to >= 1e-8."""

    n = _Checked(_require("should be > 1", lambda v: v > 1.0))
    alpha = _Checked(_require("should be strictly positive", lambda v: v > 0.0))
    psi_sat = _Checked(lambda obj, name, v: float(np.minimum(v, 0.0)))
    epsilon = _Checked(lambda obj, name, v: float(np.maximum(v, 1.0e-8)))

    def __init__(self, n=2.0, alpha=0.009, psi_sat=-100.0, epsilon=1.0e-7):
        self.n, self.alpha, self.psi_sat, self.epsilon = n, alpha, psi_sat, epsilon

    @property
    def m(self):
        return 1.0 - (1.0 / self.n)


def _water_content_rule(obj, name, value):
    """Property setters of water_content.py:88-190: the new value is taken AS GIVEN (only the constructor clips to
    [0, 1], :40-42); if 0 <= res < min < max <= 1 no longer holds the assignment is refused -- ValueError, old value
    kept (the reference's own test: ``max += 1.0`` raises, code/tests/test_water_content.py:39-58)."""
    trial = {k: getattr(obj, "_" + k) for k in ("res", "min", "max")}
    trial[name] = value
    if not (0.0 <= trial["res"] < trial["min"] < trial["max"] <= 1.0):
        label = {"res": "residual", "min": "minimum", "max": "maximum"}[name]
        raise ValueError(f" {type(obj).__name__}: The new {label}: {value} is not consistent with the rest of the values.")
    return value


class WaterContent:
    """code/src/water_content.py:13-240.  wlt / flc are pressure heads in cm and may be negative."""

    min = _Checked(_water_content_rule)
    max = _Checked(_water_content_rule)
    res = _Checked(_water_content_rule)

    def __init__(self, minimum=0.08, maximum=0.30, residual=0.05, wilting=-1500.0, field_cap=340.0):
        clip = lambda v: float(np.maximum(np.minimum(1.0, v), 0.0))  # noqa: E731
        lo, hi, res = clip(minimum), clip(maximum), clip(residual)
        if not (0.0 <= res < lo < hi <= 1.0):
            raise ValueError(" WaterContent: The volumetric water content input values are incorrect.")
        self._res, self._min, self._max = res, lo, hi
        self.wlt = wilting
        self.flc = field_cap

    @property
    def mid(self):
        return 0.5 * (self.max + self.min)


class HydraulicConductivity:
    """code/src/hydraulic_conductivity.py:12-240: the three saturated conductivities must be > 0, sigma_noise and
    lambda_exponent >= 0 (ValueError otherwise, at construction and on assignment)."""

    sat_soil = _Checked(_require("should be strictly positive", lambda v: v > 0.0))
    sat_saprolite = _Checked(_require("should be strictly positive", lambda v: v > 0.0))
    sat_fresh_bedrock = _Checked(_require("should be strictly positive", lambda v: v > 0.0))
    sigma_noise = _Checked(_require("should be non-negative", lambda v: v >= 0.0))
    lambda_exponent = _Checked(_require("should be non-negative", lambda v: v >= 0.0))

    def __init__(self, sat_soil=8.5, sat_saprolite=3.2, sat_fresh_bedrock=0.1,
                 sigma_noise=2.0, lambda_exponent=1.0):
        self.sat_soil, self.sat_saprolite, self.sat_fresh_bedrock = sat_soil, sat_saprolite, sat_fresh_bedrock
        # the constructor clamps these two at zero (hydraulic_conductivity.py:56-57); only assignment raises
        self.sigma_noise = float(np.maximum(sigma_noise, 0.0))
        self.lambda_exponent = float(np.maximum(lambda_exponent, 0.0))


# --------------------------------------------------------------------------- inverse retention curve
SATURATION_CUT = 0.99998   # hydrological_model.py:103


def inverse_retention(theta, porosity, theta_res, alpha, n, m, epsilon, dz):
    """Water content -> pressure head, the inverse van Genuchten curve of ``HydrologicalModel.pressure_head``
    (hydrological_model.py:43-119); the ONE implementation here, shared by the plugin objects (``models.py``) and the
    spin-up start profile (``ensemble.pressure_head``).

    ``theta`` and ``porosity`` broadcast against each other ([D] or [D, M]).  Effective saturation is clipped to
    [epsilon, 1]; cells at or above ``SATURATION_CUT`` count as saturated and receive a hydrostatic ramp 0, dz, 2 dz, ...
    in flattened (row-major) order of the saturated cells (:103,112); anything non-finite becomes -1e5 cm (:115).
    Returns (psi, s_eff) with the shape of ``theta``."""
    theta = np.asarray(theta, dtype=float)
    span = porosity - theta_res
    s_eff = np.clip((np.clip(theta, theta_res, porosity) - theta_res) / span, epsilon, 1.0)
    wet = s_eff >= SATURATION_CUT
    psi = np.zeros(theta.shape)
    with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
        psi[~wet] = -((s_eff[~wet] ** (-1.0 / m) - 1.0) ** (1.0 / n)) / alpha
    psi[wet] = np.arange(0, np.count_nonzero(wet)) * dz
    psi[~np.isfinite(psi)] = -1.0e+5
    return psi, s_eff


# --------------------------------------------------------------------------- static profiles
def porosity_profiles(z_grid, layers, theta, soil, p_model):
    """(porosity, field_cap, wilting_point) at the nodes -- porosity.py:66-181."""
    z_grid = np.atleast_1d(np.asarray(z_grid, dtype=float))
    if not z_grid.size:
        raise ValueError(" Porosity: Input array z_grid is empty.")
    if np.any(np.diff(z_grid) <= 0.0):
        raise RuntimeError(" Porosity: Space domain z_grid is not increasing.")
    len_z = z_grid.size
    (_, l_sapr, l_wbed, l_fbed) = layers
    sap = (z_grid >= l_sapr) & (z_grid <= l_wbed)
    web = (z_grid >= l_wbed) & (z_grid <= l_fbed)
    kind = str.upper(p_model)
    if kind == "CONSTANT":
        q_sat = theta.max * np.ones(len_z)
    elif kind == "LINEAR":
        q_sat = np.linspace(theta.max, theta.min, len_z)
    elif kind == "EXPONENTIAL":
        p0 = theta.max
        p1 = np.log(p0 / theta.min) / z_grid[-1]
        q_sat = p0 * np.exp(-z_grid * p1)
    elif kind == "STRATIFIED":
        q_sat = theta.max * np.ones(len_z)
        if np.any(sap):
            q_sat[sap] = np.linspace(theta.max, theta.mid, sap.sum())
        if np.any(web):
            p0 = theta.mid
            p1 = np.log(p0 / theta.min) / z_grid[-1]
            q_sat[web] = p0 * np.exp(-np.linspace(0, l_fbed, web.sum()) * p1)
    elif kind == "NOISY":
        # porosity.py:122-158 draws from an UNSEEDED generator: not reproducible, nothing to check against.
        raise ValueError(" Porosity: profile type 'Noisy' is not supported by the ensemble stepper "
                         "(unseeded RNG in the reference, porosity.py:124).")
    else:
        raise ValueError(f" Porosity: Wrong porosity profile type: {p_model}")
    q_sat = np.minimum(np.maximum(q_sat, theta.min), theta.max)

    def fun_wrc(psi):
        # porosity.py:172-173 writes (alpha * psi) ** n with the signed psi.  For the even integer n the reference
        # ships (n = 2) that equals (alpha * |psi|) ** n bit for bit.  For any other n and psi < 0 (the wilting point,
        # -1500 cm by default) the reference's expression is complex, Porosity returns complex128 profiles and the
        # first RHS evaluation raises "setting an array element with a sequence" (richards_pde.py:119): the reference
        # cannot run there.  EXTENSION, no reference oracle: the retention curve is evaluated on |psi|, the form
        # vrettas_fung.py:115 itself uses, so every n > 1 has a real field capacity / wilting point.
        base = soil.alpha * psi
        if not (float(soil.n) == 2.0 * round(float(soil.n) / 2.0)):
            base = np.abs(base)
        return theta.res + (q_sat - theta.res) / (1.0 + base ** soil.n) ** soil.m

    field_cap = np.maximum(fun_wrc(theta.flc), theta.res)
    wilting = np.minimum(fun_wrc(theta.wlt), field_cap)
    if np.iscomplexobj(field_cap) or np.iscomplexobj(wilting) or not (
            np.all(np.isfinite(field_cap)) and np.all(np.isfinite(wilting))):
        raise ValueError(" Porosity: field capacity / wilting point profiles are not real and finite.")
    if np.any(wilting < theta.res):
        raise ValueError(" Porosity: wilting point profile falls below the residual water content.")
    return q_sat.flatten(), field_cap, wilting


def root_profile(ln, dz, r_model):
    """Normalised root pdf on ``ln`` cells and its interpolation knots -- tree_roots.py:37-135."""
    ln = int(ln)
    n_cells = np.linspace(1, 100, ln)
    kind = str.upper(r_model)
    if kind == "UNIFORM":
        root_pdf = np.ones(ln) / ln
    elif kind == "NEGATIVE_EXP":
        root_pdf = np.exp(-(n_cells / 15)) / 15
    elif kind == "GAMMA_PDF":
        root_pdf = _gamma_pdf(n_cells, 2.5, 5.0)
    elif kind == "MIXTURE":
        root_pdf = 0.15 * (np.exp(-(n_cells / 15)) / 15) + 0.85 * _gamma_pdf(n_cells, 2.5, 5.0)
    else:
        raise ValueError(f" TreeRoots: Wrong root density profile type: {r_model}")
    root_pdf = np.maximum(root_pdf, 1.0e-8)
    total_area = np.sum(root_pdf) * dz
    return np.linspace(0.0, ln * dz, ln), np.atleast_1d(root_pdf / total_area)


def mean_k_tables(layers, k_hc):
    """1-cm knot tables of the layer-mean conductivity -- vrettas_fung.py:26-48."""
    (l0, l1, l2, l3) = layers
    z_sapr = np.arange(l1, l2 + 1)
    k_sapr = np.linspace(k_hc.sat_soil, k_hc.sat_saprolite, z_sapr.size)
    z_wbed = np.arange(l2, l3 + 1)
    p0 = k_hc.sat_saprolite
    p1 = np.log(p0 / k_hc.sat_fresh_bedrock) / l3
    k_wbed = p0 * np.exp(-np.linspace(0, l3, z_wbed.size) * p1)
    return (z_sapr, k_sapr), (z_wbed, k_wbed)


def layer_tables(z, layers, k_hc):
    """Per-cell (mean K, noise coefficient) for depths ``z`` -- vrettas_fung.py:129-200.

    Noise coefficient: 0.05 (soil) / 0.10 (saprolite) / 1.0 (weathered bedrock);
    -1 marks a cell that belongs to no layer (K_bkg stays ``sat_soil``, vrettas_fung.py:143).
    """
    z = np.atleast_1d(np.asarray(z, dtype=float))
    (l0, l1, l2, l3) = layers
    (zs, ks), (zw, kw) = mean_k_tables(layers, k_hc)
    mean_k = np.full(z.shape, float(k_hc.sat_soil))
    coef = -np.ones(z.shape)
    soil = (z >= l0) & (z < l1)
    sapr = (z >= l1) & (z < l2)
    wbed = (z >= l2) & (z <= l3)
    coef[soil] = 0.05
    if sapr.any():
        mean_k[sapr] = interp_linear(zs, ks, z[sapr])
        coef[sapr] = 0.10
    if wbed.any():
        mean_k[wbed] = interp_linear(zw, kw, z[wbed])
        coef[wbed] = 1.0
    return mean_k, coef


# --------------------------------------------------------------------------- the digest
class ColumnTables:
    """Static per-parameter-point tables and scalars (one soil column geometry)."""

    NODE_ROWS = ("por", "meank", "noisec")
    MID_ROWS = ("por", "fc", "wlt", "root", "meank", "noisec")

    def __init__(self, params, well, model_name=None):
        dz = DZ_CM
        self.dz = dz
        self.layers = (well["soil"], well["saprolite"], well["weathered"], well["max_depth"])
        if well["sat_depth"] >= well["max_depth"]:
            raise RuntimeError(" Simulation: The well seems fully saturated.")
        self.sat_cells = float(np.ceil(well["sat_depth"] / dz))
        self.z = np.arange(well["soil"], well["max_depth"] + dz, dz)
        self.dim_d = int(self.z.size)
        if self.dim_d < 4:
            raise RuntimeError(" ColumnTables: the spatial grid needs at least 4 nodes.")
        dx = np.diff(self.z)
        self.x_mid = self.z[:-1] + 0.5 * dx
        self.soil = _try_ctor(SoilProperties, "SoilProperties", lambda: (
            params["Soil_Properties"]["n"], params["Soil_Properties"]["a0"],
            params["Soil_Properties"]["psi_sat"], params["Soil_Properties"]["epsilon"]))
        self.theta = _try_ctor(WaterContent, "WaterContent", lambda: (
            params["Water_Content"]["Theta_Min"], params["Water_Content"]["Theta_Max"],
            params["Water_Content"]["Theta_Residual"], params["Water_Content"]["Wilting_Point_cm"],
            params["Water_Content"]["Field_Capacity_cm"]))
        self.k_hc = _try_ctor(HydraulicConductivity, "HydraulicConductivity", lambda: (
            params["Hydraulic_Conductivity"]["Sat_Soil"], params["Hydraulic_Conductivity"]["Sat_Saprolite"],
            params["Hydraulic_Conductivity"]["Sat_Fresh_Bedrock"],
            params["Hydraulic_Conductivity"]["Sigma_Noise"],
            params["Hydraulic_Conductivity"]["Lambda_Exponent"]))
        name = model_name if model_name is not None else params["Hydrological_Model"]["Name"]
        # simulation.py:219-231: anything that is not VRETTAS_FUNG silently becomes vanGenuchten.
        self.model = MODEL_VRETTAS_FUNG if str.upper(name) == "VRETTAS_FUNG" else MODEL_VAN_GENUCHTEN

        # Node profiles (raw) and what the plugin sees through interp1d at nodes / midpoints.
        self.por_raw, self.fc_raw, self.wlt_raw = porosity_profiles(
            self.z, self.layers, self.theta, self.soil, params["Hydrological_Model"]["Porosity_Profile"])
        prof = np.array([self.por_raw, self.fc_raw, self.wlt_raw])
        node_i = interp_linear(self.z, prof, self.z)
        if not np.any(self.z[:1]):
            # Porosity.__call__ (porosity.py:200-205): a depth array whose entries are all zero
            # returns the raw arrays; that is how the top-node BC call sees z=[0.0].
            node_i[:, 0] = prof[:, 0]
        mid_i = interp_linear(self.z, prof, self.x_mid)
        self.por_node, self.fc_node, self.wlt_node = node_i
        self.por_mid, self.fc_mid, self.wlt_mid = mid_i

        self.meank_node, self.noisec_node = layer_tables(self.z, self.layers, self.k_hc)
        self.meank_mid, self.noisec_mid = layer_tables(self.x_mid, self.layers, self.k_hc)

        # Roots (simulation.py:215; tree_roots.py:50,135; richards_pde.py:220-223).
        ln = int(np.ceil(params["Trees"]["Max_Root_Depth_cm"] / dz))
        self.max_root_depth = ln * dz
        knots, pdf = root_profile(ln, dz, params["Trees"]["Root_Pdf_Profile"])
        in_root = self.x_mid <= self.max_root_depth
        self.root_mid = np.zeros(self.x_mid.size)
        if in_root.any():
            self.root_mid[in_root] = interp_linear(knots, pdf, self.x_mid[in_root])
        self.n_root_first = int(in_root[0])
        self.n_root_int = int(in_root[1:].sum())
        if not np.all(in_root[:in_root.sum()]):
            raise RuntimeError(" ColumnTables: the root zone is not a prefix of the grid.")

        env = params["Environmental"]
        self.interception = env["Interception_pct"]
        self.lai = params["Trees"]["Leaf_Area_Index"]
        theta_50 = np.maximum(0.5 ** (1.0 / self.k_hc.lambda_exponent), 0.05) \
            if self.k_hc.lambda_exponent > 0.0 else 0.05
        self.ipsi50 = float(self.soil.alpha / np.sqrt(theta_50 ** (-2.0) - 1.0))
        self.flags = {k: bool(params["Simulation_Flags"].get(k, False))
                      for k in ("SPINUP", "ET", "LF", "HLIFT", "PREDICT")}
        # bc_fun's np.all(theta_left > theta_res) ranges over the FULL profile when z[0]==0
        # (SURVEY.md §3.6); the smallest delta_s decides it.
        if not np.any(self.z[:1]):
            self.evap_delta_min = float(np.min(self.por_raw - self.theta.res))
        else:
            self.evap_delta_min = float(self.por_node[0] - self.theta.res)
        self.groups = group_columns_tridiagonal(self.dim_d)
        self.n_groups = int(self.groups.max()) + 1

    # flat views for the C-ABI -------------------------------------------------
    def node_table(self):
        return np.ascontiguousarray(np.stack([self.por_node, self.meank_node, self.noisec_node]))

    def mid_table(self):
        return np.ascontiguousarray(np.stack([self.por_mid, self.fc_mid, self.wlt_mid, self.root_mid,
                                              self.meank_mid, self.noisec_mid]))


def _try_ctor(cls, label, get_args):
    """simulation.py:146-196: a failing holder falls back to its defaults with a message."""
    try:
        return cls(*get_args())
    except Exception as e0:  # noqa: BLE001 - mirrors the reference's blanket handler
        print(f" {label} failed to initialize: {e0}. It will use default initialization parameters.")
        return cls()


class ForcingDigest:
    """Per-row forcing series as struct-of-arrays -- simulation.py:233-352, 576-602."""

    def __init__(self, params, data, cols):
        import pandas as pd
        r_datenum = data.loc[:, "Datenum"]
        timestamps = pd.to_datetime(r_datenum - 719529, unit="D")
        rounded = timestamps.dt.round(freq="s")
        self.hour = rounded.dt.hour.to_numpy().astype(np.int32)
        self.month_rounded = rounded.dt.month.to_numpy().astype(np.int32)
        month = timestamps.dt.month.to_numpy()          # ET split uses the UNROUNDED stamps (:295-326)
        z_wtd_cm = np.array(np.abs(np.round(100.0 * data.loc[:, "WTD_m"])))
        if np.any(np.isnan(z_wtd_cm)):
            raise RuntimeError(" Simulation: Water table depth observations contain NaN values.")
        idx = np.searchsorted(cols.z, z_wtd_cm, side="left")
        if np.any(idx >= cols.dim_d):
            raise IndexError("index 0 is out of bounds for axis 0 with size 0")
        self.zwtd_cm = cols.z[idx]
        self.wtd_obs = idx.astype(np.int32)
        precip = np.array(data.loc[:, "Precipitation_cm"], dtype=float)
        if np.any(np.isnan(precip)):
            raise ValueError(" Simulation: Precipitation observations contain NaN values.")
        self.precip = precip
        env = params["Environmental"]
        total_atm = 13.4253 * 10 * env["Atmospheric_Demand"]
        et_pct = np.minimum(np.maximum(env["Wet_Season_pct"], 0.0), 1.0)
        wet_et, dry_et = et_pct * total_atm, (1.0 - et_pct) * total_atm
        dry = np.isin(month, [4, 5, 6, 7, 8, 9])
        dim_t = int(month.size)
        n_dry = int(dry.sum())
        n_wet = dim_t - n_dry
        atm = np.zeros(dim_t)
        with np.errstate(divide="ignore", invalid="ignore"):
            if n_dry:
                atm[dry] = 2.0 * dry_et / n_dry
            if n_wet:
                atm[~dry] = 2.0 * wet_et / n_wet
        atm[np.isnan(atm)] = 0.0
        self.atm = atm
        self.dim_t = dim_t
        self.surface_evap = float(2.0 * np.sum(env["Evaporation_pct"] * precip) / dim_t)
        self.daylight = ((self.hour >= 6) & (self.hour <= 17)).astype(np.uint8)
        # richards_pde.py:315 (PREDICT mode): the wet-season sink coefficient applies in Oct-Mar of the ROUNDED stamp
        self.wet_season = np.isin(self.month_rounded, [10, 11, 12, 1, 2, 3]).astype(np.uint8)
        i = np.arange(dim_t)
        self.refresh = ((precip > 0.5) | (np.mod(i, 48) == 0)).astype(np.uint8)
        self.refresh[0] = 0                                   # row 0 is the initial state, never solved
        self.refresh[self.wtd_obs < 0] = 0                    # a skipped row draws nothing (simulation.py:582-602)
        self.datenum = np.asarray(r_datenum, dtype=float)


def load_site_well(params):
    """simulation.py:106-119."""
    with open(Path(params["Site_Information"]), "r") as site_file:
        site_info = json.load(site_file)
    if not str(params["Well_No"]) in site_info["Well"]:
        raise ValueError(" Simulation: The selected well does not exist in the site information file.")
    return site_info["Well"][str(params["Well_No"])]
