"""The ``Hydrological_Model`` plugin surface, with the plugin call on the GPU.

Mirrors ``/root/reference/code/src/models/``: ``HydrologicalModel(soil, porous, k_hc, theta_res, dz)``
(``hydrological_model.py:12``) with ``pressure_head(theta, z) -> (psi, s_eff)`` (``:43-119``), and the two
subclasses ``VrettasFung`` (``vrettas_fung.py:51-257``) and ``vanGenuchten`` (``vanGenuchten.py:23-126``) whose
``__call__(psi, z, *args) -> (q, K, C, k_bkg, q_inf_max)`` is what ``RichardsPDE`` and ``Simulation.run`` call as
``h_model`` (``richards_pde.py:198``, ``simulation.py:564,623``).  ``Porosity`` is the callable the reference hands
to them (``porosity.py:22-208``).

Where the arithmetic runs: the per-cell work of ``__call__`` (theta, K, C, K_bkg) executes in libhydrocol
(``hc_plugin_eval``, the same device function the diagnostics kernel uses); the static lookups -- porosity
interpolation, layer membership, layer-mean K tables -- are host NumPy, as in ``digest.py``.  There is no CPU path for
the call: without the library or a device it raises.  ``pressure_head`` is the reference's one-off host computation
 and stays on the host.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .digest import (MODEL_VAN_GENUCHTEN, MODEL_VRETTAS_FUNG, interp_linear, inverse_retention, layer_tables,
                     porosity_profiles)


class Porosity(object):
    """Depth profiles of porosity / field capacity / wilting point -- ``porosity.py:22-208``."""

    def __init__(self, z_grid, layers, theta, soil, p_model):
        self.p_layers = tuple(layers)
        self.p_model = p_model
        self.z_grid = np.atleast_1d(np.asarray(z_grid, dtype=float))
        self.profile, self.field_cap, self.wilting_point = porosity_profiles(self.z_grid, self.p_layers, theta, soil,
                                                                             p_model)
        self._stack = np.array([self.profile, self.field_cap, self.wilting_point])

    def __call__(self, z_new=None):
        # porosity.py:200-205: a falsy depth argument (None, 0.0, an all-zero array) returns the FULL profiles
        if np.any(z_new):
            out = interp_linear(self.z_grid, self._stack, z_new)
            return out[0], out[1], out[2]
        return self.profile, self.field_cap, self.wilting_point

    @property
    def layers(self):
        return self.p_layers

    def __str__(self):
        return f" Porosity Id({id(self)}): Type={self.p_model}"


class HydrologicalModel(object):
    """Base class -- ``hydrological_model.py:4-119``."""

    MODEL_ID = None

    def __init__(self, soil, porous, k_hc, theta_res, dz, device=0):
        self.n = soil.n
        self.m = soil.m
        self.alpha = soil.alpha
        self.psi_sat = soil.psi_sat
        self.epsilon = np.maximum(soil.epsilon, 1.0e-8)
        self.k_hc = k_hc
        self.porous = porous
        self.dz = dz
        self.theta_res = theta_res
        self.device = device

    def _check(self, a, z):
        z, a = np.atleast_1d(z, a)
        dim_d, dim_m = a.shape[0], None
        if a.ndim == 2:
            dim_m = a.shape[1]
        if dim_d != z.shape[0]:
            raise ValueError(f" {self.__class__.__name__}:"
                             f" Input size dimensions don't match:"
                             f" {dim_d} not equal to {z.shape[0]}.")
        return z, a, dim_d, dim_m

    def pressure_head(self, theta, z):
        """Inverse van Genuchten, ``hydrological_model.py:43-119``: returns (psi, s_eff)."""
        z, theta, dim_d, dim_m = self._check(theta, z)
        porous_z = np.atleast_1d(self.porous(z)[0])
        if dim_m is not None:
            porous_z = porous_z[:, None]               # one porosity per depth, shared by the M columns
        return inverse_retention(theta, porous_z, self.theta_res, self.alpha, self.n, self.m, self.epsilon, self.dz)

    # -- the plugin call on the device ---------------------------------------------------
    def _params(self):
        p = L.ColumnParams()
        p.model = self.MODEL_ID
        p.theta_res, p.alpha, p.n, p.m = self.theta_res, self.alpha, self.n, self.m
        p.psi_sat, p.epsilon = self.psi_sat, float(self.epsilon)
        p.lambda_exp, p.sigma_noise = self.k_hc.lambda_exponent, self.k_hc.sigma_noise
        p.sat_soil, p.dz = self.k_hc.sat_soil, self.dz
        return p

    def _evaluate(self, psi, z, n_rnd):
        z, psi, dim_d, dim_m = self._check(psi, z)
        porous_z, *_ = self.porous(z)
        porous_z = np.atleast_1d(np.asarray(porous_z, dtype=float))
        if porous_z.shape[0] != dim_d:
            # Porosity.__call__ returned the full profile for a falsy z (porosity.py:200): the reference then
            # broadcasts / fails the same way; only the single-cell call z = [0.0] is meaningful
            if dim_d != 1:
                raise ValueError(f" {self.__class__.__name__}: porosity profile size {porous_z.shape[0]}"
                                 f" does not match the input size {dim_d}.")
            porous_z = porous_z[:1]
        mean_k, coef = layer_tables(z, self.porous.layers, self.k_hc)
        if self.MODEL_ID == MODEL_VRETTAS_FUNG:
            if n_rnd is None:
                raise TypeError("'NoneType' object is not subscriptable")     # vrettas_fung.py:154 with n_rnd = None
            n_rnd = np.atleast_1d(np.asarray(n_rnd, dtype=float))
            if n_rnd.shape[0] < dim_d:
                raise IndexError(f"index {dim_d - 1} is out of bounds for axis 0 with size {n_rnd.shape[0]}")
            noise = np.ascontiguousarray(n_rnd[:dim_d])          # LOCAL index: position in the passed slice (:154-190)
        else:
            noise = np.zeros(dim_d)
        cols = 1 if dim_m is None else dim_m
        psi_c = L.as_f64(psi.reshape(dim_d, cols))
        out = np.empty((4, dim_d, cols))
        qinf = np.empty(cols)
        p = self._params()
        lib = L.load()
        L.check(lib.hc_plugin_eval(int(self.device), C.byref(p), dim_d, cols, L.dptr(psi_c), L.dptr(L.as_f64(porous_z)),
                                   L.dptr(L.as_f64(mean_k)), L.dptr(L.as_f64(coef)), L.dptr(L.as_f64(noise)),
                                   L.dptr(out), L.dptr(qinf)))
        shape = psi.shape
        q, K, Cm, kb = (out[k].reshape(shape) for k in range(4))
        q_inf_max = qinf if dim_m is not None else qinf[0]
        return q, K, Cm, kb, q_inf_max


class VrettasFung(HydrologicalModel):
    """Stochastic-conductivity plugin -- ``vrettas_fung.py:11-258``.  ``args[0]["n_rnd"]`` holds the N(0,1) vector;
    cell i of the passed slice uses ``n_rnd[i]`` scaled by 0.05 / 0.10 / 1.0 per layer (``:154,172,190``)."""

    MODEL_ID = MODEL_VRETTAS_FUNG

    def __call__(self, psi, z, *args):
        n_rnd = None
        if args and "n_rnd" in args[0]:
            n_rnd = args[0]["n_rnd"]
        return self._evaluate(psi, z, n_rnd)


class vanGenuchten(HydrologicalModel):
    """Mualem-van Genuchten plugin -- ``vanGenuchten.py:11-128``: no noise, ``k_bkg`` is ``sat_soil`` everywhere."""

    MODEL_ID = MODEL_VAN_GENUCHTEN

    def __call__(self, psi, z, *args):
        return self._evaluate(psi, z, None)


def make_model(name, soil, porous, k_hc, theta_res, dz, device=0):
    """``Hydrological_Model.Name`` selection of ``simulation.py:219-231``: anything that is not VRETTAS_FUNG is
    vanGenuchten."""
    cls = VrettasFung if str.upper(name) == "VRETTAS_FUNG" else vanGenuchten
    return cls(soil, porous, k_hc, theta_res, dz, device=device)
