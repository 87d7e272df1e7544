"""Ensemble driver: N stochastic realisations of one soil column on one GPU (one rank).

Per-member semantics are those of ``Simulation.run`` (``/root/reference/code/src/simulation.py:495-672``):
member k has its own noise stream, all members share forcing, tables and (by default) the
initial condition produced by the spin-up of ``Simulation.initial_conditions`` (``:389-493``).
Ranks shard members with no communication while stepping; the only collective is the final
all-reduce of the per-row water-table moments (see :func:`allreduce_moments`).
"""
import numpy as np

from .digest import inverse_retention
from .stepper import EnsembleStepper, moments_to_mean_std


def pressure_head(cols, theta):
    """Inverse van Genuchten -- ``HydrologicalModel.pressure_head`` (hydrological_model.py:43-119).

    The reference interpolates the porosity AT the grid knots here (``self.porous(z)`` with the
    full grid), i.e. ``cols.por_node``.  Returns (psi, s_eff).
    """
    theta = np.atleast_1d(np.asarray(theta, dtype=float))
    if theta.shape[0] != cols.dim_d:
        raise ValueError(f" HydrologicalModel: Input size dimensions don't match:"
                         f" {theta.shape[0]} not equal to {cols.dim_d}.")
    soil = cols.soil
    return inverse_retention(theta, cols.por_node, cols.theta.res, soil.alpha, soil.n, soil.m,
                             max(soil.epsilon, 1.0e-8), cols.dz)


def spinup_on_gpu(cols, forcing, n_rnd, device=0, burn_in=1500, flags=None, verbose=False, well_no=None):
    """``Simulation.initial_conditions`` (simulation.py:389-493) with the solves on the GPU.

    Forcing row 0, fixed noise vector, SPINUP semantics, t_span = (0, 1); stops when the
    estimated water table is within 2*dz of the first observation and the mean squared change
    of the state is <= 0.01.  Returns (psi0, iterations, early_stop).
    """
    z = cols.z
    y0, _ = pressure_head(cols, cols.por_raw)
    st = EnsembleStepper(cols, forcing, 1, device=device, flags=flags)
    try:
        st.set_state(y0)
        st.set_noise_host(np.asarray(n_rnd, dtype=float)[None, :])
        early_stop, j = False, -1
        for j in range(burn_in):
            out = st.step_rows(0, 1, spinup=True, moments=False, want_wtd=True)
            y_j = st.get_state()[0]
            wtd_est = int(out["wtd"][0, 0])
            abs_error = np.abs(forcing.zwtd_cm[0] - z[wtd_est])
            mse_0 = np.mean((y_j - y0) ** 2)
            y0 = y_j.copy()
            if abs_error <= (2.0 * cols.dz) and (mse_0 <= 0.01):
                early_stop = True
                if verbose:
                    print(f" [Initial Conditions for Well no. {well_no}]"
                          f" finished at [itr: {j}] with [abs(error): {abs_error}]"
                          f" and [MSE: {mse_0}]")
                break
        if verbose and not early_stop:
            print(f" [Initial Conditions for Well no. {well_no}]"
                  f" finished at maximum number of iterations.")
    finally:
        st.close()
    return y0, j + 1, early_stop


PHILOX_DRAW_SPINUP = 0xFFFFFFFF          # include/hydrocol.h: HC_PHILOX_DRAW_SPINUP


def spinup_members_on_gpu(stepper, cols, forcing, burn_in=1500):
    """``Simulation.initial_conditions`` for EVERY member of `stepper` at once: each member starts
    from the hydrostatic-like profile of simulation.py:409-414, keeps the noise source already installed in
    `stepper` (host vectors or its Philox stream, draw 0), and stops by its own rule (simulation.py:468), all
    inside one kernel launch.  Returns (psi0[N][D], iterations[N]); iterations < 0 marks members that ran into
    `burn_in` (the reference prints "finished at maximum number of iterations" and carries on)."""
    y0, _ = pressure_head(cols, cols.por_raw)
    stepper.set_state(y0)
    iters, _ = stepper.spinup(forcing.zwtd_cm[0], cols.z[0], forcing_row=0, max_iterations=burn_in)
    return stepper.get_state(), iters


def member_generators(seed, n_members, member_offset=0):
    """NumPy streams of a parity-style ensemble: global member 0 consumes exactly the
    reference's ``default_rng(SeedSequence(seed))`` (simulation.py:66-70); member k >= 1 uses
    ``SeedSequence(seed, spawn_key=(k,))``."""
    from numpy.random import SeedSequence, default_rng
    gens = []
    for k in range(member_offset, member_offset + n_members):
        gens.append(default_rng(SeedSequence(seed) if k == 0 else SeedSequence(seed, spawn_key=(k,))))
    return gens


class EnsembleSimulation:
    """N members of one parameter point on one device.

    noise="philox" (default): counter-based normals generated in the kernel, keyed by the global member id.
    noise="numpy": host NumPy streams (`member_generators`), drawn in the reference's order -- #0 spin-up,
    #1 base vector, then one vector per refresh row (simulation.py:426,561,601) -- and uploaded per launch.
    spinup="shared" (default): one spin-up (global member 0's first draw) broadcast to all members;
    spinup="member": every member spins up with its own first draw (`spinup_members_on_gpu`).
    """

    def __init__(self, cols, forcing, n_members, seed=0, device=0, member_offset=0, psi0=None, flags=None,
                 noise="philox", spinup="shared"):
        if noise not in ("philox", "numpy") or spinup not in ("shared", "member"):
            raise ValueError(f" {self.__class__.__name__}: unknown noise / spinup mode ({noise}, {spinup}).")
        self.cols, self.forcing = cols, forcing
        self.n_members = int(n_members)
        self.member_offset = int(member_offset)
        self.seed = int(seed)
        self.device = device
        self.noise = noise
        self.spinup_iters = None
        if noise == "numpy":
            self._init_numpy(psi0, flags, spinup)
            return
        if psi0 is None and spinup == "member":
            self.stepper = EnsembleStepper(cols, forcing, self.n_members, device=device, flags=flags)
            self.stepper.set_noise_philox(self.seed, self.member_offset)
            self.psi0, self.spinup_iters = spinup_members_on_gpu(self.stepper, cols, forcing)
            # the x0.8 damping a spin-up retry applied belongs to the spin-up vector, not to the run's base vector
            self.stepper.set_noise_philox(self.seed, self.member_offset)
            self.next_row, self.kernel_ms, self.launches = 1, 0.0, 0
            return
        if psi0 is None:
            # shared initial condition: spin-up with the noise vector of global member 0, draw 0
            probe = EnsembleStepper(cols, forcing, 1, device=device, flags=flags)
            probe.set_noise_philox(self.seed, 0)
            n_rnd = probe.philox_normals(0, PHILOX_DRAW_SPINUP)     # global member 0's spin-up vector
            probe.close()
            psi0, self.spinup_iters, _ = spinup_on_gpu(cols, forcing, n_rnd, device=device, flags=flags)
        self.psi0 = np.asarray(psi0, dtype=float)
        self.stepper = EnsembleStepper(cols, forcing, self.n_members, device=device, flags=flags)
        self.stepper.set_state(self.psi0)
        self.stepper.set_noise_philox(self.seed, self.member_offset)
        self.next_row = 1
        self.kernel_ms = 0.0
        self.launches = 0

    def _init_numpy(self, psi0, flags, spinup):
        cols, forcing, N, D = self.cols, self.forcing, self.n_members, self.cols.dim_d
        self.gens = member_generators(self.seed, N, self.member_offset)
        self.stepper = EnsembleStepper(cols, forcing, N, device=self.device, flags=flags)
        if psi0 is None:
            first = np.stack([g.standard_normal(D) for g in self.gens])           # draw #0 (simulation.py:426)
            if spinup == "member":
                self.stepper.set_noise_host(first)
                psi0, self.spinup_iters = spinup_members_on_gpu(self.stepper, cols, forcing)
            else:
                if self.member_offset == 0:
                    lead = first[0]
                else:       # every shard spins up with global member 0's first draw
                    lead = member_generators(self.seed, 1, 0)[0].standard_normal(D)
                psi0, self.spinup_iters, _ = spinup_on_gpu(cols, forcing, lead, device=self.device, flags=flags)
        self.psi0 = np.asarray(psi0, dtype=float)
        self.stepper.set_state(self.psi0)
        self.stepper.set_noise_host(np.stack([g.standard_normal(D) for g in self.gens]))   # draw #1 (:561)
        self.next_row, self.kernel_ms, self.launches = 1, 0.0, 0

    def advance(self, n_rows, **kw):
        """Solve the next ``n_rows`` forcing rows for every member."""
        if self.noise == "numpy":
            n_fresh = int(self.forcing.refresh[self.next_row:self.next_row + n_rows].sum())
            fresh = np.empty((n_fresh, self.n_members, self.cols.dim_d))
            for q in range(n_fresh):                 # row order, one vector per member per refresh row (:601)
                for k, g in enumerate(self.gens):
                    fresh[q, k] = g.standard_normal(self.cols.dim_d)
            kw["fresh_noise"] = fresh
        out = self.stepper.step_rows(self.next_row, n_rows, **kw)
        self.next_row += n_rows
        self.kernel_ms += out["kernel_ms"]
        self.launches += out["launches"]
        return out

    def moments(self):
        return self.stepper.moments()

    def wtd_mean_std(self, moments=None):
        m = self.moments() if moments is None else moments
        return moments_to_mean_std(m, self.cols.dz, self.cols.z[0])

    # -- checkpoint / resume (the single-column analogue in the reference is IC_Filename, simulation.py:358-385) ------
    CHECKPOINT_KEYS = ("psi", "noise_scale", "moments", "next_row", "seed", "member_offset", "n_members", "dim_d",
                       "dim_t", "initial_cond")

    def dump(self, path):
        """Everything a stopped Philox ensemble is defined by, in the results container (HDF5 through libhdf5, ``.npz``
        where no libhdf5 loads): member states, per-member damping of the base noise vector, the moment table, the
        next forcing row and the stream keys.  ``restore`` continues bit for bit."""
        if self.noise != "philox":
            raise ValueError(" EnsembleSimulation: dump/restore serves the Philox noise source "
                             "(a NumPy stream's position is not part of the stepper).")
        from pathlib import Path
        from . import hdf5io
        arrays = dict(psi=self.stepper.get_state(), noise_scale=self.stepper.noise_scale(),
                      moments=np.asarray(self.stepper.moments()), next_row=np.array(self.next_row, dtype=np.int64),
                      seed=np.array(self.seed, dtype=np.uint64), member_offset=np.array(self.member_offset, dtype=np.int64),
                      n_members=np.array(self.n_members, dtype=np.int64), dim_d=np.array(self.cols.dim_d, dtype=np.int64),
                      dim_t=np.array(self.forcing.dim_t, dtype=np.int64), initial_cond=np.asarray(self.psi0, dtype=float))
        path = Path(path)
        if hdf5io.available() and path.suffix != ".npz":
            hdf5io.write(path, arrays)
        else:
            path = path.with_suffix(".npz")
            np.savez(path, **arrays)
        return path

    @classmethod
    def restore(cls, path, cols, forcing, device=0, flags=None):
        """A new ensemble (new handle) continuing the one ``dump`` wrote: same members, same streams, same row."""
        from pathlib import Path
        from . import hdf5io
        path = Path(path)
        data = dict(np.load(path)) if path.suffix == ".npz" else hdf5io.read(path)
        missing = [k for k in cls.CHECKPOINT_KEYS if k not in data]
        if missing:
            raise ValueError(f" EnsembleSimulation: {path} is not an ensemble checkpoint (missing {missing}).")
        n, D, T = int(data["n_members"]), int(data["dim_d"]), int(data["dim_t"])
        if D != cols.dim_d or T != forcing.dim_t:
            raise ValueError(f" EnsembleSimulation: checkpoint of a [{D}]-node column over {T} rows does not fit "
                             f"this run ([{cols.dim_d}], {forcing.dim_t}).")
        psi = np.asarray(data["psi"], dtype=float).reshape(n, D)
        sim = cls(cols, forcing, n, seed=int(data["seed"]), device=device, member_offset=int(data["member_offset"]),
                  psi0=np.asarray(data["initial_cond"], dtype=float).reshape(-1)[:D], flags=flags)
        sim.stepper.set_state(psi if n > 1 else psi[0])
        sim.stepper.set_noise_scale(np.asarray(data["noise_scale"], dtype=float).reshape(n))
        sim.stepper.set_moments(np.asarray(data["moments"], dtype=np.int64))
        sim.next_row = int(data["next_row"])
        return sim

    def close(self):
        self.stepper.close()


def allreduce_moments(moments, device, force=False):
    """Sum the int64 moment table ([3][T], or [P][3][T] for P parameter points) over all ranks -- the one collective
    of the path (RCCL over xGMI when the backend is nccl).

    Integer sums are exact and order-independent, so the ensemble mean / sigma are bitwise identical at any GPU
    count.  With the nccl backend the table is reduced in device memory (one host->device copy of the table the
    library handed over, ``all_reduce`` on the device tensor, one copy back).  No-op when torch.distributed is not
    initialised or the world has one rank, unless ``force`` asks for the collective anyway (a world-size-1 process
    group exercises the RCCL call path on a single GPU)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return moments
    if dist.get_world_size() == 1 and not force:
        return moments
    t = torch.from_numpy(np.ascontiguousarray(moments))
    if dist.get_backend() == "nccl":
        t = t.to(device, non_blocking=False)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def allreduce_stepper_moments(stepper, device, force=False):
    """The run's one collective, on device memory end to end: the stepper's moment table ([3][T] or [P][3][T]) is copied
    device-to-device into a torch tensor (``hc_export_moments``), summed over the ranks with ``all_reduce`` (RCCL when
    the backend is nccl) and only then brought to the host.  Other backends (gloo rehearsals) and single-rank runs go
    through :func:`allreduce_moments` on the host copy."""
    import torch
    import torch.distributed as dist
    active = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force)
    if not active or dist.get_backend() != "nccl":
        return allreduce_moments(stepper.moments(), device, force=force)
    shape = (3, stepper.T) if stepper.P == 1 else (stepper.P, 3, stepper.T)
    t = torch.empty(shape, dtype=torch.int64, device=device)
    stepper.export_moments(t.data_ptr())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def merge_parameters(params, override):
    """``params`` with the sections of ``override`` ({"Soil_Properties": {...}, ...}) merged in (a deep copy)."""
    import copy
    p = copy.deepcopy(params)
    for section, values in override.items():
        if isinstance(values, dict):
            p.setdefault(section, {}).update(values)
        else:
            p[section] = values
    return p


# What a parameter point may NOT change: the forcing digest (ET series, surface evaporation: simulation.py:273-352) is
# built once from the base parameters and shared by every point of a sweep, and the PREDICT gate is taken on the base.
SWEEP_SHARED_ENVIRONMENTAL = ("Atmospheric_Demand", "Wet_Season_pct", "Evaporation_pct")


def check_sweep_points(params, points):
    """Refuse parameter points that would silently run with the base point's forcing: an override of the
    ``Environmental`` keys the forcing digest reads, of ``Simulation_Flags.PREDICT`` (the repair gate is taken once, on
    the base), of the well or of the data file.  Returns the merged parameter dicts, one per point."""
    merged = [merge_parameters(params, ov) for ov in points]
    for k, mp in enumerate(merged):
        for key in SWEEP_SHARED_ENVIRONMENTAL:
            if mp["Environmental"].get(key) != params["Environmental"].get(key):
                raise ValueError(f" Sweep: point {k} overrides Environmental.{key}; every point of a sweep shares the base "
                                 f"forcing series (simulation.py:273-352) -- run such points as separate ensembles.")
        if bool(mp["Simulation_Flags"].get("PREDICT", False)) != bool(params["Simulation_Flags"].get("PREDICT", False)):
            raise ValueError(f" Sweep: point {k} overrides Simulation_Flags.PREDICT; the flag is taken from the base parameters.")
        for key in ("Well_No", "Site_Information", "Data_Filename"):
            if mp.get(key) != params.get(key):
                raise ValueError(f" Sweep: point {k} overrides {key}; every point shares the well and the forcing file.")
    return merged


class SweepSimulation:
    """BASELINE config 5: P parameter points x ``n_members`` stochastic members each, ALL in one handle and one
    launch per batch of rows (``hc_add_point``): per-member parameter point -> its own column parameters and slot
    tables in the step kernel, per-point moments.

    Global member ids are point-major over the WHOLE sweep: point k (global index ``first_point + j`` for the j-th
    point of this handle) owns members [k n, (k + 1) n) of the Philox stream, so a point's realisations do not depend
    on which rank or handle runs it.  Every point starts from its OWN spin-up (field capacity, wilting point and the
    equilibrium profile depend on the point: porosity.py:172-181, simulation.py:389-493): the P spin-ups run
    together in one ``hc_spinup`` launch (one member per point, that point's lead member's spin-up vector), and the
    result is broadcast to the point's members."""

    def __init__(self, cols_list, forcing, n_members, seed=0, device=0, first_point=0, flags=None, psi0=None,
                 point_ids=None):
        self.points = list(cols_list)
        self.P, self.n = len(self.points), int(n_members)
        self.forcing, self.seed, self.device = forcing, int(seed), device
        # global index of each of this handle's points in the whole sweep: consecutive from `first_point`, or any
        # list (`point_ids`) when points are dealt to ranks round-robin so that every rank gets the same mix of costs
        self.point_ids = (np.arange(self.P, dtype=np.int64) + int(first_point) if point_ids is None
                          else np.asarray(point_ids, dtype=np.int64))
        if self.point_ids.shape != (self.P,) or np.unique(self.point_ids).size != self.P or self.point_ids.min() < 0:
            raise ValueError(" SweepSimulation: point_ids must name each of the handle's points once.")
        self.bases = self.point_ids * self.n
        self.member_offset = int(self.bases[0])
        cols = self.points[0]
        self.cols = cols
        self.spinup_iters = None
        if psi0 is None:
            psi0, self.spinup_iters = self._spinup(flags)
        self.psi0 = np.asarray(psi0, dtype=float).reshape(self.P, cols.dim_d)
        self.stepper = EnsembleStepper(self.points, forcing, self.P * self.n, device=device, flags=flags)
        # a sweep always runs the generic-exponent cell model: a point that happens to sit on the default exponents
        # must not change its bits with the company it is stepped in
        self.stepper.set_generic_exponents(True)
        self.stepper.set_state(self.psi0 if self.P > 1 else self.psi0[0])
        self.stepper.set_noise_philox(self.seed, self.member_offset)
        if self.P > 1:
            self.stepper.set_point_member_bases(self.bases)
        self.next_row, self.kernel_ms, self.launches = 1, 0.0, 0

    def _spinup(self, flags):
        cols, forcing, P, D = self.cols, self.forcing, self.P, self.cols.dim_d
        lead = EnsembleStepper(self.points, forcing, P, device=self.device, flags=flags)
        try:
            lead.set_generic_exponents(True)
            lead.set_noise_philox(self.seed, 0)
            noise = np.stack([lead.philox_normals(int(self.bases[j]), PHILOX_DRAW_SPINUP) for j in range(P)])
            start = np.stack([pressure_head(c, c.por_raw)[0] for c in self.points])
            lead.set_state(start if P > 1 else start[0])
            lead.set_noise_host(noise)
            iters, _ = lead.spinup(forcing.zwtd_cm[0], cols.z[0], forcing_row=0, max_iterations=1500)
            return lead.get_state(), iters
        finally:
            lead.close()

    def advance(self, n_rows, **kw):
        out = self.stepper.step_rows(self.next_row, n_rows, **kw)
        self.next_row += n_rows
        self.kernel_ms += out["kernel_ms"]
        self.launches += out["launches"]
        return out

    def moments(self):
        """[P][3][T]"""
        return np.asarray(self.stepper.moments()).reshape(self.P, 3, self.forcing.dim_t)

    def close(self):
        self.stepper.close()


def deal_points(n_points, rank, world):
    """Global indices of the parameter points rank `rank` of `world` runs: round-robin."""
    return list(range(int(rank), int(n_points), int(world)))


def parameter_sweep(params, data, well, points, n_members, n_rows, seed=0, device=0, rank=0, world=1,
                    one_launch=True, rows_per_call=48 * 8):
    """BASELINE config 5: a grid of (n, a0, psi_sat, ...) points x ``n_members`` realisations each.

    ``points`` is a list of dicts ``{"Soil_Properties": {...}, "Hydraulic_Conductivity": {...}, ...}`` merged over
    ``params``; every point gets its own tables and its own spin-up.  Whole points are dealt to ranks round-robin
    (:func:`deal_points`: rank r owns points r, r + world, ... -- a grid's cost grows along its slowest axis, so
    contiguous blocks would hand one rank all the expensive points), with no communication; a point's members keep
    their global ids (point k owns members [k n, (k + 1) n) of the Philox stream) whoever runs it.  ``one_launch`` (default)
    steps all of a rank's points in one handle (:class:`SweepSimulation`); ``one_launch=False`` runs them one after
    another, one handle each -- same global member ids, bit-identical results, kept as the cross-check.
    Returns {point index: {"moments", "wtd_mean_cm", "wtd_std_cm", "psi0"}}.
    """
    from .digest import ColumnTables, ForcingDigest
    P = len(points)
    mine = deal_points(P, rank, world)
    if not mine:
        return {}
    merged = check_sweep_points(params, points)
    cols_all = [ColumnTables(merged[k], well) for k in mine]
    forcing = ForcingDigest(params, data, cols_all[0])
    groups = [(mine, cols_all)] if one_launch else [([k], [c]) for k, c in zip(mine, cols_all)]
    out = {}
    for ids, cols_list in groups:
        sim = SweepSimulation(cols_list, forcing, n_members, seed=seed, device=device, point_ids=ids)
        done = 0
        while done < n_rows:
            n = min(rows_per_call, n_rows - done)
            sim.advance(n)
            done += n
        m = sim.moments()
        for j, c in enumerate(cols_list):
            mean_cm, std_cm = moments_to_mean_std(m[j], c.dz, c.z[0])
            out[ids[j]] = {"moments": m[j], "wtd_mean_cm": mean_cm, "wtd_std_cm": std_cm, "psi0": sim.psi0[j],
                              "spinup_iterations": None if sim.spinup_iters is None else int(sim.spinup_iters[j]),
                              "kernel_ms": sim.kernel_ms}
        sim.close()
    return out
