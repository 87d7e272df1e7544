"""Round-3 GPU checks: every G4 state straight against the reference's solve, the whole year side by side (reference /
oracle / GPU), BASELINE config 5 at its full size, bit-exact ensemble resume, the in-library RCCL all-reduce."""
import copy

import numpy as np
import pytest

from helpers import WELLS, digest, forcing_frame, golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("these tests need a GPU")
    import __graft_entry__ as ge
    ge.build()
    from hydromodel_amd import stepper
    return stepper


def forcing_with_row(forcing, row, precip, atm, daylight, wtd_obs):
    """A copy of the digest whose row `row` carries the given arguments (the golden states were evaluated with
    explicit (hour, precip, atm, wtd) rather than a row of the synthetic forcing); never a refresh row."""
    f = copy.copy(forcing)
    for name in ("precip", "atm", "daylight", "wtd_obs", "refresh", "wet_season"):
        setattr(f, name, np.array(getattr(forcing, name)))
    f.precip[row], f.atm[row], f.daylight[row], f.wtd_obs[row], f.refresh[row] = precip, atm, int(daylight), wtd_obs, 0
    return f


@pytest.mark.parametrize("well", [1, 200, 300, 401, 581])
def test_every_constructed_state_matches_the_reference_solve(gpu, well):
    """G4 on the GPU directly against the reference: all 13 constructed states the reference
    solved in one attempt (the 14th, HLIFT at night, takes > 1 000 RHS evaluations and is chaotic: the CPU suite skips it
    too) x 5 wells -- among them the well the reference's input_parameters.json selects (no. 10, D = 401, 7 cells per lane)
    and its deepest (no. 14, D = 581), which the SPLIT-COLUMN kernel serves --, `RichardsPDE.solve` over t_span (7, 8) (richards_pde.py:478-537) -> forcing row 8 carrying the
    state's own (hour, precip, atm, wtd) arguments and flags.  Tiers: a regular row (<= 100 RHS evaluations) must
    reproduce the reference's nfev/njev/nlu/steps and agree to 1e-6 (1 + |psi|); stiff constructed states
    (~200 evaluations, ~75 steps, Jacobian refreshed up to 10 times) decorrelate in the last bits of the FD Jacobian
    and are held to the integrator's accuracy class, 5e-2 (1 + |psi|)  (DESIGN.md §3)."""
    _, cols, forcing = digest(well)
    g = golden(f"g34_states_{well}.npz")
    tiers = {"<1e-9": 0, "<1e-6": 0, "stiff <5e-2": 0}
    same = total = regular = 0
    for name in g["names"]:
        name = str(name)
        ref_stats = g[f"{name}_solve_stats"]
        if name == "hlift_night" or ref_stats.shape[0] != 1:
            continue
        fl = g[f"{name}_flags"]
        hour = int(g[f"{name}_hour"])
        f8 = forcing_with_row(forcing, 8, float(g[f"{name}_precip"]), float(g[f"{name}_atm"]), 6 <= hour <= 17,
                              int(g["wtd_idx"]))
        st = gpu.EnsembleStepper(cols, f8, 1, flags={"ET": bool(fl[1]), "LF": bool(fl[2]), "HLIFT": bool(fl[3])})
        st.set_state(g[f"{name}_y"][None, :])
        st.set_noise_host(g["n_rnd"][None, :])
        out = st.step_rows(8, 1, fresh_noise=np.zeros((0,)), spinup=bool(fl[0]), moments=False, want_stats=True)
        y1 = st.get_state()[0]
        noise_after = st.get_noise_base()[0]
        st.close()
        ry = g[f"{name}_solve_y"]
        err = float(np.max(np.abs(y1 - ry) / (1 + np.abs(ry))))
        is_regular = int(ref_stats[0, 0]) <= 100
        got = out["stats"][0, 0, :5].tolist()
        total += 1
        regular += is_regular
        same += got == ref_stats[0, :5].tolist()
        tiers["<1e-9" if err < 1e-9 else ("<1e-6" if err < 1e-6 else "stiff <5e-2")] += 1
        assert got[4] == 1 and np.array_equal(noise_after, g[f"{name}_solve_nrnd_after"]), name     # one attempt, noise untouched
        if is_regular:
            assert err < 1e-6 and got == ref_stats[0, :5].tolist(), (well, name, err, got, ref_stats[0])
        else:
            assert err < 5e-2, (well, name, err, got, ref_stats[0])
    print(f"[well {well}] G4 on the GPU vs the reference's solves: {same}/{total} states with the reference's "
          f"nfev/njev/nlu/steps/attempts ({regular} regular, all of them exact); error tiers {tiers}")
    assert total == 13 and same >= regular


def test_whole_year_reference_oracle_and_gpu_side_by_side(gpu):
    """The reference's year (G5, well 1, seed 911) against the C oracle AND the GPU, both free-running from the
    reference's initial condition on the reference's own noise stream (default_rng(SeedSequence(911)): draw #0 spin-up,
    #1 base, one per refresh row -- simulation.py:426,561,601): water-table index on all 17 519 solved rows.  The rows on
    which the solver gives up (x0.8 damping, richards_pde.py:522) are chaotic events that land on different rows in the
    three implementations and rescale the base noise for the rest of the year, so equality decays over the year in
    each pair; nobody is ever more than one 5-cm cell from the reference."""
    from oracle.oracle import Oracle
    _, cols, forcing = digest(1)
    g = golden("g5_traj_1.npz")
    D, T = cols.dim_d, forcing.dim_t
    rng = np.random.default_rng(np.random.SeedSequence(911))
    rng.standard_normal(D)
    base = rng.standard_normal(D)
    n_ref = int(forcing.refresh.sum())
    fresh = np.array([rng.standard_normal(D) for _ in range(n_ref)])
    ref_idx = np.rint(g["wtd_est_cm"] / cols.dz).astype(int)
    o = Oracle(cols, forcing.surface_evap)
    r = o.run(forcing, g["initial_cond"], base, fresh, 1, T, want_stats=True)
    st = gpu.EnsembleStepper(cols, forcing, 1)
    st.set_state(g["initial_cond"])
    st.set_noise_host(base[None, :])
    out = st.step_rows(1, T - 1, fresh_noise=fresh[:, None, :], want_wtd=True, want_stats=True)
    st.close()
    gpu_idx, orc_idx = out["wtd"][:, 0], r["wtd_est"][1:]
    d_gr, d_or, d_go = (np.abs(a - b) for a, b in ((gpu_idx, ref_idx[1:]), (orc_idx, ref_idx[1:]), (gpu_idx, orc_idx)))
    retried = {"reference": int((g["per_row_stats"][:, 4] > 1).sum()), "oracle": int((r["per_row"][:, 4] > 1).sum()),
               "gpu": int((out["stats"][:, 0, 4] > 1).sum())}
    print(f"whole year, water-table index equal on: GPU vs reference {(d_gr == 0).mean():.1%}, oracle vs reference "
          f"{(d_or == 0).mean():.1%}, GPU vs oracle {(d_go == 0).mean():.1%} of {d_gr.size} rows; max distance "
          f"{d_gr.max()} / {d_or.max()} / {d_go.max()} cells; rows that needed a retry: {retried}")
    assert d_gr.max() <= 1 and d_or.max() <= 1 and d_go.max() <= 1
    assert (d_gr[:1400] == 0).mean() > 0.98 and (d_or[:1400] == 0).mean() > 0.98
    # SURVEY §8c asked for >= 99 %: not reachable for ANY implementation once the retry rows differ (see the docstring;
    # DESIGN.md §3).  The gate is therefore RELATIVE to the faithful CPU restatement on this very run: the GPU may share
    # at most 3 points less of the year with the reference than the oracle does (measured 96.4 % against 97.2 %), and the
    # two of them agree with each other at least as well as the worse of them agrees with the reference, less 3 points.
    e_gr, e_or, e_go = (float((d == 0).mean()) for d in (d_gr, d_or, d_go))
    assert e_or >= 0.95                          # measured 97.2 %
    assert e_gr >= e_or - 0.03, (e_gr, e_or)
    assert e_go >= min(e_gr, e_or) - 0.03, (e_go, e_gr, e_or)


def test_whole_year_equality_is_a_distribution_not_a_number(gpu):
    """How much of the year's water-table index an implementation shares with the reference is decided by where its
    handful of give-up rows fall -- a last-bit matter (the reference's OWN statistics on such rows move when nothing but
    SuperLU's elimination order changes: tests/golden/superlu_order_check.py).  Twelve GPU members and twelve oracle runs
    start from the reference's initial condition perturbed by 1e-13 (relative) and consume the reference's own noise
    stream; every one of them stays within one cell of the reference on every row.  The acceptance is a TWO-SAMPLE one
    (VERDICT r3 item 5b): the GPU's median equality figure must lie inside the oracle's own [min, max] +- 0.03 -- a kernel
    that scatters differently from a faithful restatement of the algorithm fails, whatever the absolute numbers are.
    (One unperturbed GPU build measured 96.4 %, another -- an RHS with a different summation order, parity-green on
    every pinned row -- 75.3 %; the oracle itself 97.2 %.)"""
    from concurrent.futures import ThreadPoolExecutor
    from oracle.oracle import Oracle
    _, cols, forcing = digest(1)
    g = golden("g5_traj_1.npz")
    D, T = cols.dim_d, forcing.dim_t
    rng = np.random.default_rng(np.random.SeedSequence(911))
    rng.standard_normal(D)
    base = rng.standard_normal(D)
    n_ref = int(forcing.refresh.sum())
    fresh = np.array([rng.standard_normal(D) for _ in range(n_ref)])
    ref_idx = np.rint(g["wtd_est_cm"] / cols.dz).astype(int)[1:]
    N = 12
    pert = np.random.default_rng(5).standard_normal((N, D))
    pert[0] = 0.0
    y0 = g["initial_cond"][None, :] * (1.0 + 1e-13 * pert)
    st = gpu.EnsembleStepper(cols, forcing, N)
    st.set_state(y0)
    st.set_noise_host(np.tile(base, (N, 1)))
    out = st.step_rows(1, T - 1, fresh_noise=np.repeat(fresh[:, None, :], N, axis=1), want_wtd=True, want_stats=True)
    st.close()
    d_gpu = np.abs(out["wtd"] - ref_idx[:, None])                       # [rows][members]
    eq_gpu = (d_gpu == 0).mean(axis=0)
    retried_gpu = (out["stats"][:, :, 4] > 1).sum(axis=0)

    def oracle_run(k):
        o = Oracle(cols, forcing.surface_evap)
        r = o.run(forcing, y0[k], base, fresh, 1, T, want_stats=True)
        d = np.abs(r["wtd_est"][1:] - ref_idx)
        return float((d == 0).mean()), int(d.max()), int((r["per_row"][:, 4] > 1).sum())

    with ThreadPoolExecutor(max_workers=12) as ex:
        orc = list(ex.map(oracle_run, range(N)))
    eq_orc = np.array([e for e, _, _ in orc])
    print(f"year-long equality with the reference under 1e-13 perturbations of the initial state: GPU (12 members) "
          f"min {eq_gpu.min():.1%} median {np.median(eq_gpu):.1%} max {eq_gpu.max():.1%}, retried rows "
          f"{retried_gpu.min()}..{retried_gpu.max()}; oracle ({N} runs) min {eq_orc.min():.1%} median {np.median(eq_orc):.1%} "
          f"max {eq_orc.max():.1%}, retried rows {min(r for _, _, r in orc)}..{max(r for _, _, r in orc)}; reference 14")
    assert d_gpu.max() <= 1 and max(m for _, m, _ in orc) <= 1            # never more than one cell from the reference
    assert (d_gpu[:1400] == 0).mean() > 0.98                              # the first month: before any give-up row
    # two-sample: the GPU scatters like the oracle does
    assert eq_orc.min() - 0.03 <= np.median(eq_gpu) <= eq_orc.max() + 0.03, (eq_gpu, eq_orc)
    assert eq_gpu.min() >= eq_orc.min() - 0.15, (eq_gpu, eq_orc)          # ... and no single member falls out of that range
    assert abs(int(np.median(retried_gpu)) - int(np.median([r for _, _, r in orc]))) <= 8   # give-up rows: same count class


def _sweep_points(k=8):
    return [{"Soil_Properties": {"n": float(n), "a0": float(a0), "psi_sat": float(ps)}}
            for n in np.linspace(1.5, 3.0, k) for a0 in np.geomspace(0.003, 0.03, k) for ps in -np.geomspace(1e-3, 1.0, k)]


def test_config5_at_full_size_in_one_handle(gpu):
    """BASELINE configs[4] at its size (VERDICT r2 missing 2): the 8 x 8 x 8 (n, a0, psi_sat) grid of SURVEY.md §8d x
    4 096 members x D = 300 in ONE handle (5 GB of state), every point from its own spin-up, one simulated day in one
    launch.  Size-independent properties: every point counts 4 096 members on every row; three points spread over the
    grid are bit-equal -- states and moments -- to stand-alone handles running the same global member ids; members of
    two points (one mild, one from the costly corner) agree with the CPU ORACLE fed the same Philox normals, so the
    multi-point launch is checked against something other than itself."""
    from hydromodel_amd.digest import ColumnTables, ForcingDigest
    from hydromodel_amd.ensemble import PHILOX_DRAW_SPINUP, SweepSimulation, check_sweep_points
    from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
    from oracle.oracle import Oracle
    params = default_parameters()
    merged = check_sweep_points(params, _sweep_points())
    well = synthetic_well(300)
    cols_all = [ColumnTables(mp, well) for mp in merged]
    forcing = ForcingDigest(params, synthetic_forcing_frame(1), cols_all[0])
    P, M, D, rows = len(cols_all), 4096, 300, 48
    assert P == 512
    big = SweepSimulation(cols_all, forcing, M, seed=17)
    assert (np.asarray(big.spinup_iters) > 0).all()                      # every spin-up met its stop rule
    big.advance(rows)
    m = big.moments()
    assert m.shape == (P, 3, forcing.dim_t)
    assert np.array_equal(m[:, 0, 1:1 + rows], np.full((P, rows), M))
    assert (m[:, 0, 1 + rows:] == 0).all() and (m[:, 0, 0] == 0).all()
    cost = big.stepper.point_costs()
    assert cost.shape == (P,) and (cost >= M * rows * 8).all()            # >= 8 RHS evaluations per column-step everywhere
    cnt = big.stepper.counters()
    windows = (3, 260, 509)
    states = {j: big.stepper.get_state(j * M, M) for j in windows}
    checks = {j: big.stepper.get_state(j * M, 2) for j in (219, 405)}
    psi0 = big.psi0.copy()
    big.close()
    for j in windows:
        assert np.isfinite(states[j]).all()
        one = SweepSimulation([cols_all[j]], forcing, M, seed=17, first_point=j, psi0=psi0[j])
        one.advance(rows)
        assert np.array_equal(one.stepper.get_state(), states[j]), j
        assert np.array_equal(one.moments()[0], m[j]), j
        one.close()
    # the multi-point launch against the oracle: members 0 and 1 of two points, fed the Philox normals the kernel drew
    worst = 0.0
    for j in (219, 405):                                   # (n, a0, psi_sat) = (2.14, 0.008, -0.019) and (2.79, 0.006, -0.14)
        c = cols_all[j]
        o = Oracle(c, forcing.surface_evap)
        probe = gpu.EnsembleStepper(c, forcing, 1)
        probe.set_noise_philox(17, 0)
        n_fresh = int(forcing.refresh[1:1 + rows].sum())
        for k in range(2):
            gid = j * M + k
            base = probe.philox_normals(gid, 0)
            fresh = np.stack([probe.philox_normals(gid, q + 1) for q in range(n_fresh)]) if n_fresh else np.zeros((0, D))
            r = o.run(forcing, psi0[j], base, fresh, 1, 1 + rows)
            e = float(np.max(np.abs(checks[j][k] - r["psi"]) / (1 + np.abs(r["psi"]))))
            worst = max(worst, e)
            # 48 chained rows through the generic-exponent cell model (in-house exp/log against the oracle's libm pow)
            assert e < 5e-3, (j, k, e)
        probe.close()
    print(f"config 5 at full size: 512 x 4096 x D=300, one day in one launch; failed attempts {cnt['failed_attempts']}, "
          f"guard trips {cnt['guard_trips']}; RHS evaluations per column-step by point: min {cost.min() / (M * rows):.1f} "
          f"median {np.median(cost) / (M * rows):.1f} max {cost.max() / (M * rows):.1f}; worst member vs the oracle after "
          f"48 rows {worst:.2e}")


def test_multi_point_launch_with_a_default_exponent_point_against_the_oracle(gpu):
    """A sweep that contains the reference's own point (n = 2, a0 = 0.009, psi_sat = -0.0047, where the oracle is pinned by
    G1-G5) next to three others: its members inside the multi-point launch against the oracle, 96 chained rows."""
    from hydromodel_amd.digest import ColumnTables, ForcingDigest
    from hydromodel_amd.ensemble import SweepSimulation, check_sweep_points
    from hydromodel_amd.synthetic import default_parameters
    from oracle.oracle import Oracle
    params = default_parameters()
    pts = [{"Soil_Properties": {"n": 1.7, "a0": 0.02}}, {"Soil_Properties": {"n": 2.0}},
           {"Soil_Properties": {"n": 2.6, "a0": 0.004, "psi_sat": -0.5}}, {"Soil_Properties": {"a0": 0.015}}]
    cols_all = [ColumnTables(mp, WELLS[200]) for mp in check_sweep_points(params, pts)]
    forcing = ForcingDigest(params, forcing_frame(1), cols_all[0])
    ic = golden("g1_tables_200.npz")["initial_cond"]
    M, rows = 16, 96
    big = SweepSimulation(cols_all, forcing, M, seed=23, psi0=np.tile(ic, (4, 1)))
    out = big.advance(rows, want_wtd=True, want_psi=True)
    probe = gpu.EnsembleStepper(cols_all[1], forcing, 1)
    probe.set_noise_philox(23, 0)
    o = Oracle(cols_all[1], forcing.surface_evap)
    n_fresh = int(forcing.refresh[1:1 + rows].sum())
    equal = 0
    for k in range(3):
        gid = 1 * M + k
        base = probe.philox_normals(gid, 0)
        fresh = np.stack([probe.philox_normals(gid, q + 1) for q in range(n_fresh)])
        r = o.run(forcing, ic, base, fresh, 1, 1 + rows, want_psi=True)
        want = r["psi_rows"][1:1 + rows]
        e = np.max(np.abs(out["psi"][:, gid, :] - want) / (1 + np.abs(want)), axis=1)
        assert e[0] < 1e-8 and e.max() < 5e-3, (k, e[0], e.max())
        equal += int((out["wtd"][:, gid] == r["wtd_est"][1:1 + rows]).sum())
    assert equal >= 0.98 * 3 * rows
    probe.close()
    big.close()


def test_point_walk_order_changes_no_result(gpu, monkeypatch):
    """From the second launch on the chunk ticket walks the points costliest-first; HYDROCOL_POINT_ORDER=fixed keeps
    point order.  Same states, same moments, same per-point costs."""
    from hydromodel_amd.digest import ColumnTables, ForcingDigest
    from hydromodel_amd.ensemble import SweepSimulation, check_sweep_points
    from hydromodel_amd.synthetic import default_parameters
    params = default_parameters()
    pts = [{"Soil_Properties": {"n": n, "a0": a0}} for n, a0 in ((1.6, 0.004), (2.9, 0.03), (2.0, 0.009), (2.5, 0.02),
                                                                   (1.8, 0.012))]
    cols_all = [ColumnTables(mp, WELLS[200]) for mp in check_sweep_points(params, pts)]
    forcing = ForcingDigest(params, forcing_frame(1), cols_all[0])
    ic = golden("g1_tables_200.npz")["initial_cond"]
    res = []
    for fixed in (False, True):
        if fixed:
            monkeypatch.setenv("HYDROCOL_POINT_ORDER", "fixed")
        sim = SweepSimulation(cols_all, forcing, 40, seed=3, psi0=np.tile(ic, (5, 1)))
        sim.stepper.set_rows_per_launch(24)
        sim.advance(96)                                   # four launches: three of them in cost order
        res.append((sim.stepper.get_state(), sim.moments(), sim.stepper.point_costs()))
        assert sim.launches == 4
        sim.close()
    monkeypatch.delenv("HYDROCOL_POINT_ORDER")
    for a, b in zip(*res):
        assert np.array_equal(a, b)
    cost = res[0][2]
    print("RHS evaluations per column-step by point:", np.round(cost / (40 * 96.0), 1).tolist())
    assert cost.min() >= 40 * 96 * 8


def test_ensemble_resume_is_bit_exact(gpu, tmp_path):
    """30 days in one go == 10 days, dump, NEW handle, restore, 20 days: states, moments, damping factors, counters
    (VERDICT r2 missing 3; the reference's single-column analogue is IC_Filename, simulation.py:358-385).  Members start
    with different damping factors and a lowered iteration budget adds failed attempts on hard rows."""
    from hydromodel_amd.ensemble import EnsembleSimulation
    _, cols, forcing = digest(200)
    ic = golden("g1_tables_200.npz")["initial_cond"]
    N, budget = 1536, 60

    def start():
        sim = EnsembleSimulation(cols, forcing, N, seed=41, member_offset=7000, psi0=ic)
        sim.stepper.set_iteration_budget(budget)
        # members enter with 0..3 earlier failures behind them: the damping state is not trivial whatever the first days bring
        sim.stepper.set_noise_scale(0.8 ** (np.arange(N) % 4))
        return sim

    whole = start()
    whole.advance(48 * 30)
    first = start()
    first.advance(48 * 10)
    scale_at_dump = first.stepper.noise_scale()
    assert (scale_at_dump < 1.0).any()                                       # the damping state is not trivial
    print(f"resume test: {np.unique(scale_at_dump).size} distinct damping factors at the dump, "
          f"{(scale_at_dump < 1.0).sum()} of {N} members damped")
    c_first = first.stepper.counters()
    path = first.dump(tmp_path / "ckpt.h5")
    first.close()
    second = EnsembleSimulation.restore(path, cols, forcing)
    second.stepper.set_iteration_budget(budget)
    assert second.next_row == 1 + 480 and second.n_members == N and second.member_offset == 7000
    assert np.array_equal(second.stepper.noise_scale(), scale_at_dump)
    second.advance(48 * 20)
    assert np.array_equal(second.stepper.get_state(), whole.stepper.get_state())
    assert np.array_equal(second.moments(), whole.moments())
    assert np.array_equal(second.stepper.noise_scale(), whole.stepper.noise_scale())
    c_whole, c_second = whole.stepper.counters(), second.stepper.counters()
    for key in ("failed_attempts", "guard_trips", "jac_retry"):
        assert c_first[key] + c_second[key] == c_whole[key], key
    # a checkpoint is refused where it does not fit
    _, cols300, forcing300 = digest(300)
    with pytest.raises(ValueError, match="does not fit"):
        EnsembleSimulation.restore(path, cols300, forcing300)
    from hydromodel_amd._lib import HcError
    with pytest.raises(HcError):
        second.stepper.set_noise_scale(np.full(N, 1.5))                      # not a product of 0.8 factors
    whole.close()
    second.close()


def test_in_library_rccl_allreduce_on_one_gpu(gpu):
    """hc_allreduce_moments: the path's collective inside the library (RCCL bound with dlopen, no torch).  One GPU is
    what a round can reach: a one-handle "group" must hand the table back unchanged through ncclAllReduce, and the
    argument checks must hold (two handles on one device are refused)."""
    from hydromodel_amd._lib import HcError
    from hydromodel_amd.stepper import allreduce_handles
    _, cols, forcing = digest(200)
    ic = golden("g1_tables_200.npz")["initial_cond"]
    st = gpu.EnsembleStepper(cols, forcing, 64)
    st.set_state(ic)
    st.set_noise_philox(5, 0)
    st.step_rows(1, 48)
    before = st.moments()
    assert before[0, 1:49].tolist() == [64] * 48
    allreduce_handles([st])
    assert np.array_equal(st.moments(), before)
    other = gpu.EnsembleStepper(cols, forcing, 64)
    with pytest.raises(HcError, match="one handle per device"):
        allreduce_handles([st, other])
    other.close()
    st.close()


@pytest.mark.parametrize("dim_d", [541, 581, 640])
def test_split_column_kernel_against_the_one_wave_kernel_and_the_oracle(gpu, dim_d, monkeypatch):
    """Columns of 513..640 nodes run on TWO cooperating waves per member (DESIGN.md §5 "Split column").  Same inputs
    through the split-column kernel, through the one-wave kernel of the same depth (HYDROCOL_SPLIT_COLUMN=0) and through
    the oracle: daylight rows with evapo-transpiration, lateral flow across the cut region, host noise with a refresh
    row.  The two kernels sum norms and eliminate the tridiagonal system in different orders, so they agree to
    rounding on the first row (1e-7) and like any two implementations on the chained rows; solver statistics and
    water-table indices must be the oracle's."""
    from hydromodel_amd.digest import ColumnTables, ForcingDigest
    from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
    from oracle.oracle import Oracle
    params = default_parameters()
    cols = ColumnTables(params, synthetic_well(dim_d))
    forcing = ForcingDigest(params, synthetic_forcing_frame(1), cols)
    N, first, rows = 5, 28, 24                         # rows 28..51: daylight until row 35, night, the refresh row 48
    rng = np.random.default_rng(dim_d)
    y0 = np.tile(cols.z - 300.0, (N, 1)) + 0.2 * rng.standard_normal((N, cols.dim_d))
    y0[1] = cols.z - 2400.0                            # a member whose water table lies BELOW the cut (node 320 = 16 m)
    base = rng.standard_normal((N, cols.dim_d))
    nf = int(forcing.refresh[first:first + rows].sum())
    assert nf >= 1
    fresh = rng.standard_normal((nf, N, cols.dim_d))
    res = {}
    for mode in ("split", "one-wave"):
        monkeypatch.setenv("HYDROCOL_SPLIT_COLUMN", "0" if mode == "one-wave" else "1")     # (1: also where it is not the default)
        st = gpu.EnsembleStepper(cols, forcing, N)
        st.set_state(y0)
        st.set_noise_host(base)
        out = st.step_rows(first, rows, fresh_noise=fresh, want_wtd=True, want_stats=True, want_psi=True, want_diag=True)
        res[mode] = out
        st.close()
    monkeypatch.delenv("HYDROCOL_SPLIT_COLUMN")
    a, b = res["split"], res["one-wave"]
    assert a["kernel_ms"] > 0 and np.isfinite(a["psi"]).all()
    e = np.max(np.abs(a["psi"] - b["psi"]) / (1 + np.abs(b["psi"])), axis=2)
    assert e[0].max() < 1e-7, e[0]
    same_stats = (a["stats"] == b["stats"]).all(axis=2).mean()
    o = Oracle(cols, forcing.surface_evap)
    ok = total = 0
    worst_first = 0.0
    for k in range(N):
        r = o.run(forcing, y0[k], base[k], fresh[:, k, :], first, first + rows, want_psi=True, want_stats=True)
        want = r["psi_rows"][first:first + rows]
        eo = np.max(np.abs(a["psi"][:, k, :] - want) / (1 + np.abs(want)), axis=1)
        worst_first = max(worst_first, float(eo[0]))
        assert eo[0] < 1e-6 and eo.max() < 5e-2, (dim_d, k, eo)
        ok += int((a["stats"][:, k, :5] == r["per_row"][first:first + rows, :5]).all(axis=1).sum())
        ok_w = (a["wtd"][:, k] == r["wtd_est"][first:first + rows])
        assert ok_w.mean() >= 0.9, (dim_d, k)
        total += rows
    print(f"[D={dim_d}] split-column vs one-wave kernel: first row {e[0].max():.1e}, all rows {e.max():.1e}, "
          f"{same_stats:.0%} of the member-rows with identical statistics; vs the oracle: first row {worst_first:.1e}, "
          f"{ok}/{total} member-rows with the oracle's nfev/njev/nlu/steps/attempts")
    assert ok >= 0.8 * total
    # transpiration / lateral-flow diagnostics: the lateral-flow integral is a sum over BOTH halves
    assert a["diag"][0, :, 0].max() > 0.0 and a["diag"][:, :, 1].max() > 0.0
    assert np.allclose(a["diag"][0], b["diag"][0], rtol=1e-6, atol=1e-12)
    assert np.max(np.abs(a["diag"] - b["diag"])) < 5e-4      # later rows: the water table may cross a cell a row apart


def test_sweeps_deeper_than_576_nodes_run_on_the_split_column(gpu, monkeypatch):
    """Round 4: several parameter points in one handle at 577..640 nodes take the split-column kernel
    too -- every point brings its tables in the two-halves layout, the workgroup's chunk bookkeeping is drawn by the upper
    half of a pair and learnt by the lower half through the mailbox.  Properties: (i) each point of a three-point sweep is
    bit-equal -- states and moments -- to a stand-alone handle running that point with the same global member ids (also
    the split column); (ii) the sweep agrees with the same sweep on the one-wave kernels (HYDROCOL_SPLIT_COLUMN=0) within
    the chained-rows tier; (iii) per-point counts are complete.  Round 5: the split column runs on the TWO layout (four
    pairs per CU) and is the default from 513 nodes on, for sweeps as for single points -- D = 541 joins the depths."""
    from hydromodel_amd.digest import ColumnTables, ForcingDigest
    from hydromodel_amd.ensemble import SweepSimulation, check_sweep_points
    from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
    params = default_parameters()
    pts = [{"Soil_Properties": {"a0": 0.012}}, {"Soil_Properties": {"n": 2.0}}, {"Soil_Properties": {"n": 1.7, "psi_sat": -0.05}}]
    for depth in (541, 581, 640):
        cols_all = [ColumnTables(mp, synthetic_well(depth)) for mp in check_sweep_points(params, pts)]
        forcing = ForcingDigest(params, synthetic_forcing_frame(1), cols_all[0])
        psi0 = np.stack([c.z - 300.0 for c in cols_all])
        M, rows = 7, 30                                         # ragged: 7 members per point on pairs of waves
        big = SweepSimulation(cols_all, forcing, M, seed=2, psi0=psi0)
        big.advance(rows)
        y, mom = big.stepper.get_state(), big.moments()
        big.close()
        assert np.isfinite(y).all() and np.array_equal(mom[:, 0, 1:rows + 1], np.full((3, rows), M))
        for k in range(3):
            one = SweepSimulation([cols_all[k]], forcing, M, seed=2, first_point=k, psi0=psi0[k])
            one.advance(rows)
            assert np.array_equal(one.stepper.get_state(), y[k * M:(k + 1) * M]), (depth, k)
            assert np.array_equal(one.moments()[0], mom[k]), (depth, k)
            one.close()
        monkeypatch.setenv("HYDROCOL_SPLIT_COLUMN", "0")        # the same sweep on the one-wave kernels of 9 / 10 cells per lane
        flat = SweepSimulation(cols_all, forcing, M, seed=2, psi0=psi0)
        flat.advance(rows)
        y1 = flat.stepper.get_state()
        flat.close()
        monkeypatch.delenv("HYDROCOL_SPLIT_COLUMN")
        e = np.max(np.abs(y - y1) / (1.0 + np.abs(y1)))
        print(f"[D={depth}] three-point sweep, split column vs one-wave kernels after {rows} rows: {e:.1e}")
        assert e < 1e-3


def test_deep_columns_fall_back_to_one_wave_where_the_split_kernel_does_not_apply(gpu, monkeypatch):
    """A column whose root zone reaches into the lower half keeps the one-wave kernel (the split column's ET lives in the
    upper half); so does a sweep in which ONE point has such roots.  Both still run; the sweep agrees with stand-alone
    handles."""
    from hydromodel_amd.digest import ColumnTables, ForcingDigest
    from hydromodel_amd.ensemble import SweepSimulation, check_sweep_points
    from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
    params = default_parameters()
    pts = [{"Soil_Properties": {"a0": 0.012}}, {"Soil_Properties": {"n": 2.0}, "Trees": {"Max_Root_Depth_cm": 2000.0}}]
    cols_all = [ColumnTables(mp, synthetic_well(581)) for mp in check_sweep_points(params, pts)]
    assert cols_all[0].n_root_int <= 319 < cols_all[1].n_root_int
    forcing = ForcingDigest(params, synthetic_forcing_frame(1), cols_all[0])
    psi0 = np.stack([c.z - 300.0 for c in cols_all])
    big = SweepSimulation(cols_all, forcing, 6, seed=2, psi0=psi0)
    big.advance(10)
    y = big.stepper.get_state()
    big.close()
    monkeypatch.setenv("HYDROCOL_SPLIT_COLUMN", "0")            # a single point, one-wave kernel forced
    one = SweepSimulation([cols_all[0]], forcing, 6, seed=2, first_point=0, psi0=psi0[0])
    one.advance(10)
    assert np.array_equal(one.stepper.get_state(), y[:6])
    one.close()
    monkeypatch.delenv("HYDROCOL_SPLIT_COLUMN")
    deep_roots = default_parameters()
    deep_roots["Trees"]["Max_Root_Depth_cm"] = 2000.0           # 400 root-zone cells: beyond the upper half's 320
    cols = ColumnTables(deep_roots, synthetic_well(581))
    assert cols.n_root_int > 319
    forcing = ForcingDigest(deep_roots, synthetic_forcing_frame(1), cols)
    st = gpu.EnsembleStepper(cols, forcing, 4)
    st.set_state(cols.z - 300.0)
    st.set_noise_philox(3, 0)
    out = st.step_rows(20, 6, want_stats=True)
    assert np.isfinite(st.get_state()).all() and (out["stats"][:, :, 4] >= 1).all()
    st.close()


def test_per_member_spinup_on_the_split_column_kernel(gpu, monkeypatch):
    """`hc_spinup` (every member iterates to its own stop rule inside one launch, simulation.py:389-493) at a depth the
    split-column kernel serves: the stop rule needs the pair-wide water table and mean squared change; iteration counts
    must be those of the one-wave kernel and the states agree like two implementations of one contraction."""
    from hydromodel_amd.digest import ColumnTables, ForcingDigest
    from hydromodel_amd.ensemble import spinup_members_on_gpu
    from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
    params = default_parameters()
    cols = ColumnTables(params, synthetic_well(581))
    forcing = ForcingDigest(params, synthetic_forcing_frame(1), cols)
    res = {}
    for mode in ("split", "one-wave"):
        monkeypatch.setenv("HYDROCOL_SPLIT_COLUMN", "0" if mode == "one-wave" else "1")     # (1: also where it is not the default)
        st = gpu.EnsembleStepper(cols, forcing, 5)
        st.set_noise_philox(13, 100)
        res[mode] = spinup_members_on_gpu(st, cols, forcing)
        st.close()
    monkeypatch.delenv("HYDROCOL_SPLIT_COLUMN")
    (ya, ia), (yb, ib) = res["split"], res["one-wave"]
    assert (ia > 0).all() and (ib > 0).all()                      # every member met its stop rule
    assert np.max(np.abs(ia - ib)) <= 2, (ia, ib)
    assert np.max(np.abs(ya - yb)) < 0.05                         # cm; the spin-up contracts


def test_config3_at_full_size_properties(gpu):
    """BASELINE configs[2] at its size (262 144 members x D = 300, Philox noise): one simulated day in one launch.  Size-
    independent properties: every row counts every member, the index sums lie inside the grid, the ensemble has spread
    (members do differ) -- and any window of members is bit-equal to a small stand-alone handle given the same global
    ids, i.e. results do not depend on how many members share the launch or the grid."""
    _, cols, forcing = digest(300)
    ic = golden("g1_tables_300.npz")["initial_cond"]
    N, rows = 262144, 48
    st = gpu.EnsembleStepper(cols, forcing, N)
    st.set_state(ic)
    st.set_noise_philox(77, 0)
    out = st.step_rows(1, rows)
    assert out["launches"] == 1
    m = st.moments()
    assert np.array_equal(m[0, 1:1 + rows], np.full(rows, N))
    mean_idx = m[1, 1:1 + rows] / N
    var_idx = m[2, 1:1 + rows] / N - mean_idx ** 2
    assert (mean_idx > 40).all() and (mean_idx < 80).all() and (var_idx >= 0).all() and var_idx.max() > 0
    window = st.get_state(200000, 512)
    tail = st.get_state(N - 3, 3)
    c = st.counters()
    st.close()
    small = gpu.EnsembleStepper(cols, forcing, 512)
    small.set_state(ic)
    small.set_noise_philox(77, 200000)
    small.step_rows(1, rows)
    assert np.array_equal(small.get_state(), window)
    small.close()
    last = gpu.EnsembleStepper(cols, forcing, 3)
    last.set_state(ic)
    last.set_noise_philox(77, N - 3)
    last.step_rows(1, rows)
    assert np.array_equal(last.get_state(), tail)
    last.close()
    assert c["guard_trips"] == 0 and np.isfinite(window).all()


@pytest.mark.parametrize("well,fname", [(581, "g5s_deep_581.npz"), (401, "g5s_default_well_401.npz")])
def test_first_days_of_the_deepest_reference_well_replay_on_the_split_column_kernel(gpu, well, fname):
    """96 rows recorded inside the reference's own run at its deepest well (no. 14, max_depth 2 900 cm, D = 581;
    `make_golden.py deep`), each replayed from the reference's input state and noise vector through the split-column
    kernel: the two-wave layout against the reference itself, not only against the oracle.  The same for the well the
    reference's input_parameters.json selects (no. 10, D = 401: the 7-cells-per-lane kernel: one wave per SIMD until late round 5, the TWO layout since)."""
    _, cols, forcing = digest(well)
    g = golden(fname)
    st = gpu.EnsembleStepper(cols, forcing, 1)
    errs, same = [], 0
    for k, i in enumerate(g["rows"]):
        st.set_state(g["y0"][k][None, :])
        st.set_noise_host(g["nrnd_in"][k][None, :])
        fresh = g["nrnd_in"][k][None, None, :] if forcing.refresh[i] else np.zeros((0,))
        out = st.step_rows(int(i), 1, fresh_noise=fresh, want_stats=True)
        y1 = st.get_state()[0]
        ref = g["y1"][k]
        errs.append(np.max(np.abs(y1 - ref) / (1 + np.abs(ref))))
        same += out["stats"][0, 0, :5].tolist() == g["stats"][k].tolist()
    st.close()
    errs = np.array(errs)
    tiers = {"<1e-9": int((errs < 1e-9).sum()), "1e-9..1e-6": int(((errs >= 1e-9) & (errs < 1e-6)).sum()),
             ">=1e-6 (loose)": int((errs >= 1e-6).sum())}
    print(f"[D={well}] {len(errs)} reference rows on the {'split-column' if well > 512 else 'two-waves-per-SIMD' if well <= 448 else 'one-wave'} kernel: {same} with the reference's "
          f"nfev/njev/nlu/steps/attempts; tiers {tiers}")
    assert same >= 0.9 * len(errs), (same, len(errs))
    assert tiers[">=1e-6 (loose)"] < 0.2 * len(errs), tiers
    assert np.median(errs) < 1e-8 and errs.max() < 5e-2



def test_config2_at_full_size_whole_year(gpu):
    """BASELINE configs[1] at its size: 4 096 members x D = 200 through the WHOLE 1-year forcing (17 519 rows).  Properties
    that do not depend on the size: every solved row counts every member; the per-row moments do not depend on how the
    year is cut into launches (the library's choice for this ensemble -- 256 days per launch since round 5 -- against 5-day launches);
    the first members equal a small stand-alone handle; states stay finite and the water table stays on the grid."""
    _, cols, forcing = digest(200)
    ic = golden("g1_tables_200.npz")["initial_cond"]
    N, T = 4096, forcing.dim_t
    res = []
    for rows_per_launch in (0, 240):
        st = gpu.EnsembleStepper(cols, forcing, N)
        if rows_per_launch:
            st.set_rows_per_launch(rows_per_launch)
        st.set_state(ic)
        st.set_noise_philox(2, 0)
        out = st.step_rows(1, T - 1)
        res.append((st.moments(), st.get_state(0, 8), st.counters(), out["launches"]))
        st.close()
    (m, y, c, l0), (m2, y2, c2, l1) = res
    assert l0 < l1 and l1 == -(-(T - 1) // 240)
    assert np.array_equal(m, m2) and np.array_equal(y, y2) and c["failed_attempts"] == c2["failed_attempts"]
    assert np.array_equal(m[0, 1:], np.full(T - 1, N))
    mean_idx = m[1, 1:] / N
    assert np.isfinite(y).all() and mean_idx.min() > 30 and mean_idx.max() < 120
    small = gpu.EnsembleStepper(cols, forcing, 8)
    small.set_state(ic)
    small.set_noise_philox(2, 0)
    small.step_rows(1, T - 1)
    assert np.array_equal(small.get_state(), y)
    small.close()
    print(f"config 2 at full size: {N} members x D=200 x {T - 1} rows; failed attempts {c['failed_attempts']} "
          f"({c['failed_attempts'] / N:.2f} per member-year), budget trips {c['guard_trips']}; water-table index mean "
          f"{mean_idx.min():.1f}..{mean_idx.max():.1f}")


@pytest.mark.parametrize("mode", ["split", "one-wave"])
def test_rhs_at_the_deepest_reference_well_matches_the_reference(gpu, mode, monkeypatch):
    """G3 at the reference's deepest well (no. 14, D = 581): dy/dt of all 14 constructed states -- night / day, rain, capped
    infiltration, saturated top, lateral flow, spin-up flag, HYDRAULIC LIFT, ET off, LF off -- straight from
    `RichardsPDE.__call__`, through the RHS hook on the split-column path (two waves per member: edge states, water-table
    search, top flux and the cut's cell all cross the mailbox) and through the one-wave kernel of the same depth."""
    from helpers import rel_err
    monkeypatch.setenv("HYDROCOL_SPLIT_COLUMN", "0" if mode == "one-wave" else "1")
    _, cols, forcing = digest(581)
    g = golden("g34_states_581.npz")
    worst = 0.0
    for name in g["names"]:
        name = str(name)
        fl = g[f"{name}_flags"]
        hour = int(g[f"{name}_hour"])
        f8 = forcing_with_row(forcing, 8, float(g[f"{name}_precip"]), float(g[f"{name}_atm"]), 6 <= hour <= 17,
                              int(g["wtd_idx"]))
        st = gpu.EnsembleStepper(cols, f8, 2, flags={"ET": bool(fl[1]), "LF": bool(fl[2]), "HLIFT": bool(fl[3])})
        st.set_state(np.tile(g[f"{name}_y"], (2, 1)))
        st.set_noise_host(np.tile(g["n_rnd"], (2, 1)))
        dydt = st.rhs(8, spinup=bool(fl[0]))
        st.close()
        assert np.array_equal(dydt[0], dydt[1])
        e = rel_err(dydt[0], g[f"{name}_dydt"])          # relative to max(1, |ref|) element by element
        worst = max(worst, e)
        # hydraulic lift adds flux terms ~1e3 that cancel to ~1e-2 before the division by C ~ 1e-7
        assert e < (1e-4 if fl[3] else 1e-7), (mode, name, e)
    print(f"[{mode}] RHS of 14 constructed states at D = 581 vs the reference: worst {worst:.1e}")


@pytest.mark.parametrize("dim_d", [541, 581])
@pytest.mark.parametrize("build", ["special", "generic", "predict"])
def test_one_wave_kernels_of_the_deepest_columns_keep_their_guards(gpu, dim_d, build, monkeypatch):
    """Columns deeper than 512 nodes run on the split-column kernel by default, so the guards of the deep-column builds
    (launch partition with in-kernel noise; failure accounting with every attempt abandoned -- the two symptoms of round
    2's wrong builds, DESIGN.md §5 "Deep columns") no longer reach the one-wave kernels of 9 and 10 cells per lane there.
    Those kernels still serve sweeps at these depths, root zones below node 320 and HYDROCOL_SPLIT_COLUMN=0: the same two
    guards, on them."""
    from hydromodel_amd.digest import ColumnTables, ForcingDigest
    from hydromodel_amd.ensemble import pressure_head
    from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
    monkeypatch.setenv("HYDROCOL_SPLIT_COLUMN", "0")
    params = default_parameters()
    if build == "predict":
        params["Simulation_Flags"]["PREDICT"] = True
    cols = ColumnTables(params, synthetic_well(dim_d))
    forcing = ForcingDigest(params, synthetic_forcing_frame(1), cols)
    N, rows = 6, 50                                      # crosses the refresh row 48
    rng = np.random.default_rng(dim_d)
    y0 = np.tile(cols.z - 300.0, (N, 1)) + rng.standard_normal((N, cols.dim_d))
    res = []
    for step in (rows, 1, 7):
        st = gpu.EnsembleStepper(cols, forcing, N)
        if build == "generic":
            st.set_generic_exponents(True)
        st.set_state(y0)
        st.set_noise_philox(77, 3)
        wtd, stats = [], []
        done = 0
        while done < rows:
            n = min(step, rows - done)
            o = st.step_rows(1 + done, n, want_wtd=True, want_stats=True)
            wtd.append(o["wtd"]); stats.append(o["stats"])
            done += n
        res.append((st.get_state(), np.concatenate(wtd), np.concatenate(stats), st.moments()))
        st.close()
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert np.array_equal(a, b)
    assert np.isfinite(res[0][0]).all()
    # every attempt abandoned: five failures per row and member, the base vector x 0.8 fifteen times
    y1, _ = pressure_head(cols, cols.por_raw)
    base = np.random.default_rng(4).standard_normal((3, cols.dim_d))
    st = gpu.EnsembleStepper(cols, forcing, 3)
    if build == "generic":
        st.set_generic_exponents(True)
    st.set_iteration_budget(3)
    st.set_state(y1)
    st.set_noise_host(base)
    out = st.step_rows(1, 3, fresh_noise=np.zeros((0,)), want_stats=True)
    assert (out["stats"][:, :, 4] == 5).all() and (out["failed"] == 5).all()
    c = st.counters()
    assert c["failed_attempts"] == 45 and c["guard_trips"] == 45
    expect = base.copy()
    for _ in range(15):
        expect = expect * 0.8
    assert np.array_equal(st.get_noise_base(), expect)
    st.close()
