"""Round-2 GPU checks: device find_wtd known answers, failed-attempt accounting, exact diagnostics noise in
``Simulation.run``, the RCCL call path on one GPU, API edges added this round."""
import os

import numpy as np
import pytest

from helpers import WELLS, digest, forcing_frame, golden, rel_err

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


def test_device_find_wtd_known_answers_of_the_reference_suite(gpu):
    """code/tests/test_utilities.py:56-86 through the kernel's row epilogue (ballot + clz): all-False -> 9,
    all-True -> 0, bottom-4-True -> 6, mixed -> 7, on a 10-node column.  An iteration budget of 1 abandons every
    attempt before a step is accepted, so the row returns its start state and the epilogue sees exactly the pattern."""
    from hydromodel_amd.digest import ColumnTables, ForcingDigest
    from hydromodel_amd.synthetic import default_parameters
    g = golden("g2_pointwise_200.npz")
    params = default_parameters()
    well = {"soil": 0.0, "saprolite": 20.0, "weathered": 30.0, "max_depth": 45.0, "sat_depth": 10.0}
    cols = ColumnTables(params, well)
    assert cols.dim_d == 10
    frame = forcing_frame(1).copy()
    frame["WTD_m"] = -0.20
    forcing = ForcingDigest(params, frame, cols)
    cases = [c[:n] for c, n in zip(g["wtd_cases"], g["wtd_sizes"]) if n == 10]
    answers = [int(a) for a, n in zip(g["wtd_answers"], g["wtd_sizes"]) if n == 10]
    assert answers == [9, 0, 6, 7]
    Y = np.where(np.array(cases, dtype=bool), 1.0, -100.0)          # saturated <=> psi >= psi_sat
    st = gpu.EnsembleStepper(cols, forcing, len(Y))
    st.set_iteration_budget(1)
    st.set_state(Y)
    st.set_noise_host(np.zeros_like(Y))
    out = st.step_rows(2, 1, fresh_noise=np.zeros((0,)), want_wtd=True, want_stats=True)
    assert np.array_equal(st.get_state(), Y)
    assert out["wtd"][0].tolist() == answers
    assert (out["stats"][0, :, 4] == 5).all() and (out["failed"][0] == 5).all()
    st.close()


def test_failed_attempt_count_is_reported_per_row(gpu):
    """G5 rows the reference retried: the kernel's failed count times 0.8 reproduces the reference's damped vector."""
    _, cols, forcing = digest(1)
    g = golden("g5_traj_1.npz")
    stats, rows = g["per_row_stats"], g["rec_rows"]
    st = gpu.EnsembleStepper(cols, forcing, 1)
    seen = agree = 0
    for k, i in enumerate(rows):
        if stats[i, 4] <= 1:
            continue
        nin, nout = g["rec_nrnd_in"][k], g["rec_nrnd_out"][k]
        ref_failed = int(round(np.log(np.median(nout / nin)) / np.log(0.8)))
        st.set_state(g["rec_y0"][k][None, :])
        st.set_noise_host(nin[None, :])
        refresh = bool(forcing.refresh[i])
        out = st.step_rows(int(i), 1, fresh_noise=nin[None, None, :] if refresh else np.zeros((0,)), want_stats=True)
        failed, attempts = int(out["failed"][0, 0]), int(out["stats"][0, 0, 4])
        assert failed in (attempts - 1, attempts) and attempts <= 5 and int(out["stats"][0, 0, 5]) == int(refresh)
        if not refresh:
            expect = nin.copy()
            for _ in range(failed):
                expect *= 0.8
            assert np.array_equal(st.get_noise_base()[0], expect)
        seen += 1
        agree += failed == ref_failed
    st.close()
    print(f"retried reference rows: {seen}, same number of failed attempts on the GPU: {agree}")
    assert seen >= 10 and agree >= 0.6 * seen          # give-up rows are chaotic; most still agree


def test_simulation_diagnostics_use_the_noise_each_row_was_solved_with(gpu, tmp_path, monkeypatch):
    """ADVICE r1 (simulation.py:126): K_bkg / K_hrc of every saved row come from ``h_model(y_i, z, args_i)`` with
    ``n_rnd`` as that row's solve left it.  A low iteration budget makes some rows fail attempts; every row is then
    checked against the oracle's plugin call fed with the vector the ORACLE's own row loop ends the row with."""
    from hydromodel_amd.simulation import Simulation
    from hydromodel_amd.synthetic import default_parameters, write_site_information
    from oracle.oracle import Oracle, VIEW_NODES
    params = default_parameters()
    params["Site_Information"] = str(write_site_information(tmp_path / "site.json", {1: WELLS[1]}))
    params["Well_No"] = 1
    params["Simulation_Flags"]["ET"] = False            # short file: keep the ET demand out
    frame = forcing_frame(1).iloc[:48 * 3].reset_index(drop=True)
    ic = golden("g1_tables_1.npz")["initial_cond"]
    np.savetxt(tmp_path / "ic.csv", ic, fmt="%.17g")
    params["IC_Filename"] = str(tmp_path / "ic.csv")     # no spin-up: draw #0 is not consumed (simulation.py:358-385)
    # ordinary rows take 13-36 trips of the phase loop here (nfev + 5 njev); a budget of 24 abandons the costly ones
    monkeypatch.setenv("HYDROCOL_DEBUG_MAX_ITER", "24")
    sim = Simulation("diag", seed=12)
    sim.setupModel(params, frame)
    sim.run()
    monkeypatch.delenv("HYDROCOL_DEBUG_MAX_ITER")
    cols, forcing, out = sim.cols, sim.forcing, sim.output
    T, D = forcing.dim_t, cols.dim_d
    # replay the noise bookkeeping from the recorded states: which rows damped what
    rng = np.random.default_rng(np.random.SeedSequence(12))
    base = rng.standard_normal(D)                        # first draw: the base vector (simulation.py:561)
    o = Oracle(cols, forcing.surface_evap)
    st = gpu.EnsembleStepper(cols, forcing, 1)
    st.set_iteration_budget(24)
    damped_rows = 0
    for i in range(1, T):
        fresh = rng.standard_normal(D) if forcing.refresh[i] else None
        vec = fresh if fresh is not None else base
        # the product's own single-row solve tells how often row i failed (same kernel, same inputs)
        st.set_state(out["psi_press"][i - 1][None, :])
        st.set_noise_host(vec[None, :])
        r = st.step_rows(i, 1, fresh_noise=vec[None, None, :] if fresh is not None else np.zeros((0,)), want_stats=True)
        for _ in range(int(r["failed"][0, 0])):
            vec *= 0.8                                   # in place: the base array when the row does not refresh
        damped_rows += int(r["failed"][0, 0]) > 0
        assert np.array_equal(st.get_state()[0], out["psi_press"][i]), i
        q, K, C, kb, _ = o.model_eval(VIEW_NODES, out["psi_press"][i], vec)
        assert rel_err(out["K_bkg"][i], kb) < 1e-9, i
        assert rel_err(out["K_hrc"][i], K) < 1e-9, i
        assert rel_err(out["theta_vol"][i], q) < 1e-12, i
    st.close()
    print(f"rows with failed attempts: {damped_rows} of {T - 1}")
    assert damped_rows >= 3


def test_rccl_allreduce_of_the_moment_table_on_one_gpu(gpu):
    """The collective the driver's multi-GPU run makes, exercised on the one GPU available: a world-size-1 ``nccl``
    (= RCCL) process group, int64 all-reduce on device memory, result equal to the input."""
    import torch
    import torch.distributed as dist
    from hydromodel_amd.ensemble import allreduce_moments
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29871")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        assert dist.get_backend() == "nccl"
        _, cols, forcing = digest(200)
        st = gpu.EnsembleStepper(cols, forcing, 64)
        st.set_state(golden("g1_tables_200.npz")["initial_cond"])
        st.set_noise_philox(3, 0)
        st.step_rows(1, 48)
        m = st.moments()
        st.close()
        assert m[0, 1:49].tolist() == [64] * 48
        dev = torch.device("cuda", 0)
        assert allreduce_moments(m, dev) is m                        # world of one: no collective by default
        red = allreduce_moments(m, dev, force=True)                  # ... unless asked: RCCL runs
        assert red.dtype == np.int64 and np.array_equal(red, m)
        pts = np.stack([m, 2 * m])                                   # [P][3][T] tables of a sweep go the same way
        assert np.array_equal(allreduce_moments(pts, dev, force=True), pts)
        # the path bench.py takes: table exported device-to-device, reduced on the device, copied back once
        from hydromodel_amd.ensemble import allreduce_stepper_moments
        st = gpu.EnsembleStepper(cols, forcing, 64)
        st.set_state(golden("g1_tables_200.npz")["initial_cond"])
        st.set_noise_philox(3, 0)
        st.step_rows(1, 48)
        assert np.array_equal(allreduce_stepper_moments(st, dev, force=True), m)
        t = torch.zeros((3, st.T), dtype=torch.int64, device=dev)
        st.export_moments(t.data_ptr())
        assert np.array_equal(t.cpu().numpy(), m)
        st.close()
    finally:
        if created:
            dist.destroy_process_group()


def test_state_validation_and_moment_rules(gpu):
    from hydromodel_amd._lib import HcError
    _, cols, forcing = digest(200)
    ic = golden("g1_tables_200.npz")["initial_cond"]
    st = gpu.EnsembleStepper(cols, forcing, 3)
    bad = np.tile(ic, (3, 1))
    bad[2, 17] = np.nan                                  # beyond the first member: used to slip through
    with pytest.raises(HcError, match="not finite"):
        st.set_state(bad)
    st.set_state(ic)
    st.set_noise_philox(1, 0)
    import ctypes as C
    from hydromodel_amd import _lib as L
    a = L.StepArgs()
    a.row_begin, a.n_rows, a.spinup, a.accumulate_moments = 0, 1, 1, 1
    assert st.lib.hc_step_rows(st.h, C.byref(a)) < 0     # spin-up solves cannot feed the per-row moments
    assert b"accumulate_moments" in st.lib.hc_last_error()
    st.step_rows(0, 2, spinup=True)                      # the wrapper turns them off by itself
    assert st.moments().sum() == 0
    with pytest.raises(HcError):
        st.set_iteration_budget(0)
    st.close()


def test_water_table_mean_includes_the_depth_of_the_first_node():
    from hydromodel_amd.stepper import moments_to_mean_std
    m = np.array([[4, 4], [40, 44], [402, 486]], dtype=np.int64)
    mu0, sd0 = moments_to_mean_std(m, 5.0)
    mu1, sd1 = moments_to_mean_std(m, 5.0, z0=30.0)
    assert np.allclose(mu1, mu0 + 30.0) and np.array_equal(sd0, sd1)
    both = moments_to_mean_std(np.stack([m, m]), 5.0, 30.0)
    assert both[0].shape == (2, 2) and np.allclose(both[0][1], mu1)


@pytest.mark.parametrize("dim_d", [300, 361, 401, 461, 541, 581])
@pytest.mark.parametrize("build", ["special", "generic", "predict"])
def test_rows_in_one_launch_equal_rows_launched_one_by_one_with_philox_noise(gpu, dim_d, build):
    """Row-to-row carried state (psi, the base noise and its damping, the failure count) must not depend on where the
    launch boundaries fall, at every cells-per-lane count, for every build of the kernel (specialised / generic
    exponents / PREDICT) and with the in-kernel noise source.  Deep columns keep part of that state in a per-wave
    global region and rebuild the noise per attempt; a build whose register allocation went wrong at 10 cells per lane
    once passed every host-noise test and differed here from the second row on (DESIGN.md §5 "Deep columns")."""
    from hydromodel_amd.digest import ColumnTables, ForcingDigest
    from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
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


@pytest.mark.parametrize("dim_d", [401, 581])
@pytest.mark.parametrize("build", ["generic", "predict"])
def test_failure_accounting_of_the_other_builds_at_depth(gpu, dim_d, build):
    """test_retry_rule_on_every_noise_layout for the generic-exponent and PREDICT kernels of the deep columns: three rows
    in ONE launch, every attempt abandoned -> five failures per row and member, base noise x 0.8 fifteen times."""
    from hydromodel_amd.digest import ColumnTables, ForcingDigest
    from hydromodel_amd.ensemble import pressure_head
    from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
    params = default_parameters()
    if build == "predict":
        params["Simulation_Flags"]["PREDICT"] = True
    cols = ColumnTables(params, synthetic_well(dim_d))
    forcing = ForcingDigest(params, synthetic_forcing_frame(1), cols)
    y0, _ = pressure_head(cols, cols.por_raw)
    N = 3
    base = np.random.default_rng(4).standard_normal((N, cols.dim_d))
    st = gpu.EnsembleStepper(cols, forcing, N)
    if build == "generic":
        st.set_generic_exponents(True)
    st.set_iteration_budget(3)
    st.set_state(y0)
    st.set_noise_host(base)
    out = st.step_rows(1, 3, fresh_noise=np.zeros((0,)), want_stats=True)
    assert (out["stats"][:, :, 4] == 5).all() and (out["failed"] == 5).all()
    c = st.counters()
    assert c["failed_attempts"] == 3 * N * 5 and c["guard_trips"] == 3 * N * 5
    expect = base.copy()
    for _ in range(15):
        expect = expect * 0.8
    assert np.array_equal(st.get_noise_base(), expect)
    st.close()


def test_sweep_scheduler_with_ragged_sizes(gpu, monkeypatch):
    """Parameter points in one launch when nothing divides evenly: 3 points x 50 members, chunks of 7 (the last chunk of a
    point holds one member), more workgroups than chunks -- against each point in a handle of its own."""
    from hydromodel_amd.digest import ColumnTables, ForcingDigest
    from hydromodel_amd.ensemble import SweepSimulation, merge_parameters
    from hydromodel_amd.synthetic import default_parameters
    params = default_parameters()
    pts = [{"Soil_Properties": {"n": 2.0}}, {"Soil_Properties": {"n": 1.6, "a0": 0.02}},
           {"Soil_Properties": {"psi_sat": -0.3}, "Hydraulic_Conductivity": {"Lambda_Exponent": 1.2}}]
    cols_all = [ColumnTables(merge_parameters(params, p), WELLS[1]) for p in pts]
    forcing = ForcingDigest(params, forcing_frame(1), cols_all[0])
    ic = golden("g1_tables_1.npz")["initial_cond"]
    psi0 = np.stack([ic, ic + 3.0, ic - 5.0])
    monkeypatch.setenv("HYDROCOL_CHUNK_MEMBERS", "7")
    big = SweepSimulation(cols_all, forcing, 50, seed=9, psi0=psi0)
    big.advance(30, want_wtd=True)
    monkeypatch.delenv("HYDROCOL_CHUNK_MEMBERS")
    m_big, y_big = big.moments(), big.stepper.get_state()
    big.close()
    for j in range(3):
        one = SweepSimulation([cols_all[j]], forcing, 50, seed=9, first_point=j, psi0=psi0[j])
        one.advance(30)
        assert np.array_equal(one.stepper.get_state(), y_big[50 * j:50 * (j + 1)]), j
        assert np.array_equal(one.moments()[0], m_big[j]), j
        one.close()
    assert np.array_equal(m_big[:, 0, 1:31], np.full((3, 30), 50))


@pytest.mark.parametrize("por,root", [("Constant", "Uniform"), ("Linear", "Gamma_pdf"), ("Exponential", "Mixture")])
def test_other_profile_types_run_like_the_oracle(gpu, por, root):
    """the other porosity profiles and root densities only change the static tables; one day from the
    reference's own spin-up state of that configuration (g1q_*), RHS and 48 chained rows against the oracle."""
    from hydromodel_amd.digest import ColumnTables, ForcingDigest
    from hydromodel_amd.synthetic import default_parameters
    from oracle.oracle import Oracle
    params = default_parameters()
    params["Hydrological_Model"]["Porosity_Profile"] = por
    params["Trees"]["Root_Pdf_Profile"] = root
    cols = ColumnTables(params, WELLS[200])
    forcing = ForcingDigest(params, forcing_frame(1), cols)
    ic = golden(f"g1q_tables_{por.lower()}_{root.lower()}.npz")["initial_cond"]
    N, D, rows = 3, cols.dim_d, 48
    rng = np.random.default_rng(21)
    base = rng.standard_normal((N, D))
    o = Oracle(cols, forcing.surface_evap)
    st = gpu.EnsembleStepper(cols, forcing, N)
    st.set_state(ic)
    st.set_noise_host(base)
    for row in (2, 24):
        dydt = st.rhs(row)
        ref = o.rhs(Oracle.row(forcing.precip[row], forcing.atm[row], forcing.daylight[row], forcing.wtd_obs[row]),
                    ic, base[0])
        assert rel_err(dydt[0], ref) < 1e-7
    nf = st.n_refresh(1, rows)
    fresh = rng.standard_normal((nf, N, D))
    out = st.step_rows(1, rows, fresh_noise=fresh, want_wtd=True, want_psi=True)
    st.close()
    for k in range(N):
        r = o.run(forcing, ic, base[k], fresh[:, k, :], 1, 1 + rows, want_psi=True)
        want = r["psi_rows"][1:1 + rows]
        e = np.max(np.abs(out["psi"][:, k, :] - want) / (1 + np.abs(want)), axis=1)
        # chained rows: 1e-9 on the first, transients up to 5e-3 (measured 1.3e-3), back below 1e-3 at the end
        assert e[0] < 1e-9 and e.max() < 5e-3 and e[-1] < 1e-3, (por, k, e[0], e.max(), e[-1])
        assert (out["wtd"][:, k] == r["wtd_est"][1:1 + rows]).mean() >= 0.95
