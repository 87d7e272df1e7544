"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle and the golden fixtures.

Tolerances (fp64), also stated in DESIGN.md:
  * RHS pieces C, sink, flux: 1e-11 relative; dy/dt itself: 1e-7 * max(1, |ref|)  (dy/dt is a
    difference of nearly equal fluxes divided by C ~ 1e-7, so flux ulps are amplified)
  * one row, same inputs: identical solver statistics on regular rows and
    |d psi| <= 1e-6 * (1 + |psi|); stiff rows (> 100 RHS evaluations) 5e-2 * (1 + |psi|)
  * water-table index: equal
"""
import numpy as np
import pytest

from helpers import digest, golden, rel_err

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


def _oracle(cols, forcing, flags=None):
    from oracle.oracle import Oracle
    return Oracle(cols, forcing.surface_evap, flags=flags)


def _row(forcing, i, spinup=False):
    from oracle.oracle import Oracle
    return Oracle.row(forcing.precip[i], forcing.atm[i], forcing.daylight[i], forcing.wtd_obs[i], spinup=spinup)


STATE_NAMES = ["night_dry", "top_saturated", "lf_active", "dry_profile_day", "rough_day", "rough_night"]


def _states(well):
    g = golden(f"g34_states_{well}.npz")
    return np.array([g[f"{n}_y"] for n in STATE_NAMES]), g["n_rnd"]


# ------------------------------------------------------------------------------- RHS
@pytest.mark.parametrize("well", [1, 200, 300])
@pytest.mark.parametrize("flags", [None, {"ET": False}, {"LF": False}, {"HLIFT": True}])
def test_rhs_matches_oracle(gpu, well, flags):
    _, cols, forcing = digest(well)
    Y, n_rnd = _states(well)
    st = gpu.EnsembleStepper(cols, forcing, len(Y), flags=flags)
    st.set_state(Y)
    st.set_noise_host(np.tile(n_rnd, (len(Y), 1)))
    o = _oracle(cols, forcing, flags)
    rainy = int(np.argmax(forcing.precip > 0.0))
    for row, spin in ((2, False), (24, False), (rainy, False), (rainy + 24, False), (30, True)):
        dydt, aux = st.rhs(row, spinup=spin, want_aux=True)
        for k in range(len(Y)):
            ref, ra = o.rhs(_row(forcing, row, spin), Y[k], n_rnd, want_aux=True)
            assert rel_err(aux["c"][k], ra["c"], 1e-7) < 1e-11
            assert rel_err(aux["s"][k], ra["s"], 1e-3) < 1e-11
            assert rel_err(aux["f"][k], ra["f"]) < 1e-11
            assert abs(aux["pL"][k] - ra["pL"]) < 1e-12
            assert rel_err(dydt[k], ref) < 1e-7, (well, row, STATE_NAMES[k])
    st.close()


@pytest.mark.parametrize("well", [1, 200, 300])
def test_rhs_matches_reference_golden(gpu, well):
    """G3 straight from the reference (night/day/rain states share forcing-independent pieces)."""
    _, cols, forcing = digest(well)
    g = golden(f"g34_states_{well}.npz")
    # golden RHS were evaluated with explicit (hour, precip, atm); find forcing rows that reproduce them
    day_dry_row = int(np.argmax((forcing.daylight == 1) & (forcing.precip == 0.0)))
    night_dry_row = int(np.argmax((forcing.daylight == 0) & (forcing.precip == 0.0) & (np.arange(forcing.dim_t) > 0)))
    assert forcing.atm[day_dry_row] == float(g["day_dry_atm"])
    for name, row in (("night_dry", night_dry_row), ("day_dry", day_dry_row), ("lf_active", night_dry_row),
                      ("rough_night", night_dry_row), ("dry_profile_day", day_dry_row)):
        st = gpu.EnsembleStepper(cols, forcing, 1)
        st.set_state(g[f"{name}_y"][None, :])
        st.set_noise_host(g["n_rnd"][None, :])
        dydt, aux = st.rhs(row, want_aux=True)
        assert rel_err(dydt[0], g[f"{name}_dydt"]) < 1e-7, name
        assert rel_err(aux["c"][0][1:], g[f"{name}_mid_c"], 1e-7) < 1e-11
        assert rel_err(aux["f"][0][1:], g[f"{name}_mid_f"]) < 1e-11
        assert rel_err(aux["s"][0][1:], g[f"{name}_mid_s"], 1e-3) < 1e-11
        st.close()


@pytest.mark.parametrize("model,n,lam", [("vanGenuchten", 2.0, 1.0), ("vrettas_fung", 1.7, 1.3)])
def test_generic_exponent_path_matches_oracle(gpu, model, n, lam):
    """vanGenuchten plugin and non-default exponents go through pow(): same kernel, generic branch."""
    from hydromodel_amd.digest import ColumnTables, ForcingDigest
    from hydromodel_amd.synthetic import default_parameters
    from helpers import WELLS, forcing_frame
    params = default_parameters()
    params["Hydrological_Model"]["Name"] = model
    params["Soil_Properties"]["n"] = n
    params["Hydraulic_Conductivity"]["Lambda_Exponent"] = lam
    cols = ColumnTables(params, WELLS[200])
    forcing = ForcingDigest(params, forcing_frame(1), cols)
    Y, n_rnd = _states(200)
    st = gpu.EnsembleStepper(cols, forcing, len(Y))
    st.set_state(Y)
    st.set_noise_host(np.tile(n_rnd, (len(Y), 1)))
    o = _oracle(cols, forcing)
    for row in (2, 24):
        dydt = st.rhs(row)
        for k in range(len(Y)):
            assert rel_err(dydt[k], o.rhs(_row(forcing, row), Y[k], n_rnd)) < 1e-7
    out = st.step_rows(24, 1, fresh_noise=np.zeros((0,)), want_stats=True)
    y1 = st.get_state()
    for k in range(len(Y)):
        yo, so, _, _ = o.solve_row(_row(forcing, 24), 23, 24, Y[k], n_rnd)
        if so["nfev"] <= 100:           # stiff rows are chaotic in the FD-Jacobian's last bits
            assert np.max(np.abs(y1[k] - yo) / (1 + np.abs(yo))) < 1e-6
        else:
            assert np.isfinite(y1[k]).all()
    st.close()


# ------------------------------------------------------------------------------- plugin at the nodes
@pytest.mark.parametrize("well", [1, 200, 300])
@pytest.mark.parametrize("model,tag", [("vrettas_fung", "vf"), ("vanGenuchten", "vg")])
def test_model_nodes_matches_reference_golden(gpu, well, model, tag):
    _, cols, forcing = digest(well, model)
    g = golden(f"g2_pointwise_{well}.npz")
    names = ("sweep", "ic", "moist", "dry")
    Y = np.array([g[f"psi_{n}"] for n in names])
    st = gpu.EnsembleStepper(cols, forcing, len(Y))
    st.set_state(Y)
    st.set_noise_host(np.tile(g["n_rnd"], (len(Y), 1)))
    out = st.model_nodes()
    for k, name in enumerate(names):
        assert rel_err(out["theta"][k], g[f"{tag}_{name}_node_q"]) < 1e-13
        # K_bkg: log(v/m^2 + 1) is ill-conditioned as v -> 0 (SURVEY.md / DESIGN.md): 1e-9 near saturation
        assert rel_err(out["K_bkg"][k], g[f"{tag}_{name}_node_kbkg"]) < 1e-9
        assert rel_err(out["K"][k], g[f"{tag}_{name}_node_K"]) < 1e-9
        assert rel_err(out["C"][k], g[f"{tag}_{name}_node_C"], 1e-7) < 1e-11
        assert abs(out["q_inf_max"][k] - float(g[f"{tag}_{name}_node_qinf"])) < 1e-9
    st.close()


# ------------------------------------------------------------------------------- one row
@pytest.mark.parametrize("well", [1, 200, 300])
def test_single_row_matches_oracle(gpu, well):
    _, cols, forcing = digest(well)
    Y, n_rnd = _states(well)
    o = _oracle(cols, forcing)
    st = gpu.EnsembleStepper(cols, forcing, len(Y))
    same = total = 0
    tiers = {"<1e-9": 0, "<1e-6": 0, "stiff <5e-2": 0}
    for row in (2, 24, 49):
        st.set_state(Y)
        st.set_noise_host(np.tile(n_rnd, (len(Y), 1)))
        nf = st.n_refresh(row, 1)
        fresh = np.tile(n_rnd[::-1], (nf, len(Y), 1)) if nf else np.zeros((0,))
        out = st.step_rows(row, 1, fresh_noise=fresh, want_wtd=True, want_stats=True)
        y1 = st.get_state()
        for k in range(len(Y)):
            noise = n_rnd[::-1].copy() if nf else n_rnd
            yo, so, _, _ = o.solve_row(_row(forcing, row), row - 1, row, Y[k], noise)
            err = np.max(np.abs(y1[k] - yo) / (1 + np.abs(yo)))
            # stiff rows (constructed states, ~200 RHS evaluations, ~75 steps) decorrelate in the last
            # bits of the FD Jacobian: held to 50x the integrator's own tolerance scale
            tol = 1e-6 if so["nfev"] <= 100 else 5e-2
            assert err < tol, (well, row, STATE_NAMES[k], err)
            tiers["<1e-9" if err < 1e-9 else ("<1e-6" if err < 1e-6 else "stiff <5e-2")] += 1
            gs = out["stats"][0, k]
            total += 1
            same += [int(gs[0]), int(gs[1]), int(gs[2]), int(gs[3]), int(gs[4])] == \
                [so["nfev"], so["njev"], so["nlu"], so["nsteps"], so["attempts"]]
            if so["nfev"] <= 100:
                assert int(out["wtd"][0, k]) == o.find_wtd(yo >= cols.soil.psi_sat)
            assert int(gs[5]) == nf
    print(f"[well {well}] one-row solves vs the oracle: {same}/{total} with identical nfev/njev/nlu/steps/attempts; "
          f"error tiers {tiers}")
    assert same >= 0.8 * total, (same, total)
    # the loose tier is for the CONSTRUCTED stiff states (saturated top, lateral flow far from equilibrium: ~200 RHS
    # evaluations); on rows of the reference's own trajectory it stays below 20 % (next-but-one test)
    assert tiers["stiff <5e-2"] <= 0.5 * total, tiers
    st.close()


@pytest.mark.parametrize("well", [1, 200, 300])
def test_single_row_matches_reference_golden(gpu, well):
    """G4: reference solve outputs for constructed states (t_span = (7, 8))."""
    _, cols, forcing = digest(well)
    g = golden(f"g34_states_{well}.npz")
    night_dry_row = 8          # hour 4, no rain in the synthetic forcing's first day? checked below
    day_dry_row = int(np.argmax((forcing.daylight == 1) & (forcing.precip == 0.0)))
    for name in ("dry_profile_day",):
        st = gpu.EnsembleStepper(cols, forcing, 1)
        st.set_state(g[f"{name}_y"][None, :])
        st.set_noise_host(g["n_rnd"][None, :])
        # the golden solve used t_span (7, 8); a forcing row with the same args but another t differs only
        # through min_step, which never binds here
        out = st.step_rows(day_dry_row, 1, fresh_noise=np.zeros((0,)), want_stats=True)
        y1 = st.get_state()[0]
        ry = g[f"{name}_solve_y"]
        assert np.max(np.abs(y1 - ry) / (1 + np.abs(ry))) < 1e-6
        assert out["stats"][0, 0, :4].tolist() == g[f"{name}_solve_stats"][0, :4].tolist()
        st.close()
    assert night_dry_row > 0


@pytest.mark.parametrize("well", [1, 200])
def test_reference_trajectory_rows_replay(gpu, well):
    """G5 rows recorded inside the reference's own year-long run, replayed on the GPU."""
    _, cols, forcing = digest(well)
    g = golden(f"g5_traj_{well}.npz")
    stats = g["per_row_stats"]
    rows = g["rec_rows"]
    st = gpu.EnsembleStepper(cols, forcing, 1)
    errs, same, n_reg = [], 0, 0
    for k, i in enumerate(rows):
        if stats[i, 4] > 1 or i < 1:
            continue
        st.set_state(g["rec_y0"][k][None, :])
        refresh = bool(forcing.refresh[i])
        st.set_noise_host(g["rec_nrnd_in"][k][None, :])
        fresh = g["rec_nrnd_in"][k][None, None, :] if refresh else np.zeros((0,))
        out = st.step_rows(int(i), 1, fresh_noise=fresh, want_stats=True)
        y1 = st.get_state()[0]
        ref = g["rec_y1"][k]
        errs.append(np.max(np.abs(y1 - ref) / (1 + np.abs(ref))))
        n_reg += 1
        same += out["stats"][0, 0, :3].tolist() == stats[i, :3].tolist()
    errs = np.array(errs)
    assert n_reg > 250
    tiers = {"<1e-9": int((errs < 1e-9).sum()), "1e-9..1e-6": int(((errs >= 1e-9) & (errs < 1e-6)).sum()),
             "1e-6..1e-2 (loose)": int((errs >= 1e-6).sum())}
    print(f"[well {well}] {n_reg} rows of the reference's trajectory replayed: {same} with the reference's "
          f"nfev/njev/nlu; error tiers {tiers}")
    assert tiers["1e-6..1e-2 (loose)"] < 0.2 * n_reg, tiers
    assert same >= 0.95 * n_reg, (same, n_reg)
    assert np.median(errs) < 1e-8
    assert np.quantile(errs, 0.95) < 1e-5
    assert errs.max() < 1e-2
    st.close()


def test_retry_rule_damps_base_noise_in_place(gpu):
    """Rows the reference needed several attempts for: attempts and the x0.8 damping carry over."""
    _, cols, forcing = digest(1)
    g = golden("g5_traj_1.npz")
    stats = g["per_row_stats"]
    o = _oracle(cols, forcing)
    st = gpu.EnsembleStepper(cols, forcing, 1)
    checked = retried = agree = 0
    for k, i in enumerate(g["rec_rows"]):
        if stats[i, 4] <= 1 or forcing.refresh[i]:
            continue
        y0, nin = g["rec_y0"][k], g["rec_nrnd_in"][k]
        yo, so, n_out, _ = o.solve_row(_row(forcing, i), i - 1, i, y0, nin)
        st.set_state(y0[None, :])
        st.set_noise_host(nin[None, :])
        out = st.step_rows(int(i), 1, fresh_noise=np.zeros((0,)), want_stats=True)
        att = int(out["stats"][0, 0, 4])
        base_after = st.get_noise_base()[0]
        # every failed attempt multiplies the base vector by 0.8 in place; the last attempt may or may
        # not have succeeded (a row that exhausts its 5 attempts is damped 5 times)
        cands = []
        v = nin.copy()
        for k in range(att + 1):
            if k >= att - 1:
                cands.append(v.copy())
            v = v * 0.8
        assert any(np.array_equal(base_after, c) for c in cands), (i, att)
        checked += 1
        retried += att >= 2
        agree += att == so["attempts"]
    # Whether a razor-edge row fails at all is decided in the last bits (DESIGN.md "Parity tiers"): most of
    # the reference's failing rows must fail here too, not necessarily every one of them.
    assert checked >= 3 and retried >= 3 and retried >= 0.6 * checked, (checked, retried, agree)
    st.close()


# ------------------------------------------------------------------------------- many rows, many members
def test_two_days_six_members_match_oracle(gpu):
    _, cols, forcing = digest(200)
    g = golden("g5_traj_200.npz")
    D, N, rows = cols.dim_d, 6, 96
    rng = np.random.default_rng(33)
    base = rng.standard_normal((N, D))
    nf = int(forcing.refresh[1:1 + rows].sum())
    fresh = rng.standard_normal((nf, N, D))
    st = gpu.EnsembleStepper(cols, forcing, N)
    st.set_state(g["initial_cond"])
    st.set_noise_host(base)
    out = st.step_rows(1, rows, fresh_noise=fresh, want_wtd=True, want_stats=True, want_psi=True)
    o = _oracle(cols, forcing)
    mom = st.moments()
    for k in range(N):
        r = o.run(forcing, g["initial_cond"], base[k], fresh[:, k, :], 1, 1 + rows, want_psi=True, want_stats=True)
        assert np.array_equal(out["wtd"][:, k], r["wtd_est"][1:1 + rows])
        err = np.abs(out["psi"][:, k, :] - r["psi_rows"][1:1 + rows]) / (1 + np.abs(r["psi_rows"][1:1 + rows]))
        # chained rows: each row amplifies last-bit differences ~1e7x through the FD Jacobian + inexact Newton
        assert err[0].max() < 1e-9 and err.max() < 1e-3, (k, err.max())
        assert (out["stats"][:, k, 0] == r["per_row"][1:1 + rows, 0]).mean() > 0.9
    # ensemble moments = exact integer sums of the per-member indices
    w = out["wtd"].astype(np.int64)
    assert np.array_equal(mom[0, 1:1 + rows], np.full(rows, N))
    assert np.array_equal(mom[1, 1:1 + rows], w.sum(axis=1))
    assert np.array_equal(mom[2, 1:1 + rows], (w ** 2).sum(axis=1))
    assert mom[:, 1 + rows:].sum() == 0 and mom[:, 0].sum() == 0
    st.close()


def test_launch_partition_does_not_change_bits(gpu, monkeypatch):
    """48 rows in one launch == 4 launches of 12 rows (state, noise counters, moments)."""
    _, cols, forcing = digest(200)
    g = golden("g5_traj_200.npz")
    res = []
    for chunks in ((48,), (12, 12, 12, 12), (5, 43)):
        st = gpu.EnsembleStepper(cols, forcing, 16)
        st.set_state(g["initial_cond"])
        st.set_noise_philox(99, 0)
        row = 1
        for n in chunks:
            st.step_rows(row, n)
            row += n
        res.append((st.get_state(), st.moments()))
        st.close()
    for y, m in res[1:]:
        assert np.array_equal(y, res[0][0]) and np.array_equal(m, res[0][1])


def test_member_sharding_is_bit_exact(gpu):
    """8 members on one handle == 2 handles x 4 members with member_offset (the multi-GPU layout)."""
    _, cols, forcing = digest(200)
    g = golden("g5_traj_200.npz")
    whole = gpu.EnsembleStepper(cols, forcing, 8)
    whole.set_state(g["initial_cond"])
    whole.set_noise_philox(1234, 0)
    whole.step_rows(1, 60)
    parts = []
    for off in (0, 4):
        st = gpu.EnsembleStepper(cols, forcing, 4)
        st.set_state(g["initial_cond"])
        st.set_noise_philox(1234, off)
        st.step_rows(1, 60)
        parts.append((st.get_state(), st.moments()))
        st.close()
    assert np.array_equal(whole.get_state(), np.concatenate([p[0] for p in parts]))
    assert np.array_equal(whole.moments(), parts[0][1] + parts[1][1])
    whole.close()


def test_philox_stream(gpu):
    _, cols, forcing = digest(300)
    st = gpu.EnsembleStepper(cols, forcing, 1)
    st.set_noise_philox(7, 0)
    a = np.concatenate([st.philox_normals(m, d) for m in range(8) for d in range(8)])
    assert abs(a.mean()) < 0.03 and abs(a.std() - 1.0) < 0.03
    assert abs(np.mean(a ** 3)) < 0.1 and abs(np.mean(a ** 4) - 3.0) < 0.3
    assert np.array_equal(st.philox_normals(3, 5), st.philox_normals(3, 5))
    assert not np.array_equal(st.philox_normals(3, 5), st.philox_normals(3, 6))
    assert not np.array_equal(st.philox_normals(3, 5), st.philox_normals(4, 5))
    st.close()


def test_philox_run_matches_oracle_fed_with_the_same_normals(gpu):
    _, cols, forcing = digest(200)
    g = golden("g5_traj_200.npz")
    rows, N = 60, 3
    st = gpu.EnsembleStepper(cols, forcing, N)
    st.set_state(g["initial_cond"])
    st.set_noise_philox(555, 10)
    out = st.step_rows(1, rows, want_wtd=True, want_psi=True)
    o = _oracle(cols, forcing)
    draws = np.cumsum(forcing.refresh)[1:1 + rows][forcing.refresh[1:1 + rows] == 1]
    for k in range(N):
        base = st.philox_normals(10 + k, 0)
        fresh = np.array([st.philox_normals(10 + k, int(d)) for d in draws])
        r = o.run(forcing, g["initial_cond"], base, fresh, 1, 1 + rows, want_psi=True)
        assert np.array_equal(out["wtd"][:, k], r["wtd_est"][1:1 + rows])
        err = np.abs(out["psi"][:, k, :] - r["psi_rows"][1:1 + rows]) / (1 + np.abs(r["psi_rows"][1:1 + rows]))
        assert err[0].max() < 1e-9 and err.max() < 1e-3
    st.close()


def test_spinup_on_gpu_reproduces_reference_initial_condition(gpu):
    """Simulation.initial_conditions (simulation.py:389-493) against the reference's own record of it (g1s_spinup_*: solves
    used, the state after each of the first 12 solves, their solver statistics -- tests/golden/make_golden.py spinup):
    the SAME number of solves, the first ten states at the chained-rows tier against the reference and at 1e-9 against the
    oracle, the end state (~110 chained solves, chaotic last bits) within 0.02 cm."""
    from hydromodel_amd.ensemble import pressure_head, spinup_on_gpu
    for well in (1, 200):
        _, cols, forcing = digest(well)
        g, gs = golden(f"g1_tables_{well}.npz"), golden(f"g1s_spinup_{well}.npz")
        n_rnd = np.random.default_rng(np.random.SeedSequence(911)).standard_normal(cols.dim_d)
        ic, iters, early = spinup_on_gpu(cols, forcing, n_rnd)
        assert early and iters == int(gs["iterations"]), (well, iters, int(gs["iterations"]))   # 115 / 110 solves
        assert np.max(np.abs(ic - g["initial_cond"])) < 0.02
        # the first solves, one by one
        y0, _ = pressure_head(cols, cols.por_raw)
        assert np.array_equal(y0, gs["y_start"])                                     # host plugin, bit-exact (G2)
        st = gpu.EnsembleStepper(cols, forcing, 1)
        st.set_state(y0)
        st.set_noise_host(n_rnd[None, :])
        o = _oracle(cols, forcing)
        y_prev = y0.copy()
        e_orc, e_ref = [], []
        for j in range(10):
            out = st.step_rows(0, 1, spinup=True, moments=False, want_stats=True)
            y_g = st.get_state()[0]
            # the oracle on the SAME input (the kernel's previous state): one solve
            y_o, s_o, _, _ = o.solve_row(_row(forcing, 0, spinup=True), 0.0, 1.0, y_prev, n_rnd.copy())
            e_orc.append(rel_err(y_g, y_o))
            e_ref.append(rel_err(y_g, gs["y_first"][j]))
            # solver statistics: the oracle's on the same input at every solve; the reference's own for solve 0, the one
            # solve whose input is the reference's to the bit (after it the chains are ~1e-4 apart and a later solve may
            # take a few evaluations more or fewer)
            assert out["stats"][0, 0, :4].tolist() == [s_o[q] for q in ("nfev", "njev", "nlu", "nsteps")], (well, j)
            if j == 0:
                assert out["stats"][0, 0, :4].tolist() == gs["stats_first"][0].tolist(), well
            y_prev = y_g
        print(f"[well {well}] spin-up: {iters} solves (reference {int(gs['iterations'])}); first ten solves, relative difference "
              f"to the oracle on the same input {['%.1e' % e for e in e_orc]}, to the reference's chain {['%.1e' % e for e in e_ref]}")
        # The spin-up's first solves start far from equilibrium (hydrostatic profile, ~135 steps in solve 0): below the
        # water table C = epsilon divides flux differences and last-bit differences of the RHS show up at 1e-4 with identical
        # solver statistics -- the envelope DESIGN.md §3 documents for 0.2 % of random one-row cases (profiles/r04_fuzz.txt).
        # Against the reference's own chain the ten states are compared as chained stiff rows (5e-2 (1 + |psi|)):
        # measured 4e-4 / 7e-3 after solve 0, up to 6e-2 in solves 1-6 while the transient is steep, 2e-3 / 8e-3 by solve 9; the
        # chains meet again at the end state (0.02 cm above).
        # (measured, wells 1 / 200: solve 0 against the oracle 2e-4 / 6e-3, later solves <= 5e-4 / 8e-7)
        assert e_orc[0] < 2e-2 and max(e_orc[1:]) < 2e-3, (well, e_orc)
        assert e_ref[0] < 2e-2 and max(e_ref) < 1e-1, (well, e_ref)
        st.close()


def test_full_size_day_properties(gpu):
    """BASELINE config 3 shape on a reduced member count: determinism, ranges, moment counts."""
    _, cols, forcing = digest(300)
    g = golden("g1_tables_300.npz")
    N = 8192
    runs = []
    for _ in range(2):
        st = gpu.EnsembleStepper(cols, forcing, N)
        st.set_state(g["initial_cond"])
        st.set_noise_philox(42, 0)
        st.step_rows(1, 48)
        runs.append((st.get_state(), st.moments()))
        st.close()
    y, m = runs[0]
    assert np.array_equal(y, runs[1][0]) and np.array_equal(m, runs[1][1])        # idempotent / deterministic
    assert np.isfinite(y).all()
    assert np.array_equal(m[0, 1:49], np.full(48, N))
    mean_idx = m[1, 1:49] / N
    assert np.all(mean_idx >= 0) and np.all(mean_idx <= cols.dim_d - 1)
    assert np.all(m[2, 1:49] * N >= m[1, 1:49] ** 2)                              # variance >= 0
    assert np.std(y[:, 10]) > 0.0                                                 # members really differ


def test_baseline_config3_full_size_day(gpu):
    """BASELINE.json config 3 at its FULL size on one GPU (262 144 members x D=300, one simulated day), checked
    through size-independent properties: (i) any window of members equals a small stand-alone run of exactly those
    global member ids, bit for bit (the Philox stream is keyed by the global id, the kernel has no cross-member
    state); (ii) the int64 moments are exactly the sums of the per-member indices; (iii) a window checked
    against the CPU oracle fed the same normals."""
    _, cols, forcing = digest(300)
    g = golden("g1_tables_300.npz")
    N, D, rows = 262144, cols.dim_d, 48
    st = gpu.EnsembleStepper(cols, forcing, N)
    st.set_state(g["initial_cond"])
    st.set_noise_philox(7, 0)
    out = st.step_rows(1, rows, want_wtd=True)
    m = st.moments()
    w = out["wtd"].astype(np.int64)
    assert np.array_equal(m[0, 1:1 + rows], np.full(rows, N))
    assert np.array_equal(m[1, 1:1 + rows], w.sum(axis=1)) and np.array_equal(m[2, 1:1 + rows], (w ** 2).sum(axis=1))
    assert w.min() >= 0 and w.max() <= D - 1
    windows = {k: st.get_state(first=k, count=16) for k in (0, 131071, N - 16)}
    for k, y_big in windows.items():
        assert np.isfinite(y_big).all()
        small = gpu.EnsembleStepper(cols, forcing, 16)
        small.set_state(g["initial_cond"])
        small.set_noise_philox(7, k)
        o2 = small.step_rows(1, rows, want_wtd=True)
        assert np.array_equal(small.get_state(), y_big), k
        assert np.array_equal(o2["wtd"], out["wtd"][:, k:k + 16]), k
        small.close()
    # oracle on two members of the middle window, same normals (draw 0 = base, draw k = k-th refresh row)
    o = _oracle(cols, forcing)
    draws = np.cumsum(forcing.refresh)[1:1 + rows][forcing.refresh[1:1 + rows] == 1]
    for k in (131071, 131075):
        base = st.philox_normals(k, 0)
        fresh = np.array([st.philox_normals(k, int(d)) for d in draws])
        r = o.run(forcing, g["initial_cond"], base, fresh, 1, 1 + rows)
        assert (out["wtd"][:, k] == r["wtd_est"][1:1 + rows]).mean() > 0.95
        assert np.max(np.abs(windows[131071][k - 131071] - r["psi"]) / (1 + np.abs(r["psi"]))) < 1e-3
    st.close()


def test_north_star_one_million_members_in_one_handle(gpu):
    """BASELINE.json north_star: N = 1e6 at D = 300 on ONE GPU (1 048 576 members, 2.5 GB of state, one 48-row launch of
    ~3 s), checked through the size-independent properties of the full-size tests: every row counts every member; the
    int64 moments are finite sums in range; any 16-member window equals a stand-alone handle running exactly those global
    member ids, bit for bit (`bench.py`'s `n1e6` leg times the same shape)."""
    _, cols, forcing = digest(300)
    g = golden("g1_tables_300.npz")
    N, D, rows = 1048576, cols.dim_d, 48
    st = gpu.EnsembleStepper(cols, forcing, N)
    st.set_state(g["initial_cond"])
    st.set_noise_philox(7, 0)
    out = st.step_rows(1, rows)
    m = st.moments()
    assert out["launches"] == 1
    assert np.array_equal(m[0, 1:1 + rows], np.full(rows, N))
    assert (m[1, 1:1 + rows] >= 0).all() and (m[1, 1:1 + rows] <= N * (D - 1)).all()
    assert (m[2, 1:1 + rows] >= m[1, 1:1 + rows]).all()                          # sum idx^2 >= sum idx for integer idx >= 0
    for k in (0, 524287, N - 16):
        y_big = st.get_state(first=k, count=16)
        assert np.isfinite(y_big).all()
        small = gpu.EnsembleStepper(cols, forcing, 16)
        small.set_state(g["initial_cond"])
        small.set_noise_philox(7, k)
        small.step_rows(1, rows)
        assert np.array_equal(small.get_state(), y_big), k
        small.close()
    print(f"1 048 576 members x 48 rows x D = 300 in one launch: {out['kernel_ms']:.0f} ms = "
          f"{N / (out['kernel_ms'] * 1e-3):.0f} column-days/s")
    st.close()


# ------------------------------------------------------------------------------- errors
def test_error_paths(gpu):
    from hydromodel_amd._lib import HcError
    from hydromodel_amd.digest import ColumnTables, ForcingDigest
    from hydromodel_amd.synthetic import default_parameters, synthetic_well
    from helpers import forcing_frame
    _, cols, forcing = digest(200)
    st = gpu.EnsembleStepper(cols, forcing, 2)
    st.set_state(cols.z - 300.0)
    with pytest.raises(HcError, match="noise"):
        st.step_rows(1, 1)
    st.set_noise_philox(1, 0)
    with pytest.raises(HcError, match="outside"):
        st.step_rows(0, 1)
    with pytest.raises(HcError, match="outside"):
        st.step_rows(forcing.dim_t - 1, 5)
    with pytest.raises(HcError, match="finite"):
        st.set_state(np.full(cols.dim_d, np.nan))
    with pytest.raises(ValueError):
        st.set_state(np.zeros(7))
    st.set_noise_host(np.zeros((2, cols.dim_d)))
    with pytest.raises(HcError, match="fresh_noise"):
        st.step_rows(48, 1)                     # a refresh row in host-noise mode needs the vectors
    st.close()
    params = default_parameters()
    big = ColumnTables(params, synthetic_well(701))          # deeper than any reference well (max D = 581)
    fb = ForcingDigest(params, forcing_frame(1), big)
    with pytest.raises(HcError, match="dim_d"):
        gpu.EnsembleStepper(big, fb, 1)
    # PREDICT: the reference's TypeError (richards_pde.py:327-330) is kept by Simulation / the CLI unless the repair is
    # requested (tests/test_predict.py); the stepper itself runs the repaired mode
    st = gpu.EnsembleStepper(cols, forcing, 1, flags={"PREDICT": True})
    assert st.params.flag_predict == 1 and st.params.sat_cells == int(cols.sat_cells)
    st.close()


@pytest.mark.parametrize("dim_d", [361, 401, 461, 541, 581])
def test_deep_reference_wells_match_oracle(gpu, dim_d):
    """Wells 13, 10 (the reference's default), 5, 15, 14 of site_information.json: 6..10 cells per lane."""
    from hydromodel_amd.digest import ColumnTables, ForcingDigest
    from hydromodel_amd.ensemble import pressure_head
    from hydromodel_amd.synthetic import default_parameters, synthetic_well
    from helpers import forcing_frame
    params = default_parameters()
    well = synthetic_well(dim_d)
    well["sat_depth"] = 125.0 if dim_d == 401 else 100.0
    cols = ColumnTables(params, well)
    forcing = ForcingDigest(params, forcing_frame(1), cols)
    o = _oracle(cols, forcing)
    rng = np.random.default_rng(dim_d)
    D, N = cols.dim_d, 3
    Y = np.tile(cols.z - 300.0, (N, 1)) + 2.0 * rng.standard_normal((N, D))
    base = rng.standard_normal((N, D))
    st = gpu.EnsembleStepper(cols, forcing, N)
    st.set_state(Y)
    st.set_noise_host(base)
    for row in (2, 24):
        dydt, aux = st.rhs(row, want_aux=True)
        for k in range(N):
            ref, ra = o.rhs(_row(forcing, row), Y[k], base[k], want_aux=True)
            assert rel_err(aux["f"][k], ra["f"]) < 1e-11
            assert rel_err(aux["c"][k], ra["c"], 1e-7) < 1e-11
            assert rel_err(dydt[k], ref) < 1e-7
    rows = 30
    nf = st.n_refresh(1, rows)
    out = st.step_rows(1, rows, fresh_noise=np.zeros((nf, N, D)), want_wtd=True, want_stats=True)
    y1 = st.get_state()
    for k in range(N):
        r = o.run(forcing, Y[k], base[k], np.zeros((max(nf, 1), D)), 1, 1 + rows, want_stats=True)
        assert np.array_equal(out["wtd"][:, k], r["wtd_est"][1:1 + rows])
        err = np.max(np.abs(y1[k] - r["psi"]) / (1 + np.abs(r["psi"])))
        frac = (out["stats"][:, k, 0] == r["per_row"][1:1 + rows, 0]).mean()
        print(f"[D={dim_d}] member {k}: {rows} chained rows, final error {err:.1e}, rows with the oracle's nfev {frac:.0%}")
        assert err < 1e-3
        assert frac >= 0.7
    st.close()


def test_num_jac_retry_branch_matches_oracle(gpu, monkeypatch):
    """scipy's num_jac retries a column with a 10x step when f moved by < EPS**0.875 of its size.

    The model never gets there with the real threshold, so both sides raise it through their debug hooks
    (HYDROCOL_DEBUG_JAC_REJECT / ho_debug_set_jac_reject) and must still agree row for row.
    """
    from oracle.oracle import lib as oracle_lib
    _, cols, forcing = digest(200)
    Y, n_rnd = _states(200)
    keep = [0, 3]                                    # night_dry, dry_profile_day: regular rows
    Y = Y[keep]
    for reject in (1e-2, 0.3):
        monkeypatch.setenv("HYDROCOL_DEBUG_JAC_REJECT", repr(reject))
        oracle_lib().ho_debug_set_jac_reject(reject)
        try:
            st = gpu.EnsembleStepper(cols, forcing, len(Y))
            o = _oracle(cols, forcing)
            before = oracle_lib().ho_debug_jac_retry_count()
            for row in (2, 24):
                st.set_state(Y)
                st.set_noise_host(np.tile(n_rnd, (len(Y), 1)))
                out = st.step_rows(row, 1, fresh_noise=np.zeros((0,)), want_stats=True)
                y1 = st.get_state()
                for k in range(len(Y)):
                    yo, so, _, _ = o.solve_row(_row(forcing, row), row - 1, row, Y[k], n_rnd)
                    assert np.max(np.abs(y1[k] - yo) / (1 + np.abs(yo))) < 1e-6, (reject, row, k)
                    assert out["stats"][0, k, :4].tolist() == [so["nfev"], so["njev"], so["nlu"], so["nsteps"]]
            assert st.counters()["jac_retry"] > 0
            assert oracle_lib().ho_debug_jac_retry_count() > before
            st.close()
        finally:
            oracle_lib().ho_debug_set_jac_reject(0.0)
            monkeypatch.delenv("HYDROCOL_DEBUG_JAC_REJECT")


def test_default_runs_never_take_the_retry_branch(gpu):
    _, cols, forcing = digest(200)
    g = golden("g5_traj_200.npz")
    st = gpu.EnsembleStepper(cols, forcing, 64)
    st.set_state(g["initial_cond"])
    st.set_noise_philox(8, 0)
    st.step_rows(1, 96)
    c = st.counters()
    assert c["jac_retry"] == 0 and c["guard_trips"] == 0
    st.close()


def test_parameter_sweep_points_are_independent_runs(gpu, monkeypatch):
    """Config 5 in miniature: 4 (n, a0, psi_sat) points x 32 members stepped by ONE launch per batch of rows
    (hc_add_point) equal the same points run one handle at a time, bit for bit -- states, spin-ups and moments --
    whatever the scheduler's chunk size and however the points are dealt to ranks."""
    from hydromodel_amd.ensemble import SweepSimulation, parameter_sweep
    from hydromodel_amd.synthetic import default_parameters
    from helpers import WELLS, forcing_frame
    params = default_parameters()
    pts = [{"Soil_Properties": {"n": n, "a0": a0, "psi_sat": ps}}
           for n, a0, ps in ((2.0, 0.009, -0.0047), (1.5, 0.003, -0.001), (3.0, 0.03, -1.0), (2.2, 0.012, -0.05))]
    res = parameter_sweep(params, forcing_frame(1), WELLS[200], pts, n_members=32, n_rows=96, seed=5)
    assert sorted(res) == [0, 1, 2, 3]
    for k in res:
        m = res[k]["moments"]
        assert m.shape == (3, 17520)
        assert np.array_equal(m[0, 1:97], np.full(96, 32)) and np.isfinite(res[k]["psi0"]).all()
        assert res[k]["spinup_iterations"] > 0
    assert not np.array_equal(res[0]["psi0"], res[2]["psi0"])          # the equilibrium depends on the point
    assert not np.array_equal(res[0]["moments"], res[2]["moments"])
    # the same points, one handle each
    alone = parameter_sweep(params, forcing_frame(1), WELLS[200], pts, n_members=32, n_rows=96, seed=5,
                            one_launch=False)
    for k in range(4):
        assert np.array_equal(alone[k]["psi0"], res[k]["psi0"]), k
        assert np.array_equal(alone[k]["moments"], res[k]["moments"]), k
    # another chunking of the members
    monkeypatch.setenv("HYDROCOL_CHUNK_MEMBERS", "5")
    again = parameter_sweep(params, forcing_frame(1), WELLS[200], pts, n_members=32, n_rows=96, seed=5)
    monkeypatch.delenv("HYDROCOL_CHUNK_MEMBERS")
    for k in range(4):
        assert np.array_equal(again[k]["moments"], res[k]["moments"]), k
    # rank split: 2 "ranks" cover the grid exactly once, points dealt round-robin -- each rank's handle then holds
    # NON-consecutive points of the sweep and keys every point's Philox stream by its own member base
    r0 = parameter_sweep(params, forcing_frame(1), WELLS[200], pts, 32, 96, seed=5, rank=0, world=2)
    r1 = parameter_sweep(params, forcing_frame(1), WELLS[200], pts, 32, 96, seed=5, rank=1, world=2)
    assert sorted(r0) == [0, 2] and sorted(r1) == [1, 3]
    for part in (r0, r1):
        for k in part:
            assert np.array_equal(part[k]["moments"], res[k]["moments"]), k
            assert np.array_equal(part[k]["psi0"], res[k]["psi0"]), k
    # states, not only moments: the handle of all four points against point 2 alone
    from hydromodel_amd.digest import ColumnTables, ForcingDigest
    from hydromodel_amd.ensemble import merge_parameters
    cols_all = [ColumnTables(merge_parameters(params, p), WELLS[200]) for p in pts]
    forcing = ForcingDigest(params, forcing_frame(1), cols_all[0])
    psi0 = np.stack([res[k]["psi0"] for k in range(4)])
    big = SweepSimulation(cols_all, forcing, 32, seed=5, psi0=psi0)
    big.advance(48)
    one = SweepSimulation([cols_all[2]], forcing, 32, seed=5, first_point=2, psi0=psi0[2])
    one.advance(48)
    assert np.array_equal(big.stepper.get_state(64, 32), one.stepper.get_state())
    assert big.stepper.counters()["guard_trips"] == 0
    big.close()
    one.close()


@pytest.mark.parametrize("fname,well,model,n_rows", [("g5s_vangenuchten_200.npz", 200, "vanGenuchten", 480),
                                                     ("g5s_hlift_200.npz", 200, "vrettas_fung", 240),
                                                     ("g5s_noet_nolf_300.npz", 300, "vrettas_fung", 240)])
def test_plugin_and_flag_variants_replay_reference_rows(gpu, fname, well, model, n_rows):
    """Rows of reference runs with the vanGenuchten plugin / HLIFT on / ET+LF off, replayed on the GPU.

    All recorded rows ride in ONE launch per forcing row: rows that share a forcing row index are independent
    members, so each row is replayed as a 1-member launch (cheap: ~0.2 ms each)."""
    _, cols, forcing = digest(well, model)
    g = golden(fname)
    fl = g["flags"]
    flags = {"ET": bool(fl[1]), "LF": bool(fl[2]), "HLIFT": bool(fl[3])}
    st = gpu.EnsembleStepper(cols, forcing, 1, flags=flags)
    errs, same = [], 0
    step = 3 if fname.startswith("g5s_hlift") else 1        # HLIFT rows cost ~1000 RHS evaluations each
    rows = list(enumerate(g["rows"]))[::step]
    for k, i in rows:
        st.set_state(g["y0"][k][None, :])
        st.set_noise_host(g["nrnd_in"][k][None, :])
        fresh = g["nrnd_in"][k][None, None, :] if forcing.refresh[i] else np.zeros((0,))
        out = st.step_rows(int(i), 1, fresh_noise=fresh, want_stats=True)
        y1 = st.get_state()[0]
        ref = g["y1"][k]
        errs.append(np.max(np.abs(y1 - ref) / (1 + np.abs(ref))))
        same += out["stats"][0, 0, :5].tolist() == g["stats"][k].tolist()
    errs = np.array(errs)
    hlift = bool(fl[3])       # hydraulic-lift night rows are very stiff (~1000 RHS evaluations): chaotic in the last bits
    assert same >= (0.7 if hlift else 0.85) * len(errs), (same, len(errs))
    assert np.median(errs) < 1e-8
    assert np.quantile(errs, 0.7 if hlift else 0.9) < 1e-4
    assert errs.max() < (0.3 if hlift else 5e-2)
    st.close()


# ------------------------------------------------------------------------------- §8(f2) diagnostics
@pytest.mark.parametrize("well", [1, 200])
def test_row_diagnostics_match_oracle_and_reference(gpu, well):
    """transpiration / lateral_flow = pde_model.arg_out after the row's solve (simulation.py:629-630):
    the two integrals of the interior pde_fun call of the row's LAST RHS evaluation."""
    _, cols, forcing = digest(well)
    g = golden(f"g5_traj_{well}.npz")
    stats = g["per_row_stats"]
    o = _oracle(cols, forcing)
    st = gpu.EnsembleStepper(cols, forcing, 1)
    e_or, e_ref, n_day, n_lf = [], [], 0, 0
    for k, i in enumerate(g["rec_rows"]):
        if stats[i, 4] > 1 or i < 1:
            continue
        y0, nin = g["rec_y0"][k], g["rec_nrnd_in"][k]
        refresh = bool(forcing.refresh[i])
        st.set_state(y0[None, :])
        st.set_noise_host(nin[None, :])
        fresh = nin[None, None, :] if refresh else np.zeros((0,))
        out = st.step_rows(int(i), 1, fresh_noise=fresh, want_stats=True, want_diag=True)
        _, so, _, _ = o.solve_row(_row(forcing, i), i - 1, i, y0, nin.copy())
        d_or = o.last_arg_out()
        d = out["diag"][0, 0]
        if out["stats"][0, 0, 0] == so["nfev"]:
            e_or.append(np.max(np.abs(d - d_or) / (1e-6 + np.abs(d_or))))
        ref = np.array([g["transpiration"][i - 1], g["lateral_flow"][i - 1]])
        if out["stats"][0, 0, 0] == stats[i, 0]:
            e_ref.append(np.max(np.abs(d - ref) / (1e-6 + np.abs(ref))))
        n_day += d[0] > 0
        n_lf += d[1] > 0
    assert len(e_or) > 250 and len(e_ref) > 250
    assert n_day > 50 and n_lf > 20          # both branches exercised
    assert np.median(e_or) < 1e-10 and max(e_or) < 1e-5, (np.median(e_or), max(e_or))
    assert np.median(e_ref) < 1e-10 and max(e_ref) < 1e-5, (np.median(e_ref), max(e_ref))
    st.close()


def test_diagnostics_off_and_on_give_the_same_states(gpu):
    """Asking for the diagnostics must not change one bit of the trajectory."""
    _, cols, forcing = digest(200)
    g = golden("g5_traj_200.npz")
    res = []
    for want in (False, True):
        st = gpu.EnsembleStepper(cols, forcing, 5)
        st.set_state(g["initial_cond"])
        st.set_noise_philox(11)
        out = st.step_rows(1, 48, want_wtd=True, want_diag=want)
        res.append((st.get_state(), out["wtd"], out.get("diag")))
        st.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    d = res[1][2]
    assert d.shape == (48, 5, 2) and (d >= 0).all() and d[:, :, 0].max() > 0
    night = ~forcing.daylight[1:49].astype(bool)
    assert (d[night, :, 0] == 0).all()      # ET acts in daylight only (richards_pde.py:258)


# ------------------------------------------------------------------------------- §8(f1) per-member spin-up
def test_per_member_spinup_in_one_launch(gpu):
    """hc_spinup: every member iterates with its own noise vector until its own stop rule (simulation.py:468)
    holds, all inside one launch.  Member 0 carries the reference's spin-up draw -> the reference's IC."""
    from hydromodel_amd.ensemble import pressure_head, spinup_members_on_gpu, spinup_on_gpu
    _, cols, forcing = digest(200)
    g = golden("g1_tables_200.npz")
    N, D = 6, cols.dim_d
    noise = np.random.default_rng(5).standard_normal((N, D))
    noise[0] = np.random.default_rng(np.random.SeedSequence(911)).standard_normal(D)
    st = gpu.EnsembleStepper(cols, forcing, N)
    st.set_noise_host(noise)
    psi0, iters = spinup_members_on_gpu(st, cols, forcing)
    assert psi0.shape == (N, D) and (iters > 0).all(), iters
    assert np.max(np.abs(psi0[0] - g["initial_cond"])) < 0.02          # ~115 chained solves, chaotic last bits
    assert len(set(iters.tolist())) > 1 or not np.allclose(psi0[1], psi0[2])    # members really differ
    # same answer as the host-driven loop (one launch per solve, stop rule in NumPy) for two members
    for k in (0, 3):
        ic, it, early = spinup_on_gpu(cols, forcing, noise[k])
        assert early and abs(it - iters[k]) <= 2, (k, it, iters[k])
        assert np.max(np.abs(ic - psi0[k])) < 0.05
    # and as the CPU oracle's spin-up
    o = _oracle(cols, forcing)
    y0, _ = pressure_head(cols, cols.por_raw)
    for k in (1, 4):
        ic, it = o.spinup(_row(forcing, 0), forcing.zwtd_cm[0], y0, noise[k])
        assert abs(it - iters[k]) <= 2, (k, it, iters[k])
        assert np.max(np.abs(ic - psi0[k])) < 0.05
    # the cap: members that cannot finish in 3 solves report -3 and keep their 3-solve state
    st.set_state(y0)
    st.set_noise_host(noise)
    it3, _ = st.spinup(forcing.zwtd_cm[0], cols.z[0], max_iterations=3)
    assert (it3 == -3).all()
    st.close()


def test_numpy_seeded_ensemble_member0_is_the_single_column_run(gpu):
    """Config C2 seeding: member 0 consumes default_rng(SeedSequence(seed)) exactly like the
    reference's single-column run, member k the stream spawn_key=(k,); every member equals the oracle fed the
    same vectors."""
    from hydromodel_amd.ensemble import EnsembleSimulation, member_generators
    _, cols, forcing = digest(1)
    g = golden("g5_traj_1.npz")
    N, D, rows = 5, cols.dim_d, 96
    sim = EnsembleSimulation(cols, forcing, N, seed=911, noise="numpy")      # shared spin-up with member 0's draw #0
    assert np.max(np.abs(sim.psi0 - g["initial_cond"])) < 0.02
    out = sim.advance(rows, want_wtd=True, want_psi=True)
    # member 0 = the reference's stream (spin-up, base, refresh...): the reference's own first two days
    ref_idx = np.rint(g["wtd_est_cm"][1:1 + rows] / cols.dz).astype(int)
    assert (out["wtd"][:, 0] == ref_idx).mean() > 0.97
    assert np.max(np.abs(out["psi"][0, 0] - g["rec_y1"][0])) < 0.02
    o = _oracle(cols, forcing)
    n_fresh = int(forcing.refresh[1:1 + rows].sum())
    for k, gen in enumerate(member_generators(911, N)):
        gen.standard_normal(D)                              # draw #0: spin-up (simulation.py:426)
        base = gen.standard_normal(D)                       # draw #1: base vector (:561)
        fresh = np.stack([gen.standard_normal(D) for _ in range(n_fresh)])
        r = o.run(forcing, sim.psi0, base, fresh, 1, 1 + rows, want_psi=True)
        assert (out["wtd"][:, k] == r["wtd_est"][1:1 + rows]).mean() > 0.97, k
        assert np.max(np.abs(out["psi"][0, k] - r["psi_rows"][1])) < 1e-8
    assert not np.array_equal(out["wtd"][:, 1], out["wtd"][:, 2]) or not np.array_equal(out["psi"][-1, 1], out["psi"][-1, 2])
    sim.close()


def test_philox_spinup_modes(gpu):
    from hydromodel_amd.ensemble import PHILOX_DRAW_SPINUP, EnsembleSimulation
    _, cols, forcing = digest(200)
    shared = EnsembleSimulation(cols, forcing, 4, seed=3)
    per = EnsembleSimulation(cols, forcing, 4, seed=3, spinup="member")
    assert per.psi0.shape == (4, cols.dim_d) and (per.spinup_iters > 0).all()
    # global member 0 spins up with the same vector in both modes
    assert np.max(np.abs(per.psi0[0] - shared.psi0)) < 0.05
    assert not np.allclose(per.psi0[1], per.psi0[2])
    # the spin-up vector is its own Philox draw, distinct from the base vector
    v_spin = per.stepper.philox_normals(0, PHILOX_DRAW_SPINUP)
    assert not np.array_equal(v_spin, per.stepper.philox_normals(0, 0))
    a = per.advance(48, want_wtd=True)
    assert a["wtd"].shape == (48, 4)
    shared.close(); per.close()


# ------------------------------------------------------------------------------- edge cases
@pytest.mark.parametrize("dim_d", [41, 64, 65, 128, 129, 193, 257, 320, 321])
def test_depth_counts_around_the_lane_boundaries(gpu, dim_d):
    """D below one wavefront, at exact multiples of 64 (no padding lane) and one node past them (a single
    node in a new cell slot): RHS and one solved row against the oracle, three members."""
    from hydromodel_amd.digest import ColumnTables, ForcingDigest
    from hydromodel_amd.ensemble import pressure_head
    from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
    params = default_parameters()
    cols = ColumnTables(params, synthetic_well(dim_d))
    # the observed water table must lie on the grid (else the reference skips the row): 1.5 m for the shallow columns
    forcing = ForcingDigest(params, synthetic_forcing_frame(1, wtd_m=-1.5 if dim_d <= 65 else -3.0), cols)
    assert cols.dim_d == dim_d and (forcing.wtd_obs >= 0).all()
    y0, _ = pressure_head(cols, cols.por_raw)
    N = 3
    rng = np.random.default_rng(dim_d)
    noise = rng.standard_normal((N, dim_d))
    # smooth perturbations of the hydrostatic-like profile: regular solves, so that the tight tolerance tier applies
    psi = y0[None, :] + np.array([0.8 * np.sin(cols.z / 60.0 + k) for k in range(N)])
    o = _oracle(cols, forcing)
    st = gpu.EnsembleStepper(cols, forcing, N)
    for row in (3, 25):                                   # night, daylight
        st.set_state(psi)
        st.set_noise_host(noise)
        f = st.rhs(row)
        for k in range(N):
            fo = o.rhs(_row(forcing, row), psi[k], noise[k])
            assert rel_err(f[k], fo) < 1e-7, (row, k)
        # Solved rows start from a settled state (80 spin-up solves in one launch).  The first solves from the raw
        # start profile take 300-400 RHS evaluations and are ill-conditioned at rtol = 1e-3: SciPy's own BDF and
        # the C oracle, driven by the same RHS, already differ by 3e-2 there -- not a row to compare bits on.
        st.set_state(y0)
        st.set_noise_host(noise)
        st.spinup(forcing.zwtd_cm[0], cols.z[0], max_iterations=80)
        y_eq = st.get_state()
        st.set_noise_host(noise)                       # spin-up retries may have damped the vectors in place
        out = st.step_rows(row, 1, fresh_noise=np.zeros((0,)), want_stats=True, want_wtd=True)
        y1 = st.get_state()
        for k in range(N):
            yo, so, _, _ = o.solve_row(_row(forcing, row), row - 1, row, y_eq[k], noise[k].copy())
            tol = 1e-6 if so["nfev"] == out["stats"][0, k, 0] and so["nfev"] <= 100 else 5e-2
            assert np.max(np.abs(y1[k] - yo) / (1 + np.abs(yo))) < tol, (row, k)
            assert out["wtd"][0, k] == Oracle_find_wtd(yo, cols.soil.psi_sat)
    st.close()


def Oracle_find_wtd(y, psi_sat):
    from oracle.oracle import Oracle
    return Oracle.find_wtd(y >= psi_sat)


@pytest.mark.parametrize("n_members", [1, 3, 5, 7, 1023, 1025])
def test_member_counts_that_do_not_fill_a_workgroup(gpu, n_members):
    """4 waves share a workgroup and the grid is persistent: any member count gives each member the result it
    has in a run of its own (same global id)."""
    _, cols, forcing = digest(200)
    g = golden("g5_traj_200.npz")
    st = gpu.EnsembleStepper(cols, forcing, n_members)
    st.set_state(g["initial_cond"])
    st.set_noise_philox(9, 0)
    out = st.step_rows(1, 4, want_wtd=True)
    y = st.get_state()
    m = st.moments()
    assert np.array_equal(m[0, 1:5], np.full(4, n_members)) and np.isfinite(y).all()
    for k in sorted({0, n_members // 2, n_members - 1}):
        one = gpu.EnsembleStepper(cols, forcing, 1)
        one.set_state(g["initial_cond"])
        one.set_noise_philox(9, k)
        o1 = one.step_rows(1, 4, want_wtd=True)
        assert np.array_equal(one.get_state()[0], y[k]) and np.array_equal(o1["wtd"][:, 0], out["wtd"][:, k])
        one.close()
    st.close()


def test_empty_launch_and_skipped_rows(gpu):
    """n_rows = 0 is a no-op; rows whose observation is off the grid are skipped as in simulation.py:582-588:
    the state is carried over unchanged, no noise is consumed, the moments do not count them."""
    import copy
    _, cols, forcing = digest(200)
    g = golden("g5_traj_200.npz")
    N, D = 4, cols.dim_d
    fc = copy.copy(forcing)
    fc.wtd_obs = forcing.wtd_obs.copy()
    skipped = [2, 3, 48, 50]                              # 48 is a refresh row
    fc.wtd_obs[skipped] = -1
    rng = np.random.default_rng(1)
    base = rng.standard_normal((N, D))
    st = gpu.EnsembleStepper(cols, fc, N)
    st.set_state(g["initial_cond"])
    st.set_noise_host(base)
    out0 = st.step_rows(1, 0)
    assert out0["launches"] == 0 and np.array_equal(st.get_state(), np.tile(g["initial_cond"], (N, 1)))
    rows = 60
    fc.refresh = forcing.refresh.copy()
    fc.refresh[fc.wtd_obs < 0] = 0                        # what ForcingDigest does: a skipped row draws nothing
    n_fresh = st.n_refresh(1, rows)
    assert n_fresh == int(forcing.refresh[1:1 + rows].sum()) - 1
    fresh = rng.standard_normal((n_fresh, N, D))
    out = st.step_rows(1, rows, fresh_noise=fresh, want_wtd=True, want_psi=True, want_stats=True)
    m = st.moments()
    for i in skipped:
        assert np.array_equal(out["psi"][i - 1], np.zeros((N, D)))          # psi[i] stays as initialised (zeros)
        assert (out["stats"][i - 1, :, 0] == 0).all() and m[0, i] == 0
    live = [i for i in range(1, 1 + rows) if i not in skipped]
    assert (m[0, live] == N).all()
    # the state really is carried across the gap: row 4 starts from row 1's result
    o = _oracle(cols, fc)
    for k in range(N):
        r = o.run(fc, g["initial_cond"], base[k], fresh[:, k, :], 1, 1 + rows, want_psi=True)
        assert (out["wtd"][[i - 1 for i in live], k] == r["wtd_est"][live]).mean() > 0.95
        err = np.abs(out["psi"][3, k] - r["psi_rows"][4]) / (1 + np.abs(r["psi_rows"][4]))
        assert err.max() < 1e-3, (k, err.max())       # chained solves: same tier as test_two_days_six_members_match_oracle
    st.close()


def test_ensemble_mean_sigma_match_the_oracle_over_a_month(gpu):
    """ensemble mu / sigma of the water table vs the CPU restatement on identical seeds (target
    |d mu|, |d sigma| <= 0.05 dz), here 48 members x 30 days: 1 440 chained rows per member, long past the point
    where individual members agree bit for bit."""
    from concurrent.futures import ThreadPoolExecutor
    from hydromodel_amd.ensemble import EnsembleSimulation, member_generators
    from hydromodel_amd.stepper import moments_to_mean_std
    _, cols, forcing = digest(200)
    g = golden("g5_traj_200.npz")
    N, D, rows = 48, cols.dim_d, 30 * 48
    sim = EnsembleSimulation(cols, forcing, N, seed=77, noise="numpy", psi0=g["initial_cond"])
    done = 0
    while done < rows:
        sim.advance(240)
        done += 240
    mean_g, std_g = sim.wtd_mean_std()
    sim.close()
    n_fresh = int(forcing.refresh[1:1 + rows].sum())

    def member(k):
        gen = member_generators(77, 1, k)[0]
        base = gen.standard_normal(D)                      # psi0 given: no spin-up draw
        fresh = np.stack([gen.standard_normal(D) for _ in range(n_fresh)])
        return _oracle(cols, forcing).run(forcing, g["initial_cond"], base, fresh, 1, 1 + rows)["wtd_est"][1:1 + rows]

    with ThreadPoolExecutor(max_workers=8) as ex:
        w = np.array(list(ex.map(member, range(N))), dtype=np.int64)          # [N][rows]
    m = np.zeros((3, forcing.dim_t), dtype=np.int64)
    m[0, 1:1 + rows], m[1, 1:1 + rows], m[2, 1:1 + rows] = N, w.sum(axis=0), (w ** 2).sum(axis=0)
    mean_o, std_o = moments_to_mean_std(m, cols.dz)
    dmu = np.abs(mean_g[1:1 + rows] - mean_o[1:1 + rows])
    dsg = np.abs(std_g[1:1 + rows] - std_o[1:1 + rows])
    # The observable is a thresholded integer (the cell index of the water table) and the members move together
    # with the forcing: when the table crosses a cell boundary, members whose chaotic last bits differ cross it one
    # half-hour row earlier or later, which shows up as a transient of a fraction of a cell in mu for a few rows
    # (measured: 71 of 1 440 rows above 0.05 dz, max 0.29 dz).  Averaged over the month the statistics agree to
    # 0.008 dz; that, a bound on the transients and the share of quiet rows are what is asserted.
    assert dmu.mean() <= 0.05 * cols.dz and dsg.mean() <= 0.05 * cols.dz, (dmu.mean(), dsg.mean())
    assert dmu.max() <= 0.5 * cols.dz and dsg.max() <= 0.5 * cols.dz, (dmu.max(), dsg.max())
    assert (dmu <= 0.05 * cols.dz).mean() >= 0.90 and (dsg <= 0.05 * cols.dz).mean() >= 0.85
    assert dmu[:96].max() == 0.0 and dsg[:96].max() == 0.0       # the first two days: identical integer moments


def test_attempt_that_chatters_on_a_discontinuity_is_abandoned_and_retried(gpu):
    """A row found in the 262 144-member run (member 165 062, row 2 140 of the 1-year forcing, tools/dev/guard_hunt.py):
    from this state the BDF step controller slides along a discontinuity of the RHS -- Newton only converges for
    h ~ 1e-11, the controller cycles halve / accept twice / x10 and time advances ~1e-11 per cycle.  SciPy's
    algorithm has no exit there; the CPU oracle does the same on most 1e-13 perturbations of this input.  Kernel and
    oracle give such an attempt a work budget, then apply the reference's failure rule (noise x0.8, retry)."""
    from hydromodel_amd.digest import ColumnTables, ForcingDigest
    from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
    params = default_parameters()
    cols = ColumnTables(params, synthetic_well(300))
    forcing = ForcingDigest(params, synthetic_forcing_frame(1), cols)
    g = golden("chatter_row_300.npz")
    row = int(g["row"])
    st = gpu.EnsembleStepper(cols, forcing, 1)
    st.set_state(g["y_before"])
    st.set_noise_host(g["z"][None, :])
    out = st.step_rows(row, 1, fresh_noise=np.zeros((0,)), want_stats=True)
    c = st.counters()
    nfev, attempts = int(out["stats"][0, 0, 0]), int(out["stats"][0, 0, 4])
    y = st.get_state()[0]
    base_after = st.get_noise_base()[0]
    assert np.isfinite(y).all()
    assert c["guard_trips"] >= 1 and attempts == c["guard_trips"] + 1 and c["guard_last_row"] == row
    assert 20000 * c["guard_trips"] < nfev + 5 * int(out["stats"][0, 0, 1]) + 10 < 20000 * (c["guard_trips"] + 1)
    assert out["kernel_ms"] < 3000.0                                   # bounded: ~0.1 s per abandoned attempt
    assert np.allclose(base_after, g["z"] * 0.8 ** c["guard_trips"], rtol=1e-15)    # the x0.8 rule, in place
    # the oracle on the same input happens to get through in one attempt; both answers are valid rtol = 1e-3 solves
    o = _oracle(cols, forcing)
    yo, so, _, _ = o.solve_row(_row(forcing, row), row - 1, row, g["y_before"], g["z"].copy())
    assert so["success"] == 1
    assert np.max(np.abs(y - yo) / (1 + np.abs(yo))) < 5e-2
    st.close()


@pytest.mark.parametrize("dim_d", [300, 401, 581])
def test_retry_rule_on_every_noise_layout(gpu, dim_d, monkeypatch):
    """The x0.8 rule with the noise vector in LDS (D <= 384) and without it (deep columns re-read / regenerate the
    vector per attempt): a test hook lowers the iteration budget so that every attempt of a row is abandoned."""
    from hydromodel_amd.digest import ColumnTables, ForcingDigest
    from hydromodel_amd.ensemble import pressure_head
    from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
    params = default_parameters()
    cols = ColumnTables(params, synthetic_well(dim_d))
    forcing = ForcingDigest(params, synthetic_forcing_frame(1), cols)
    y0, _ = pressure_head(cols, cols.por_raw)
    N, D = 3, cols.dim_d
    rng = np.random.default_rng(4)
    base = rng.standard_normal((N, D))
    # a regular run first: the state after 4 rows, nothing abandoned
    ref = gpu.EnsembleStepper(cols, forcing, N)
    ref.set_state(y0); ref.set_noise_host(base)
    ref.step_rows(1, 4, fresh_noise=np.zeros((0,)))
    y_ref = ref.get_state()
    assert ref.counters()["failed_attempts"] == 0 and np.array_equal(ref.get_noise_base(), base)
    ref.close()
    monkeypatch.setenv("HYDROCOL_DEBUG_MAX_ITER", "3")
    st = gpu.EnsembleStepper(cols, forcing, N)
    st.set_state(y0); st.set_noise_host(base)
    out = st.step_rows(1, 2, fresh_noise=np.zeros((0,)), want_stats=True)
    c = st.counters()
    assert (out["stats"][:, :, 4] == 5).all() and c["guard_trips"] == 2 * N * 5 and c["failed_attempts"] == 2 * N * 5
    # two rows x five abandoned attempts: the base vector carries x0.8 ten times, applied one at a time
    expect = base.copy()
    for _ in range(10):
        expect = expect * 0.8
    assert np.array_equal(st.get_noise_base(), expect)
    assert np.array_equal(st.get_state(), np.tile(y0, (N, 1)))          # no step was accepted: y0 is returned
    # a refresh row: the fresh vector lives one row, the base vector is untouched
    row = 48
    fresh = rng.standard_normal((1, N, D))
    before = st.get_noise_base()
    st.step_rows(row, 1, fresh_noise=fresh)
    assert np.array_equal(st.get_noise_base(), before)
    st.close()
    # Philox mode: the same rule through the per-member scale; afterwards a regular run proceeds
    ph = gpu.EnsembleStepper(cols, forcing, N)
    ph.set_state(y0); ph.set_noise_philox(5, 0)
    ph.step_rows(1, 1)
    assert ph.counters()["failed_attempts"] == N * 5
    ph.close()
    assert np.isfinite(y_ref).all()
