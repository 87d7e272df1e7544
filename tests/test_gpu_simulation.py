"""Config 1: the reference's single-column workflow (Simulation / CLI) with the loop on the GPU."""
import json

import numpy as np
import pytest

from helpers import WELLS, forcing_frame, golden
from hydromodel_amd.synthetic import default_parameters, write_forcing_csv, write_site_information

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def year_run(tmp_path_factory):
    import __graft_entry__ as ge
    ge.build()
    from hydromodel_amd.simulation import Simulation
    tmp = tmp_path_factory.mktemp("sim")
    params = default_parameters()
    params["Site_Information"] = str(write_site_information(tmp / "site.json", {1: WELLS[1]}))
    params["Well_No"] = 1
    sim = Simulation("golden_1", seed=911)
    sim.setupModel(params, forcing_frame(1))
    sim.run()
    return sim


def test_output_dictionary_has_the_reference_layout(year_run):
    out = year_run.output
    T, D = 17520, 101
    assert sorted(out) == sorted(["K_hrc", "K_bkg", "S_eff", "psi_press", "theta_vol", "abs_error",
                                  "wtd_est_cm", "lateral_flow", "transpiration"])
    for k in ("K_hrc", "K_bkg", "S_eff", "psi_press", "theta_vol"):
        assert out[k].shape == (T, D) and np.isfinite(out[k]).all()
    assert out["abs_error"].shape == out["wtd_est_cm"].shape == (T,)
    assert out["lateral_flow"].shape == out["transpiration"].shape == (T - 1,)
    assert np.all(out["S_eff"] <= 1.0 + 1e-12) and np.all(out["S_eff"] > 0.0)
    assert np.all(out["K_hrc"] <= out["K_bkg"] * (1 + 1e-15))


def test_year_long_run_tracks_the_reference(year_run):
    g = golden("g5_traj_1.npz")
    out = year_run.output
    assert np.max(np.abs(year_run.mData["initial_cond"] - g["initial_cond"])) < 0.02     # spin-up, ~115 solves
    # first rows: still on the reference trajectory (the IC differs by ~1e-3 cm after the chaotic spin-up)
    d = np.abs(out["psi_press"][1] - g["rec_y1"][0])
    assert d.max() < 0.02
    ref_idx = np.rint(g["wtd_est_cm"] / 5.0).astype(int)
    idx = np.rint(out["wtd_est_cm"] / 5.0).astype(int)
    diff = np.abs(idx - ref_idx)
    # Never more than one 5-cm cell away over 17 520 rows.  Exact equality is a weaker statement: the
    # handful of rows on which the solver gives up (x0.8 noise damping, richards_pde.py:522) fall on
    # different rows in every implementation -- the event is chaotic in the last bits -- and each one
    # rescales the base noise for the rest of the year (measured: reference 14, C oracle 10, GPU 11 such
    # rows; DESIGN.md "Parity tiers").  The first month, well before the first such event, agrees closely.
    print(f"year run vs the reference: water-table index equal on {(diff == 0).mean():.1%} of {diff.size} rows "
          f"(first 1400: {(diff[:1400] == 0).mean():.1%}, first 2900: {(diff[:2900] == 0).mean():.1%}), "
          f"one cell apart on {(diff == 1).mean():.1%}, never more")
    assert diff.max() <= 1
    assert (diff[:1400] == 0).mean() > 0.97
    assert (diff[:2900] == 0).mean() > 0.85
    assert (diff == 0).mean() >= 0.90          # measured 93.3 % (the oracle from the reference's own IC: 97.2 %); see
    #                                            test_whole_year_reference_oracle_and_gpu_side_by_side for all three
    assert abs(out["abs_error"].mean() - g["abs_error"].mean()) < 2.5      # half a grid cell, year average
    # daily theta profile statistics agree (noise-free diagnostic)
    keep = g["daily_rows"]
    assert np.abs(out["theta_vol"][keep] - g["theta_daily"]).mean() < 2e-3


def test_cli_end_to_end(tmp_path, monkeypatch, capsys):
    import __graft_entry__ as ge
    ge.build()
    from hydromodel_amd import cli
    from hydromodel_amd.simulation import loadResults
    params = default_parameters()
    params["Site_Information"] = str(write_site_information(tmp_path / "site.json", {10: WELLS[1]}))
    params["Data_Filename"] = str(write_forcing_csv(tmp_path / "forcing.csv", 1))
    params["Output_Name"] = "Sim 00"
    (tmp_path / "p.json").write_text(json.dumps(params))
    monkeypatch.chdir(tmp_path)
    cli.run_cli(["berkeley_hydro_main.py", "--params", str(tmp_path / "p.json"), "--seed", "5"])
    text = capsys.readouterr().out
    for msg in (" Model parameters are given correctly.", " Selected model: Vrettas-Fung",
                "Burn in period started", "finished at [itr:", " [Well No. 10] 100: MAE =", " Elapsed time:",
                " Simulation completed."):
        assert msg in text, msg
    files = list(tmp_path.glob("Sim_00.*"))
    assert [f.name for f in files] == ["Sim_00.h5"]          # simulation.py:693
    data = loadResults(files[0])
    assert sorted(data) == sorted(["K_hrc", "K_bkg", "S_eff", "psi_press", "theta_vol", "abs_error",
                                   "wtd_est_cm", "lateral_flow", "transpiration"])
    assert data["psi_press"].shape == (17520, 101) and data["psi_press"].dtype == np.float64
    assert data["transpiration"].shape == (17519,) and data["transpiration"].max() > 0


def test_cli_ensemble_block(tmp_path, monkeypatch, capsys):
    from hydromodel_amd import cli
    params = default_parameters()
    params["Site_Information"] = str(write_site_information(tmp_path / "site.json", {10: WELLS[200]}))
    params["Data_Filename"] = str(write_forcing_csv(tmp_path / "forcing.csv", 1))
    params["Ensemble"] = {"Members": 256, "Seed": 3, "Days": 2}
    (tmp_path / "p.json").write_text(json.dumps(params))
    monkeypatch.chdir(tmp_path)
    cli.run_cli(["berkeley_hydro_main.py", "--params", str(tmp_path / "p.json")])
    from hydromodel_amd.simulation import loadResults
    data = loadResults(tmp_path / "Sim_00_ensemble.h5")
    assert int(data["members"]) == 256 and int(data["rows"]) == 96
    assert np.array_equal(data["moments"][0, 1:97], np.full(96, 256))
    assert np.all(np.isfinite(data["wtd_mean_cm"][1:97]))


def test_year_long_diagnostics_track_the_reference(year_run):
    """transpiration / lateral_flow (simulation.py:629-630) over the same 17 519 solves.  Transpiration is
    demand-limited at this site, so it is insensitive to the chaotic last bits; lateral flow follows the
    water table and inherits its drift."""
    g = golden("g5_traj_1.npz")
    out = year_run.output
    tr, lf = out["transpiration"], out["lateral_flow"]
    assert np.max(np.abs(tr[:48] - g["transpiration"][:48])) < 1e-12
    assert abs(tr.sum() / g["transpiration"].sum() - 1) < 1e-6
    # first day, free-running after the ~110-solve spin-up (IC differs by ~1e-3 cm): measured 8e-4 ... 6e-3 across builds
    assert np.max(np.abs(lf[:48] - g["lateral_flow"][:48]) / (1e-6 + g["lateral_flow"][:48])) < 3e-2
    assert abs(lf.sum() / g["lateral_flow"].sum() - 1) < 0.03
    assert np.corrcoef(lf, g["lateral_flow"])[0, 1] > 0.999


def test_cli_ensemble_numpy_streams_and_member_spinup(tmp_path, monkeypatch, capsys):
    from hydromodel_amd import cli
    from hydromodel_amd.simulation import loadResults
    params = default_parameters()
    params["Site_Information"] = str(write_site_information(tmp_path / "site.json", {10: WELLS[1]}))
    params["Data_Filename"] = str(write_forcing_csv(tmp_path / "forcing.csv", 1))
    params["Ensemble"] = {"Members": 6, "Seed": 911, "Days": 1, "Noise": "NumPy", "Spinup": "Member"}
    (tmp_path / "p.json").write_text(json.dumps(params))
    monkeypatch.chdir(tmp_path)
    cli.run_cli(["berkeley_hydro_main.py", "--params", str(tmp_path / "p.json")])
    data = loadResults(tmp_path / "Sim_00_ensemble.h5")
    assert int(data["members"]) == 6 and data["initial_cond"].shape == (6, 101)
    assert np.array_equal(data["moments"][0, 1:49], np.full(48, 6))
    # member 0 spun up with the reference's own first draw: the reference's initial condition
    assert np.max(np.abs(data["initial_cond"][0] - golden("g5_traj_1.npz")["initial_cond"])) < 0.02


def test_cli_parameter_sweep_block(tmp_path, monkeypatch, capsys):
    """BASELINE config 5 from the reference's own command line: "Ensemble": {"Points": [...]}."""
    from hydromodel_amd import cli
    from hydromodel_amd.simulation import loadResults
    params = default_parameters()
    params["Site_Information"] = str(write_site_information(tmp_path / "site.json", {10: WELLS[200]}))
    params["Data_Filename"] = str(write_forcing_csv(tmp_path / "forcing.csv", 1))
    params["Ensemble"] = {"Members": 64, "Seed": 3, "Days": 2,
                          "Points": [{}, {"Soil_Properties": {"n": 1.7, "a0": 0.012}},
                                     {"Soil_Properties": {"psi_sat": -0.5}, "Hydraulic_Conductivity": {"Sigma_Noise": 1.0}}]}
    (tmp_path / "p.json").write_text(json.dumps(params))
    monkeypatch.chdir(tmp_path)
    cli.run_cli(["berkeley_hydro_main.py", "--params", str(tmp_path / "p.json")])
    data = loadResults(tmp_path / "Sim_00_ensemble.h5")
    assert int(data["points"]) == 3 and int(data["members"]) == 64 and int(data["rows"]) == 96
    assert data["moments"].shape == (3, 3, 17520) and data["initial_cond"].shape == (3, 200)
    assert np.array_equal(data["moments"][:, 0, 1:97], np.full((3, 96), 64))
    assert np.all(np.isfinite(data["wtd_mean_cm"][:, 1:97])) and np.all(data["spinup_iterations"] > 0)
    assert not np.array_equal(data["initial_cond"][0], data["initial_cond"][1])
    # PREDICT from the command line keeps the reference's error unless the repair is requested
    params["Simulation_Flags"]["PREDICT"] = True
    (tmp_path / "q.json").write_text(json.dumps(params))
    with pytest.raises(SystemExit):
        cli.run_cli(["berkeley_hydro_main.py", "--params", str(tmp_path / "q.json")])
    assert "cannot be interpreted as an integer" in capsys.readouterr().out
