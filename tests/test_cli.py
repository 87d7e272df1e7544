"""CLI surface of the reference (code/berkeley_hydro_main.py) -- host logic only, no GPU."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

from hydromodel_amd import cli
from hydromodel_amd.synthetic import default_parameters, synthetic_well, write_forcing_csv, write_site_information

REPO = Path(__file__).resolve().parent.parent


def _params(tmp_path):
    p = default_parameters()
    p["Site_Information"] = str(write_site_information(tmp_path / "site.json", {10: synthetic_well(200)}))
    p["Data_Filename"] = str(tmp_path / "forcing.csv")
    return p


def test_required_keys_are_the_reference_twelve():
    assert len(cli.REQUIRED_KEYS) == 12
    ref = json.load(open(REPO / "tests" / "golden" / "reference_input_keys.json"))
    assert sorted(cli.REQUIRED_KEYS) == sorted(ref["required"])


def test_missing_key_exits_with_status_one(tmp_path, capsys):
    p = _params(tmp_path)
    del p["Trees"]
    (tmp_path / "p.json").write_text(json.dumps(p))
    with pytest.raises(SystemExit) as e:
        cli.main(str(tmp_path / "p.json"))
    assert e.value.code == 1
    assert " Key: Trees, is not given." in capsys.readouterr().out


def test_no_params_exits(capsys):
    with pytest.raises(SystemExit) as e:
        cli.main(None)
    assert e.value.code == 1
    assert "can't run without input parameters" in capsys.readouterr().out


def test_missing_data_file_exits_with_status_one(tmp_path, capsys):
    p = _params(tmp_path)
    (tmp_path / "p.json").write_text(json.dumps(p))
    with pytest.raises(SystemExit) as e:
        cli.main(str(tmp_path / "p.json"))
    assert e.value.code == 1
    out = capsys.readouterr().out
    assert "Model parameters are given correctly." in out and "Simulation water data file" in out


def test_script_without_arguments():
    r = subprocess.run([sys.executable, str(REPO / "berkeley_hydro_main.py")], capture_output=True, text=True)
    assert r.returncode == 1 and "Not enough input parameters" in r.stderr


def test_unknown_well_is_an_error(tmp_path, capsys):
    p = _params(tmp_path)
    p["Well_No"] = 77
    (tmp_path / "p.json").write_text(json.dumps(p))
    write_forcing_csv(tmp_path / "forcing.csv", 1)
    with pytest.raises(SystemExit) as e:
        cli.main(str(tmp_path / "p.json"))
    assert e.value.code == 1
    assert "does not exist in the site information file" in capsys.readouterr().out


def test_predict_mode_fails_like_the_reference(tmp_path, capsys):
    p = _params(tmp_path)
    p["Simulation_Flags"]["PREDICT"] = True
    (tmp_path / "p.json").write_text(json.dumps(p))
    write_forcing_csv(tmp_path / "forcing.csv", 1)
    with pytest.raises(SystemExit) as e:
        cli.main(str(tmp_path / "p.json"))
    assert e.value.code == 1
    assert "cannot be interpreted as an integer" in capsys.readouterr().out
