"""The product's multi-GPU path on the one GPU a test box has: `berkeley_hydro_main.py --gpus 2` starts two ranks that SHARE
the card (HYDROCOL_SHARE_DEVICE) and talk over gloo; what they write must equal the one-rank run to the bit -- the
moments are integer sums, the assembled per-point tables meet only zeros (hydromodel_amd/multigpu.py).

Reference surface: /root/reference/code/berkeley_hydro_main.py:128-137 (one run, one results file)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from hydromodel_amd.simulation import loadResults
from hydromodel_amd.synthetic import default_parameters, synthetic_well, write_forcing_csv, write_site_information

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent


def _run(tmp, name, params, gpus):
    d = tmp / name
    d.mkdir()
    p = dict(params, Output_Name=name)
    (d / "p.json").write_text(json.dumps(p))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(HYDROCOL_DIST_BACKEND="gloo", HYDROCOL_SHARE_DEVICE="1")
    cmd = [sys.executable, str(REPO / "berkeley_hydro_main.py"), "--params", str(d / "p.json")]
    if gpus > 1:
        cmd += ["--gpus", str(gpus)]
    r = subprocess.run(cmd, cwd=d, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count(" Simulation completed.") == 1
    files = sorted(f.name for f in d.glob(f"{name}_ensemble.*"))
    assert files == [f"{name}_ensemble.h5"], files               # rank 0 wrote it, once
    return loadResults(d / files[0]), r.stdout


def _base(tmp_path, depth=200):
    params = default_parameters()
    params["Site_Information"] = str(write_site_information(tmp_path / "site.json", {10: synthetic_well(depth)}))
    params["Data_Filename"] = str(write_forcing_csv(tmp_path / "forcing.csv", 1))
    return params


def _same(a, b, skip=("gpus",)):
    assert sorted(a) == sorted(b)
    for k in a:
        if k in skip:
            continue
        x, y = np.asarray(a[k]), np.asarray(b[k])
        assert x.shape == y.shape and x.dtype == y.dtype, k
        assert x.tobytes() == y.tobytes(), f"{k} differs between the one-rank and the two-rank run"


def test_sweep_of_eight_points_on_two_ranks_equals_one_rank(tmp_path):
    params = _base(tmp_path)
    pts = [{"Soil_Properties": {"n": n, "a0": a0}} for n in (1.6, 2.0, 2.4, 2.8) for a0 in (0.006, 0.012)]
    params["Ensemble"] = {"Members": 64, "Seed": 11, "Days": 1, "Points": pts}
    one, _ = _run(tmp_path, "one", params, 1)
    two, log = _run(tmp_path, "two", params, 2)
    assert int(two["gpus"]) == 2 and int(one["gpus"]) == 1
    assert two["moments"].shape == (8, 3, one["moments"].shape[2]) and two["initial_cond"].shape == (8, 200)
    assert np.array_equal(two["moments"][:, 0, 1:49], np.full((8, 48), 64))          # every point complete: 64 members per row
    _same(one, two)
    assert "on 2 GPUs" in log


def test_ensemble_sharded_over_two_ranks_equals_one_rank(tmp_path):
    params = _base(tmp_path)
    params["Ensemble"] = {"Members": 250, "Seed": 5, "Days": 2}                      # 125 + 125
    one, _ = _run(tmp_path, "ens1", params, 1)
    two, _ = _run(tmp_path, "ens2", params, 2)
    assert np.array_equal(two["moments"][0, 1:97], np.full(96, 250))
    _same(one, two)


def test_ensemble_with_one_spinup_per_member_assembles_the_initial_conditions(tmp_path):
    params = _base(tmp_path)
    params["Ensemble"] = {"Members": 33, "Seed": 2, "Days": 1, "Spinup": "member", "GPUs": 2}      # 17 + 16, GPUs from the JSON
    two, _ = _run(tmp_path, "mem2", params, 1)           # (no --gpus: Ensemble.GPUs starts the ranks)
    params["Ensemble"]["GPUs"] = 1
    one, _ = _run(tmp_path, "mem1", params, 1)
    assert two["initial_cond"].shape == (33, 200) and int(two["gpus"]) == 2
    _same(one, two)
