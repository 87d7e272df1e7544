"""HDF5 results container (§8 f3): hydromodel_amd/hdf5io.py over libhdf5, replacing the reference's h5py
calls (code/src/simulation.py:697-706 saveResults, :735-741 loadResults).  Checked against the HDF5
distribution's own tools (h5ls / h5dump) as the independent reader."""
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

from hydromodel_amd import hdf5io
from hydromodel_amd.simulation import Simulation, loadResults

pytestmark = pytest.mark.skipif(not hdf5io.available(), reason="no HDF5 C library in this environment")


def _tool(name):
    for cand in (shutil.which(name), f"/opt/conda/bin/{name}"):
        if cand and Path(cand).exists():
            return cand
    return None


def _fake_output(T=500, D=101, seed=0):
    rng = np.random.default_rng(seed)
    out = {k: rng.standard_normal((T, D)) for k in ("K_hrc", "K_bkg", "S_eff", "psi_press", "theta_vol")}
    out["abs_error"] = rng.random(T)
    out["wtd_est_cm"] = 5.0 * rng.integers(0, D, T)
    out["lateral_flow"] = rng.random(T - 1)
    out["transpiration"] = rng.random(T - 1)
    return out


def test_round_trip_is_bit_exact(tmp_path):
    out = _fake_output()
    out["extreme"] = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 5e-324, 1.7976931348623157e308])
    hdf5io.write(tmp_path / "a.h5", out)
    back = hdf5io.read(tmp_path / "a.h5")
    assert sorted(back) == sorted(out)
    for k in out:
        assert back[k].dtype == np.float64 and back[k].shape == out[k].shape
        assert back[k].tobytes() == out[k].tobytes()


def test_dtypes_shapes_and_empties(tmp_path):
    d = {"i32": np.arange(-5, 5, dtype=np.int32), "u16": np.arange(7, dtype=np.uint16).reshape(7, 1),
         "i64": np.array([[2 ** 62, -2 ** 62]]), "f32": np.linspace(0, 1, 9, dtype=np.float32).reshape(3, 3),
         "scalar": np.float64(2.5), "empty": np.zeros((0, 4)), "flag": np.array([True, False, True]),
         "cube": np.arange(2 * 3 * 4, dtype=np.float64).reshape(2, 3, 4),
         "strided": np.arange(20.0).reshape(4, 5)[:, ::2]}
    hdf5io.write(tmp_path / "b.h5", d)
    back = hdf5io.read(tmp_path / "b.h5")
    assert hdf5io.keys(tmp_path / "b.h5") == sorted(d)          # name order, as h5py iterates
    for k, v in d.items():
        v = np.asarray(v)
        want = v.astype(np.uint8) if v.dtype == np.bool_ else v
        assert back[k].dtype == want.dtype and back[k].shape == want.shape, k
        assert np.array_equal(back[k], want), k


def test_errors(tmp_path):
    with pytest.raises(OSError):
        hdf5io.read(tmp_path / "missing.h5")
    (tmp_path / "junk.h5").write_bytes(b"not an hdf5 file at all")
    with pytest.raises(OSError):
        hdf5io.read(tmp_path / "junk.h5")
    with pytest.raises(TypeError):
        hdf5io.write(tmp_path / "c.h5", {"s": np.array(["a", "b"])})
    with pytest.raises(RuntimeError):
        loadResults(None)                                          # simulation.py:727-729


def test_overwrite_truncates(tmp_path):
    hdf5io.write(tmp_path / "d.h5", {"a": np.zeros(3), "b": np.ones(3)})
    hdf5io.write(tmp_path / "d.h5", {"c": np.ones(2)})            # h5py.File(..., 'w'): truncate if exists
    assert hdf5io.keys(tmp_path / "d.h5") == ["c"]


def test_save_results_writes_the_reference_layout(tmp_path, monkeypatch, capsys):
    monkeypatch.chdir(tmp_path)
    sim = Simulation("Sim 07 ")
    sim.saveResults()                                              # empty output: message only (:684-687)
    assert "Simulation data structure 'output' is empty." in capsys.readouterr().out
    assert not list(tmp_path.iterdir())
    sim.output = _fake_output()
    sim.saveResults()
    assert " Saving the results to: Sim 07 .h5" in capsys.readouterr().out
    path = tmp_path / "Sim_07.h5"                                  # spaces -> underscores (:693)
    assert path.exists()
    back = loadResults(path)
    for k, v in sim.output.items():
        assert np.array_equal(back[k], v)
    h5ls = _tool("h5ls")
    if h5ls:
        text = subprocess.run([h5ls, "-v", str(path)], capture_output=True, text=True, check=True).stdout
        for k in sim.output:
            assert k in text
        assert text.count("deflate") == len(sim.output) and "{4}" in text       # gzip level 4 on every dataset
        assert "Dataset {500/500, 101/101}" in text and "native double" in text


def test_independent_reader_sees_the_same_numbers(tmp_path):
    h5dump = _tool("h5dump")
    if not h5dump:
        pytest.skip("h5dump not installed")
    a = np.array([[1.5, -2.25, 3.0], [4.125, 5.0, -6.5]])
    hdf5io.write(tmp_path / "e.h5", {"psi_press": a})
    text = subprocess.run([h5dump, "-d", "/psi_press", "-w", "0", str(tmp_path / "e.h5")], capture_output=True,
                          text=True, check=True).stdout
    assert "H5T_IEEE_F64LE" in text and "SIMPLE { ( 2, 3 ) / ( 2, 3 ) }" in text
    body = text[text.index("DATA {"):]
    nums = [float(tok.strip(",")) for line in body.splitlines() if line.strip().startswith("(")
            for tok in line.split(":")[1].split()]
    assert nums == a.ravel().tolist()


def test_npz_results_still_load(tmp_path):
    np.savez_compressed(tmp_path / "old.npz", psi_press=np.ones((3, 2)))
    assert np.array_equal(loadResults(tmp_path / "old.npz")["psi_press"], np.ones((3, 2)))
