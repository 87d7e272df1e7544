"""The product's multi-GPU path on CPU (gloo, world sizes 2 and 8): member shards, sweep-result assembly, the CLI's self-launch.

Reference surface: /root/reference/code/berkeley_hydro_main.py:128-137 (one `sim.run(); sim.saveResults()`); here N ranks
deliver one file (hydromodel_amd/multigpu.py, cli.py)."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch.multiprocessing as mp

from hydromodel_amd import multigpu
from hydromodel_amd.ensemble import deal_points
from hydromodel_amd.synthetic import default_parameters, synthetic_well, write_forcing_csv, write_site_information

REPO = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_member_shards_partition_the_ensemble():
    for n, w in ((2_097_152, 8), (4096, 3), (7, 8), (1, 1)):
        parts = [multigpu.shard(n, r, w) for r in range(w)]
        assert parts[0][0] == 0 and parts[-1][1] == n
        assert all(parts[r][1] == parts[r + 1][0] for r in range(w - 1))
        sizes = [hi - lo for lo, hi in parts]
        assert max(sizes) - min(sizes) <= 1
    assert multigpu.shard(2_097_152, 3, 8) == (786_432, 1_048_576)      # BASELINE configs[3]: 262 144 per GPU


def _fake_point(k, T, D):
    rng = np.random.default_rng(1000 + k)
    return {"moments": rng.integers(0, 2**40, size=(3, T)).astype(np.int64), "psi0": rng.standard_normal(D) * 1e3,
            "spinup_iterations": 100 + k}


def _assemble_worker(rank, world, port, P, T, D, out_dir, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HYDROCOL_DIST_BACKEND="gloo")
    ranks = multigpu.Ranks(expect=world)
    mine = deal_points(P, rank, world)
    if mode == "duplicate":
        mine = sorted(set(mine) | {0})                    # both ranks claim point 0
    local = {k: _fake_point(k, T, D) for k in mine}
    try:
        m, psi0, spin = multigpu.assemble_points(ranks, P, local, T, D)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), m=m, psi0=psi0, spin=spin)
    except RuntimeError as e:
        Path(out_dir, f"err{rank}.txt").write_text(str(e))
    ranks.close()


@pytest.mark.parametrize("P", [8, 1])       # 1 point on 2 ranks: rank 1 has nothing and still joins the collectives
def test_sweep_assembly_world2_keeps_every_point_bit_for_bit(tmp_path, P):
    world, T, D = 2, 29, 17
    mp.spawn(_assemble_worker, args=(world, _free_port(), P, T, D, str(tmp_path), "ok"), nprocs=world, join=True)
    got = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    for k in range(P):
        ref = _fake_point(k, T, D)
        for g in got:                                     # every rank holds the whole table
            assert np.array_equal(g["m"][k], ref["moments"])
            assert np.array_equal(g["psi0"][k].view(np.int64), ref["psi0"].view(np.int64))      # to the bit
            assert int(g["spin"][k]) == ref["spinup_iterations"]


@pytest.mark.parametrize("P", [512, 509])   # BASELINE configs[4] on 8 ranks; 509: ranks 5-7 are dealt one point fewer
def test_sweep_assembly_world8_keeps_every_point_bit_for_bit(tmp_path, P):
    world, T, D = 8, 7, 5
    mp.spawn(_assemble_worker, args=(world, _free_port(), P, T, D, str(tmp_path), "ok"), nprocs=world, join=True)
    deals = [len(deal_points(P, r, world)) for r in range(world)]
    assert sum(deals) == P and max(deals) - min(deals) <= 1 and (P % world == 0 or deals[-1] == deals[0] - 1)
    got = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    ref = [_fake_point(k, T, D) for k in range(P)]
    for g in got:                                         # every one of the 8 ranks holds the whole table
        assert np.array_equal(g["m"], np.stack([r["moments"] for r in ref]))
        assert np.array_equal(g["psi0"].view(np.int64), np.stack([r["psi0"] for r in ref]).view(np.int64))
        assert g["spin"].tolist() == [r["spinup_iterations"] for r in ref]


def _shard_worker(rank, world, port, n_members, T, out_dir):
    """What the ensemble path reduces: every rank's (count, sum idx, sum idx^2) rows over ITS member block."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HYDROCOL_DIST_BACKEND="gloo")
    ranks = multigpu.Ranks(expect=world)
    lo, hi = multigpu.shard(n_members, ranks.rank, ranks.world)
    ids = np.arange(lo, hi, dtype=np.int64)
    m = np.zeros((3, T), dtype=np.int64)
    for t in range(T):                                    # a stand-in water-table index that depends on the GLOBAL member id
        idx = (ids * 7 + t) % 300
        m[:, t] = (ids.size, idx.sum(), (idx * idx).sum())
    total = ranks.allreduce_sum(m)
    if ranks.rank == 0:
        np.save(os.path.join(out_dir, "total.npy"), total)
    ranks.close()


def test_member_shards_reduce_to_the_whole_ensemble_on_eight_ranks(tmp_path):
    """configs[3]'s shape in small: 8 ranks, contiguous member blocks, ONE int64 all-reduce; the reduced count row must
    read N on every row (what cli.py / bench.py check as `members_in_reduced_moments`), the sums those of one rank."""
    world, N, T = 8, 4099, 6                              # 4099 = 8 x 512 + 3: three ranks own one member more
    mp.spawn(_shard_worker, args=(world, _free_port(), N, T, str(tmp_path)), nprocs=world, join=True)
    total = np.load(tmp_path / "total.npy")
    ids = np.arange(N, dtype=np.int64)
    for t in range(T):
        idx = (ids * 7 + t) % 300
        assert total[:, t].tolist() == [N, int(idx.sum()), int((idx * idx).sum())]


def test_sweep_assembly_refuses_a_point_delivered_twice(tmp_path):
    mp.spawn(_assemble_worker, args=(2, _free_port(), 4, 5, 3, str(tmp_path), "duplicate"), nprocs=2, join=True)
    for r in range(2):
        assert "delivered by [2] ranks" in (tmp_path / f"err{r}.txt").read_text()


def test_one_rank_is_the_identity():
    ranks = multigpu.Ranks()
    assert (ranks.rank, ranks.world) == (0, 1)
    a = np.arange(6, dtype=np.int64)
    assert np.array_equal(ranks.allreduce_sum(a), a)
    local = {k: _fake_point(k, 4, 3) for k in range(3)}
    m, psi0, spin = multigpu.assemble_points(ranks, 3, local, 4, 3)
    assert all(np.array_equal(m[k], local[k]["moments"]) for k in range(3)) and spin.tolist() == [100, 101, 102]


def test_gpus_option_precedence():
    assert multigpu.requested_gpus(None, {}) == 1
    assert multigpu.requested_gpus(None, {"Ensemble": {"GPUs": 8}}) == 8
    assert multigpu.requested_gpus(2, {"Ensemble": {"GPUs": 8}}) == 2


def _cli_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env["HYDROCOL_DIST_BACKEND"] = "gloo"
    return env


def test_cli_starts_its_own_ranks_and_hands_on_their_failure(tmp_path):
    """`berkeley_hydro_main.py --gpus 2` with no launcher around it: the parent starts two ranks (children, before any GPU
    call); here both fail the same way -- no "Ensemble" block -- and the command ends with the reference's status 1."""
    p = default_parameters()
    p["Site_Information"] = str(write_site_information(tmp_path / "site.json", {10: synthetic_well(200)}))
    p["Data_Filename"] = str(write_forcing_csv(tmp_path / "forcing.csv", 1))
    (tmp_path / "p.json").write_text(json.dumps(p))
    r = subprocess.run([sys.executable, str(REPO / "berkeley_hydro_main.py"), "--params", str(tmp_path / "p.json"), "--gpus", "2"],
                       cwd=tmp_path, env=_cli_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 1, r.stdout + r.stderr
    assert 'needs an "Ensemble" block' in r.stdout
    assert r.stdout.count(" Simulation water data file:") == 1            # rank 0 reports, rank 1 stays quiet
    assert "Simulation completed" not in r.stdout


def test_cli_refuses_a_world_that_is_not_gpus(tmp_path):
    p = default_parameters()
    p["Site_Information"] = str(write_site_information(tmp_path / "site.json", {10: synthetic_well(200)}))
    p["Data_Filename"] = str(write_forcing_csv(tmp_path / "forcing.csv", 1))
    p["Ensemble"] = {"Members": 8, "GPUs": 4}
    (tmp_path / "p.json").write_text(json.dumps(p))
    env = dict(_cli_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, str(REPO / "berkeley_hydro_main.py"), "--params", str(tmp_path / "p.json")],
                       cwd=tmp_path, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "WORLD_SIZE=1" in r.stdout
