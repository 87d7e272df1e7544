"""N > 1 path on CPU: world_size-2 gloo run of the single collective (moments all-reduce)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hydromodel_amd.ensemble import allreduce_moments
from hydromodel_amd.stepper import moments_to_mean_std


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, T, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(100 + rank)
    n_local = 1000 + 17 * rank
    idx = rng.integers(40, 80, size=(T, n_local))
    m = np.stack([np.full(T, n_local), idx.sum(axis=1), (idx ** 2).sum(axis=1)]).astype(np.int64)
    total = allreduce_moments(m, torch.device("cpu"))
    np.save(os.path.join(out_dir, f"idx_{rank}.npy"), idx)
    np.save(os.path.join(out_dir, f"tot_{rank}.npy"), total)
    dist.destroy_process_group()


def test_moments_allreduce_world2_is_exact(tmp_path):
    world, T = 2, 37
    mp.spawn(_worker, args=(world, _free_port(), T, str(tmp_path)), nprocs=world, join=True)
    idx = np.concatenate([np.load(tmp_path / f"idx_{r}.npy") for r in range(world)], axis=1)
    tots = [np.load(tmp_path / f"tot_{r}.npy") for r in range(world)]
    assert np.array_equal(tots[0], tots[1])                       # every rank holds the same table
    assert np.array_equal(tots[0][0], np.full(T, idx.shape[1]))
    assert np.array_equal(tots[0][1], idx.sum(axis=1))
    assert np.array_equal(tots[0][2], (idx ** 2).sum(axis=1))
    mu, sd = moments_to_mean_std(tots[0], 5.0)
    assert np.allclose(mu, 5.0 * idx.mean(axis=1), rtol=1e-13)
    assert np.allclose(sd, 5.0 * idx.std(axis=1), rtol=1e-9)


def test_allreduce_is_identity_without_process_group():
    m = np.arange(12, dtype=np.int64).reshape(3, 4)
    assert allreduce_moments(m, torch.device("cpu")) is m


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` WITHOUT a launcher around it: the parent starts two ranks itself.
    --probe-ranks stops after the process group is up, so this runs on the CPU box over gloo."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    repo = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--probe-ranks"], cwd=repo,
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and out["ranks_counted"] == 2 and out["backend"] == "gloo"


def test_bench_refuses_a_world_that_is_not_gpus():
    import subprocess
    import sys
    from pathlib import Path
    repo = Path(__file__).resolve().parent.parent
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "4", "--probe-ranks"], cwd=repo, env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr


def test_points_are_dealt_round_robin():
    from hydromodel_amd.ensemble import deal_points
    parts = [deal_points(512, r, 8) for r in range(8)]
    assert sorted(sum(parts, [])) == list(range(512)) and all(len(p) == 64 for p in parts)
    assert parts[3][:3] == [3, 11, 19]
    assert deal_points(3, 5, 8) == []


def test_bench_refuses_more_ranks_than_sweep_points():
    """`--workload sweep --gpus N` with N > points would leave a rank without a point, dead before the first
    barrier of the others -- refused up front, before any process group exists."""
    import subprocess
    import sys
    from pathlib import Path
    repo = Path(__file__).resolve().parent.parent
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29998")
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--workload", "sweep", "--points", "1"], cwd=repo, env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "a rank would have no parameter point" in p.stderr
