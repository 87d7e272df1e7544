"""bench.py end to end on the GPU box: the one-rank line, and the N > 1 launch path rehearsed with two ranks
sharing the card over gloo (RCCL needs one GPU per rank; the driver runs that on an 8-GPU node)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent


def _run(cmd, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    p = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                  # exactly ONE JSON line
    return json.loads(lines[0])


def test_bench_contract_one_rank():
    out = _run([sys.executable, "bench.py", "--members", "4096", "--steps", "2", "--warmup", "1", "--cpu-seconds", "4",
                "--sustained-members", "2048", "--sustained-days", "3"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "sustained"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1 and out["dtype"] == "f64"
    assert out["unit"] == "column-days/s" and out["value"] > 1e4 and out["vs_baseline"] is None
    r = out["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # achieved = algorithmic bytes per launch / mean launch duration (HIP events on the library's stream)
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["algorithmic_bytes_per_launch"] == 4096 * 48 * (16 * 300 + 16)
    c = out["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "members" in c["sample"]
    assert c["cores"] == c["host_cores_usable"] and abs(c["per_core"] * c["cores"] - c["value"]) < 1e-9 * c["value"]
    su = out["sustained"]
    assert su["unit"] == "column-days/s" and su["members"] == 2048 and su["days"] == 3 and su["value"] > 1e3
    assert su["failed_attempts"] >= 0 and su["guard_trips"] >= 0 and 1 <= su["launches"] <= 3
    assert out["value"] / c["value"] > 10            # a reported baseline, not a target -- but it must be the same unit


def test_bench_two_ranks_share_the_moments():
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", "29517", "bench.py", "--gpus", "2", "--backend", "gloo",
                "--members", "2048", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"])
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and two["cpu_baseline"] is None and two["sustained"] is None
    # whole-job aggregate: both ranks' members over the max-over-ranks time
    assert abs(two["value"] - 2 * 2048 * two["steps"] / (two["ms_per_step"] * 1e-3 * two["steps"])) < 1e-6 * two["value"]
    one = _run([sys.executable, "bench.py", "--members", "4096", "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
                "--no-sustained"])
    # members are keyed by their global id: 2 x 2048 sharded == 4096 on one rank, to the last bit of the statistics
    assert two["wtd_mean_cm_last_row"] == one["wtd_mean_cm_last_row"]
    assert two["wtd_std_cm_last_row"] == one["wtd_std_cm_last_row"]
