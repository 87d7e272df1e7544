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
                "--sustained-members", "2048", "--sustained-days", "3", "--heavy-members", "1024", "--heavy-days", "2",
                "--n1e6-members", "8192", "--n1e6-days", "2"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "sustained", "sustained_heavy",
                "ranks", "backend"):
        assert key in out, key
    assert out["ranks"] == 1 and out["members_in_reduced_moments"] == 4096
    hv = out["sustained_heavy"]
    assert hv["members"] == 1024 and hv["days"] == 2 and hv["value"] > 1e3 and "1-year" in hv["workload"]
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1 and out["dtype"] == "f64"
    assert out["unit"] == "column-days/s" and out["value"] > 1e4 and out["vs_baseline"] is None
    r = out["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # achieved = algorithmic bytes per launch / mean launch duration (HIP events on the library's stream)
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["algorithmic_bytes_per_launch"] == 4096 * 48 * (16 * 300 + 16)
    # traffic: this round's committed counter constants when they belong to the loaded library's kernel build, else null
    # WITH the reason -- never a constant typed into bench.py
    consts = json.loads((REPO / "profiles" / "pmc_constants.json").read_text())
    from hydromodel_amd import _lib
    if consts["kernel_hash"] == _lib.kernel_hash():
        assert r["traffic"] == consts["kernels"]["300/special"]["fabric_bytes_per_member_launch"] * 4096
        assert "profiles/r05_pmc_300_special_fetch.csv" in r["traffic_source"] and out["valu_f64"]["frac"] > 0.05
        # the fabric leg prices that traffic (Infinity-Cache + HBM requests) against the guide's measured rates
        f = r["fabric"]
        assert f["unit"] == "GB/s" and f["peak"] == 8600.0 and f["hbm_achievable"] == 6290.0
        assert abs(f["achieved"] - r["traffic"] / (r["launch_ms"] * 1e-3) / 1e9) < 1e-6 * f["achieved"]
        assert abs(f["traffic_over_algorithmic"] - r["traffic"] / r["algorithmic_bytes_per_launch"]) < 1e-9
        # the bound that binds: VALU wave instructions of the counter pass against one per SIMD per 4 cycles
        k = consts["kernels"]["300/special"]
        iss = out["valu_f64"]["issue"]
        assert iss["valu_wave_instructions_per_column_step"] == k["valu_wave_instructions_per_column_step"]
        slots = out["value"] * 48 * iss["valu_wave_instructions_per_column_step"] / (1024 * 2.4e9 / 4)
        assert abs(iss["frac_of_issue_slots"] - slots) < 1e-9 and 0.0 < slots < 1.0
        assert 0.5 < iss["fp64_arithmetic_share"] < 0.8
    else:
        assert r["traffic"] is None and r["fabric"] is None and "re-run tools/gpu_r5_pmc.sh" in r["traffic_source"]
    # per-launch spread of the timed steps (one launch per step at this size)
    assert r["launches"] == 2 and r["launch_ms_min"] <= r["launch_ms"] <= r["launch_ms_max"]
    n6 = out["n1e6"]
    assert n6["members"] == 8192 and n6["days"] == 2 and n6["members_counted_last_row"] == 8192 and n6["value"] > 1e4
    assert len(out["moments_sha1"]) == 40
    c = out["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "members" in c["sample"]
    assert c["cores"] == c["host_cores_usable"] and abs(c["per_core"] * c["cores"] - c["value"]) < 1e-9 * c["value"]
    su = out["sustained"]
    assert su["unit"] == "column-days/s" and su["members"] == 2048 and su["days"] == 3 and su["value"] > 1e3
    assert su["failed_attempts"] >= 0 and su["guard_trips"] >= 0 and 1 <= su["launches"] <= 3
    assert out["value"] / c["value"] > 10            # a reported baseline, not a target -- but it must be the same unit


def test_bench_two_ranks_share_the_moments():
    """The driver's spelling, `python bench.py --gpus 2 ...` with NO launcher around it: bench.py starts the two ranks
    itself (they share the one card here, over gloo; RCCL needs one GPU per rank)."""
    two = _run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo",
                "--members", "2048", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"])
    assert two["n_gpus"] == 2 and two["ranks"] == 2 and two["backend"] == "gloo"
    assert two["scaling"] == "weak" and two["cpu_baseline"] is None and two["sustained"] is None
    assert two["members_in_reduced_moments"] == 4096
    # whole-job aggregate: both ranks' members over the max-over-ranks time
    assert abs(two["value"] - 2 * 2048 * two["steps"] / (two["ms_per_step"] * 1e-3 * two["steps"])) < 1e-6 * two["value"]
    one = _run([sys.executable, "bench.py", "--members", "4096", "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
                "--no-sustained", "--no-heavy", "--no-n1e6"])
    # members are keyed by their global id: 2 x 2048 sharded == 4096 on one rank, to the last bit of the statistics
    assert two["wtd_mean_cm_last_row"] == one["wtd_mean_cm_last_row"]
    assert two["wtd_std_cm_last_row"] == one["wtd_std_cm_last_row"]


def test_bench_five_ranks_share_the_card_and_reduce_to_the_one_rank_statistics():
    """The N-rank path with more than two ranks on the one card a test box has (gloo; the box allows six processes on its GPU,
    this test is the sixth): 5 x 512 members in contiguous blocks = 2 560 members on one rank, to the bit of the reduced
    int64 moments; a sweep of 8 points dealt 2 + 2 + 2 + 1 + 1 assembles the table one rank computes.  (World size 8 is
    rehearsed on CPU: tests/test_multigpu_cpu.py.)"""
    common = ["--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-sustained", "--no-heavy", "--no-n1e6"]
    five = _run([sys.executable, "bench.py", "--gpus", "5", "--backend", "gloo", "--members", "512"] + common)
    one = _run([sys.executable, "bench.py", "--members", "2560"] + common)
    assert five["ranks"] == 5 and five["n_gpus"] == 5 and five["backend"] == "gloo"
    assert five["members_in_reduced_moments"] == 5 * 512 == one["members_in_reduced_moments"]
    assert five["moments_sha1"] == one["moments_sha1"]
    assert five["wtd_mean_cm_last_row"] == one["wtd_mean_cm_last_row"] and five["wtd_std_cm_last_row"] == one["wtd_std_cm_last_row"]
    sw = ["--workload", "sweep", "--points", "8", "--members", "64", "--steps", "1", "--warmup", "1"]
    s5 = _run([sys.executable, "bench.py", "--gpus", "5", "--backend", "gloo"] + sw)
    s1 = _run([sys.executable, "bench.py"] + sw)
    assert s5["ranks"] == 5 and s5["config"]["points_per_gpu"] == 2
    for line in (s5, s1):
        a = line["sweep_assembled"]
        assert a["points"] == 8 and a["points_complete_last_row"] == 8 and a["members_per_point_last_row_min_max"] == [64, 64]
    assert s5["sweep_assembled"]["sha1"] == s1["sweep_assembled"]["sha1"]


def test_bench_under_a_launcher_still_works():
    """... and the pre-launched spelling of the contract (`python -m torch.distributed.run ... bench.py --gpus 2`)."""
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", "29517", "bench.py", "--gpus", "2", "--backend", "gloo",
                "--members", "1024", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"])
    assert two["n_gpus"] == 2 and two["ranks"] == 2 and two["members_in_reduced_moments"] == 2048


def test_bench_refuses_more_rccl_ranks_than_gpus():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--members", "1024", "--steps", "1"], cwd=REPO, env=env,
                       capture_output=True, text=True, timeout=600)
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible: the RCCL launch is legitimate here")
    assert p.returncode != 0 and "RCCL ranks need" in (p.stderr + p.stdout)


def test_bench_sweep_workload_contract():
    """--workload sweep (BASELINE configs[4]) at a reduced size: same JSON contract, the generic kernel, the per-point
    census; two gloo ranks deal the points round-robin and cover the grid once."""
    one = _run([sys.executable, "bench.py", "--workload", "sweep", "--points", "8", "--members", "256", "--steps", "1",
                "--warmup", "1"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "sweep_rank0"):
        assert key in one, key
    assert one["scaling"] == "strong" and one["config"]["points"] == 8 and one["config"]["points_per_gpu"] == 8
    r = one["roofline"]
    assert r["algorithmic_bytes_per_launch"] == 8 * 256 * 48 * (16 * 300 + 16)
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    sw = one["sweep_rank0"]
    assert sw["spinup_capped"] == 0 and len(sw["costliest_points"]) == 5
    lo, mid, hi = sw["rhs_evaluations_per_column_step_min_median_max"]
    assert 5.0 < lo <= mid <= hi
    two = _run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--workload", "sweep", "--points", "8",
                "--members", "256", "--steps", "1", "--warmup", "1"])
    assert two["n_gpus"] == 2 and two["config"]["points_per_gpu"] == 4
    assert abs(two["value"] - 8 * 256 / (two["ms_per_step"] * 1e-3)) < 1e-6 * two["value"]
    # the result is ONE table whoever ran which point (VERDICT r3 missing 1): every point complete on both lines
    for line in (one, two):
        a = line["sweep_assembled"]
        assert a["points"] == 8 and a["points_complete_last_row"] == 8 and a["members_per_point_last_row_min_max"] == [256, 256]
        assert a["table_shape"][:2] == [8, 3] and a["initial_cond_shape"] == [8, 300]

