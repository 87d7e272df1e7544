#!/usr/bin/env python3
"""bench.py -- ensemble column-days/s of the MI355X Richards-column stepper.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE SIMULATED DAY (48 half-hour forcing rows) for every member of the rank's
shard.  Workload (BASELINE.json configs[2], the one the metric is quoted on): 262 144 members
per GPU, D = 300 depth nodes, 10-year synthetic forcing digest (175 200 rows), fp64, Philox
noise generated in-kernel, shared initial condition from the member-0 spin-up; the timed region
covers the K days after the W warm-up days (a prefix of the 10-year run -- SURVEY.md §8d; default
K = 30: days 2..31.  The first days after the spin-up are the costliest, a whole year of the 1-year
forcing sustains ~206 k column-days/s: DESIGN.md §5 "Soak").
Members shard across ranks with no communication while stepping ("weak" scaling: per-GPU
members fixed); the single collective -- the int64 all-reduce of the per-row water-table
moments over RCCL -- runs after the timed region and is reported separately.

Prints ONE JSON line (rank 0).  roofline.achieved uses the algorithmic bytes of SURVEY.md §8d,
(16*D + 16) B per column-step, over the step kernel's mean launch duration measured with HIP
events on the library's stream.  Two more legs run AFTER the timed region on rank 0 of a one-GPU run:
`sustained` -- 16 384 members through the first 365 days of the same forcing digest (what the stepper holds
over a whole simulated year, with the failed-attempt / iteration-guard counters) -- and `cpu_baseline` -- the C oracle
(oracle/, a port of the same algorithm) on every host core this process may use, for >= 30 s.
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

ROWS_PER_DAY = 48
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
# HBM bytes per member per 48-row launch at D=300, from rocprofv3 PMC passes on this launch shape
# (profiles/r02_pmc_fetch_size.csv, r02_pmc_write_size.csv; `tools/prof_traffic.py 65536`, dispatch 12; N = 65 536):
# 2 x FETCH_SIZE (gfx950 counts half of the fetched bytes -- confirmed by dispatch 9 of the same run, a calibration
# launch that only loads and stores psi: 153 600 KiB each way, FETCH_SIZE 78 024.5, WRITE_SIZE 154 624)
# + WRITE_SIZE = (2 x 78 574.5 + 178 688) KiB / 65 536 members = 5.1 KiB.
PMC_HBM_BYTES_PER_MEMBER_LAUNCH_D300 = (2 * 78574.5 + 178688.0) * 1024.0 / 65536.0
# fp64 work per column-step at D=300 from rocprofv3 SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 summed over the 31 ensemble
# launches of this very bench (profiles/r02_pmc_sq_f64_bench.csv; wave instructions per column-step: 2 019.5 / 3 674.7 /
# 7 846.5 / 685.3 -- unchanged from round 1; x64 lanes, FMA = 2 flop) -- the secondary, honest roofline.
PMC_F64_FLOP_PER_COLUMN_STEP_D300 = (2019.5 + 3674.7 + 685.3 + 2 * 7846.5) * 64.0
FP64_VALU_PEAK_TFLOPS = 78.6    # 256 CU x 64 FMA/clk x 2 x 2.4 GHz


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30,
                    help="timed simulated days (one 48-row launch each); the default month takes ~30 s on one MI355X")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--members", type=int, default=262144, help="members per GPU")
    ap.add_argument("--depth", type=int, default=300)
    ap.add_argument("--years", type=int, default=10)
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=30.0, help="wall time of the CPU leg (BASELINE.md §3: >= 30 s)")
    ap.add_argument("--no-sustained", action="store_true")
    ap.add_argument("--sustained-members", type=int, default=16384)
    ap.add_argument("--sustained-days", type=int, default=365)
    ap.add_argument("--ic-file", default="", help="npz cache of the spun-up initial condition (written if missing)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    return ap.parse_args()


def host_cores():
    """(cores this process may use, cores of the host).  The GPU box gives a one-GPU job a SHARE of the host through a
    cgroup CPU quota (cpu.max "1600000 100000" = 16 cores) while `nproc` / the affinity mask still show all 256:
    threads beyond the quota only take turns."""
    total = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = total
    quota = None
    try:                                            # cgroup v2
        q, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:                                        # cgroup v1
            q = float(Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())
            period = float(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        usable = min(usable, max(1, int(round(quota))))
    return max(1, min(usable, total)), total


def cpu_baseline(cols, forcing, psi0, threads, seconds, first_row=1 + ROWS_PER_DAY, days_per_member=10, seed=5):
    """C oracle (CPU port of the same algorithm) on `threads` host threads for ~`seconds` of wall time: every thread
    integrates members (own noise stream) through the same `days_per_member` days the GPU leg starts its timed region
    on, one after another, until the deadline; only completed members count."""
    from oracle.oracle import Oracle, lib
    lib()
    D = cols.dim_d
    rows = days_per_member * ROWS_PER_DAY
    n_ref = int(forcing.refresh[first_row:first_row + rows].sum())
    deadline = time.perf_counter() + seconds

    def work(tid):
        orc = Oracle(cols, forcing.surface_evap)
        rng = np.random.default_rng(seed + tid)
        done = 0
        while True:
            orc.run(forcing, psi0, rng.standard_normal(D), rng.standard_normal((max(n_ref, 1), D)), first_row,
                    first_row + rows)
            done += 1
            if time.perf_counter() >= deadline:
                return done

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        done = sum(ex.map(work, range(threads)))
    wall = time.perf_counter() - t0
    days = done * days_per_member
    usable, total = host_cores()
    return {"value": days / wall, "unit": "column-days/s", "cores": threads, "kind": "port",
            "per_core": days / wall / threads, "host_cores_total": total, "host_cores_usable": usable,
            "sample": f"{done} members x {rows} rows (days {(first_row - 1) // ROWS_PER_DAY + 1}.."
                      f"{(first_row - 1) // ROWS_PER_DAY + days_per_member}, D={D}) of the same forcing, C oracle, "
                      f"{threads} threads (every core this process may use: affinity mask and cgroup quota), {wall:.1f} s wall"}


def sustained_leg(cols, forcing, psi0, members, days, seed, device):
    """What the stepper sustains over a whole simulated year: `members` members through the first `days` days of the
    same digest, in 30-day requests that the library splits into launches of its own choosing, with the failed-attempt
    and iteration-guard counters of the run."""
    from hydromodel_amd.ensemble import EnsembleSimulation
    days = min(days, (forcing.dim_t - 1) // ROWS_PER_DAY)
    sim = EnsembleSimulation(cols, forcing, members, seed=seed, device=device, psi0=psi0)
    t0 = time.perf_counter()
    done = 0
    while done < days:                       # 30-day requests; the library sizes its launches by the ensemble
        n = min(30, days - done)             # (hc_set_rows_per_launch: 192 rows per launch at 16 384 members)
        sim.advance(ROWS_PER_DAY * n)
        done += n
    wall = time.perf_counter() - t0
    cnt = sim.stepper.counters()
    mean_cm, std_cm = sim.wtd_mean_std()
    last = sim.next_row - 1
    out = {"value": members * days / wall, "unit": "column-days/s", "members": members, "days": days,
           "wall_s": wall, "kernel_s": sim.kernel_ms * 1e-3, "launches": sim.launches,
           "failed_attempts": cnt["failed_attempts"], "guard_trips": cnt["guard_trips"],
           "failed_attempts_per_member_year": cnt["failed_attempts"] / members * 365.0 / days,
           "wtd_mean_cm_last_row": float(mean_cm[last]), "wtd_std_cm_last_row": float(std_cm[last]),
           "wtd_std_cm_max": float(np.nanmax(std_cm[1:last + 1])),
           "workload": f"{members} members x D={cols.dim_d}, days 1..{days} of the same {forcing.dim_t}-row digest"}
    sim.close()
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    n_dev = max(torch.cuda.device_count(), 1)
    dev_index = local_rank % n_dev                 # one rank per GPU; a gloo rehearsal may share a card
    if world > 1:
        torch.cuda.set_device(dev_index)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)
    dev = torch.device("cuda", dev_index)
    local_rank = dev_index

    from hydromodel_amd.digest import ColumnTables, ForcingDigest
    from hydromodel_amd.ensemble import PHILOX_DRAW_SPINUP, EnsembleSimulation, allreduce_stepper_moments, spinup_on_gpu
    from hydromodel_amd.stepper import EnsembleStepper
    from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well

    params = default_parameters()
    cols = ColumnTables(params, synthetic_well(args.depth))
    forcing = ForcingDigest(params, synthetic_forcing_frame(args.years), cols)
    D, N = cols.dim_d, args.members
    need_rows = 1 + (args.warmup + args.steps) * ROWS_PER_DAY
    if need_rows > forcing.dim_t:
        raise SystemExit(f"forcing has {forcing.dim_t} rows, need {need_rows}")

    # shared initial condition: every rank computes the same member-0 spin-up (deterministic)
    probe = EnsembleStepper(cols, forcing, 1, device=local_rank)
    probe.set_noise_philox(args.seed, 0)
    n_rnd0 = probe.philox_normals(0, PHILOX_DRAW_SPINUP)
    probe.close()
    ic_file = Path(args.ic_file) if args.ic_file else None
    if ic_file is not None and ic_file.exists():
        # profiling runs: reuse the initial condition of an earlier run, so that every step_kernel launch rocprofv3
        # sees is one of the ensemble launches (the spin-up is ~110 one-member launches of the same kernel)
        saved = np.load(ic_file)
        psi0, spin_iters = saved["psi0"], int(saved["iterations"])
        assert psi0.shape == (D,)
    else:
        psi0, spin_iters, _ = spinup_on_gpu(cols, forcing, n_rnd0, device=local_rank)
        if ic_file is not None and rank == 0:
            ic_file.parent.mkdir(parents=True, exist_ok=True)
            np.savez(ic_file, psi0=psi0, iterations=spin_iters)

    sim = EnsembleSimulation(cols, forcing, N, seed=args.seed, device=local_rank,
                             member_offset=rank * N, psi0=psi0)
    for _ in range(args.warmup):
        sim.advance(ROWS_PER_DAY)
    sim.kernel_ms, sim.launches = 0.0, 0

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sim.advance(ROWS_PER_DAY)          # hc_step_rows synchronises the library's stream
    sync()
    elapsed = time.perf_counter() - t0
    t_el = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
    elapsed = float(t_el.item())

    # the one collective of the path: moments all-reduce (outside the timed region)
    t1 = time.perf_counter()
    moments = allreduce_stepper_moments(sim.stepper, dev)      # device-to-device export, RCCL all-reduce, one copy back
    torch.cuda.synchronize()
    allreduce_s = time.perf_counter() - t1
    mean_cm, std_cm = sim.wtd_mean_std(moments)
    last_row = sim.next_row - 1

    col_days = float(N) * world * args.steps
    value = col_days / elapsed
    rows_per_launch = ROWS_PER_DAY * args.steps / max(sim.launches, 1)
    bytes_per_launch = float(N) * rows_per_launch * (16 * D + 16)
    launch_ms = sim.kernel_ms / max(sim.launches, 1)
    achieved = bytes_per_launch / (launch_ms * 1e-3) / 1e9

    out = {
        "metric": "ensemble column-days/sec", "value": value, "unit": "column-days/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": ("BASELINE.json configs[2] (x8 GPUs = configs[3]): " if (N == 262144 and D == 300) else "")
                               + f"{N} members/GPU x D={D}, {args.years}-yr half-hourly synthetic forcing "
                               f"({forcing.dim_t} rows), vrettas_fung + Stratified, ET+LF on; timed prefix = "
                               f"days {args.warmup + 1}..{args.warmup + args.steps}",
                   "members_per_gpu": N, "depth_nodes": D, "rows_per_step": ROWS_PER_DAY,
                   "noise": "philox4x32-10 in-kernel", "parallelism": f"members sharded x{world}, no data-path collective"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS,
                     "traffic": (PMC_HBM_BYTES_PER_MEMBER_LAUNCH_D300 * N
                                 if (D == 300 and abs(rows_per_launch - 48) < 1e-9) else None),
                     "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on the same launch shape "
                                       "(profiles/r02_pmc_*_size.csv), scaled by members; psi stays in LDS for the "
                                       "48 rows of a launch, so HBM sees ~1/45 of the algorithmic bytes",
                     "kernel": "hc::step_kernel", "launch_ms": launch_ms, "launches": sim.launches,
                     "algorithmic_bytes_per_launch": bytes_per_launch,
                     "note": "path is fp64-VALU/recurrence bound (SURVEY.md §8d): ~24 RHS evaluations per "
                             "column-step at ~10^2 flop per byte of state"},
        "valu_f64": ({"achieved": value * ROWS_PER_DAY * PMC_F64_FLOP_PER_COLUMN_STEP_D300 / 1e12 / world,
                      "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s per GPU",
                      "frac": value * ROWS_PER_DAY * PMC_F64_FLOP_PER_COLUMN_STEP_D300 / 1e12 / world
                      / FP64_VALU_PEAK_TFLOPS,
                      "source": "fp64 instruction mix per column-step from rocprofv3 PMC (profiles/README.md)"}
                     if D == 300 else None),
        "moments_allreduce_s": allreduce_s,
        "wtd_mean_cm_last_row": float(mean_cm[last_row]), "wtd_std_cm_last_row": float(std_cm[last_row]),
        "spinup_iterations": spin_iters,
    }
    sim.close()
    if rank == 0 and world == 1 and not args.no_sustained:
        out["sustained"] = sustained_leg(cols, forcing, psi0, args.sustained_members, args.sustained_days, args.seed,
                                         local_rank)
    elif rank == 0:
        out["sustained"] = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        threads = args.cpu_threads or host_cores()[0]
        out["cpu_baseline"] = cpu_baseline(cols, forcing, psi0, threads, args.cpu_seconds,
                                           first_row=1 + args.warmup * ROWS_PER_DAY)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
