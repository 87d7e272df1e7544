#!/usr/bin/env python3
"""bench.py -- ensemble column-days/s of the MI355X Richards-column stepper.

    python bench.py --gpus N --steps K --warmup W          (N > 1: this process starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (same thing, pre-launched)

A "step" is ONE SIMULATED DAY (48 half-hour forcing rows) for every member of the rank's shard.

--workload ensemble (default; BASELINE.json configs[2], the one the metric is quoted on; x N GPUs = configs[3]):
262 144 members per GPU, D = 300 depth nodes, 10-year synthetic forcing digest (175 200 rows), fp64, Philox noise
generated in-kernel, shared initial condition from the member-0 spin-up; the timed region covers the K days after
the W warm-up days (a prefix of the 10-year run -- SURVEY.md §8d; default K = 30: days 2..31).  Members shard across
ranks with no communication while stepping ("weak" scaling: per-GPU members fixed); the single collective -- the int64
all-reduce of the per-row water-table moments over RCCL -- runs after the timed region and is reported separately.

--workload sweep (BASELINE.json configs[4]): a P-point (n, a0, psi_sat) grid (default 8 x 8 x 8 = 512) x
4 096 members each, D = 300, 1-year forcing, every point from its own spin-up; whole points are dealt to ranks
round-robin ("strong" scaling: the grid is fixed), all of a rank's points advance in ONE launch per day.

Prints ONE JSON line (rank 0).  roofline.achieved uses the algorithmic bytes of SURVEY.md §8d, (16*D + 16) B per
column-step, over the step kernel's mean launch duration measured with HIP events on the library's stream.  More legs
run AFTER the timed region on rank 0 of a one-GPU ensemble run: `sustained` -- 16 384 members through the first 365 days
of the same (10-year, quiet) digest; `sustained_heavy` -- 16 384 members through the whole 1-YEAR forcing, ten times the
evapo-transpiration per row, the regime with failed BDF attempts; `cpu_baseline` -- the C oracle (oracle/, a port of the
same algorithm) on every host core this process may use, for >= 30 s.
"""
import argparse
import json
import os
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

ROWS_PER_DAY = 48
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FABRIC_PEAK_GBPS = 8600.0       # ... its measured Infinity-Cache gather rate (38 MB table, uniformly random rows): what the
                                # L2 <-> fabric counters of a cache-resident working set are priced against
HBM_ACHIEVABLE_GBPS = 6290.0    # ... its measured streaming rate (float4 copy)
FP64_VALU_PEAK_TFLOPS = 78.6    # 256 CU x 64 FMA/clk x 2 x 2.4 GHz
VALU_ISSUE_SLOTS_PER_S = 256 * 4 * 2.4e9 / 4.0   # 1024 SIMDs, one 64-lane fp64 VALU instruction per 4 cycles at the peak clock
# Per-kernel constants from rocprofv3 PMC passes: profiles/pmc_constants.json (written by tools/pmc_constants.py from the
# counter CSVs it names), keyed by "<depth>/<cell model>" and valid for ONE device-code identity (hc_version()'s kernel
# hash: kernel sources + compile flags + compiler).  A library built from other kernel code gets traffic: null -- counters
# of another build are not this build's traffic.
#   fabric_bytes_per_member_launch: 2 x FETCH_SIZE + WRITE_SIZE, in bytes per member of a 48-row launch.  These counters
#     sit on the L2's memory side: they count Infinity-Cache hits and HBM accesses alike (MI355X_MICROARCH.md).  The factors
#     are measured on known byte counts in the kernel's own access shapes (tools/pmc_calib.hip, profiles/r05_pmc_calib.txt):
#     FETCH_SIZE reads 0.500 of the bytes of raw_buffer_load_b64 (512 B per wave instruction) and of the state's strided
#     global loads, WRITE_SIZE 1.000 / 1.002 of the corresponding stores;
#   f64_flop_per_column_step: SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 wave instructions x 64 lanes, FMA = 2 flop.
PMC_FILE = REPO / "profiles" / "pmc_constants.json"


def pmc_constants(depth, model):
    """(constants or None, why): the committed counter constants of this build's kernel for (depth, cell model)."""
    try:
        table = json.loads(PMC_FILE.read_text())
    except (OSError, ValueError) as e:
        return None, f"{PMC_FILE.name} unreadable ({e})"
    from hydromodel_amd import _lib
    have = _lib.kernel_hash()
    if table.get("kernel_hash") != have:
        return None, (f"{PMC_FILE.name} holds counters of kernel build {table.get('kernel_hash')}, the loaded library is "
                      f"{have}: re-run tools/gpu_r5_pmc.sh + tools/pmc_constants.py")
    rec = table.get("kernels", {}).get(f"{depth}/{model}")
    if rec is None:
        return None, f"no counter pass for depth {depth} / {model} in {PMC_FILE.name}"
    return rec, rec["source"]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed simulated days (one 48-row launch each); default 30 (ensemble: ~30 s) or 2 (sweep: ~60 s)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=("ensemble", "sweep"), default="ensemble")
    ap.add_argument("--members", type=int, default=None,
                    help="ensemble: members per GPU (default 262 144); sweep: members per parameter point (default 4 096)")
    ap.add_argument("--depth", type=int, default=300)
    ap.add_argument("--years", type=int, default=None, help="forcing length (default: 10 for the ensemble, 1 for the sweep)")
    ap.add_argument("--points", type=int, default=512, help="sweep: grid points, a cube (n x a0 x psi_sat)")
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=30.0, help="wall time of the CPU leg (BASELINE.md §3: >= 30 s)")
    ap.add_argument("--no-sustained", action="store_true")
    ap.add_argument("--sustained-members", type=int, default=65536)
    ap.add_argument("--sustained-days", type=int, default=365)
    ap.add_argument("--no-heavy", action="store_true")
    ap.add_argument("--heavy-members", type=int, default=65536)
    ap.add_argument("--heavy-days", type=int, default=120)
    ap.add_argument("--no-n1e6", action="store_true", help="skip the 1 048 576-member leg (north_star's N = 1e6 on one GPU)")
    ap.add_argument("--n1e6-members", type=int, default=1048576)
    ap.add_argument("--n1e6-days", type=int, default=3)
    ap.add_argument("--ic-file", default="", help="npz cache of the spun-up initial condition (written if missing)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--probe-ranks", action="store_true",
                    help="only start the ranks, all-reduce a one and print what the process group saw (no GPU work)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 30 if args.workload == "ensemble" else 2
    if args.members is None:
        args.members = 262144 if args.workload == "ensemble" else 4096
    if args.years is None:
        args.years = 10 if args.workload == "ensemble" else 1
    return args


def launch_ranks(n_ranks):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as CHILD processes (one per
    GPU, torch.distributed.run on 127.0.0.1) and hand their output through.  Runs before this process has imported
    torch or touched the GPU; nothing is re-executed in place, nothing is retried.  Returns the launcher's exit code."""
    # --standalone: the launcher's own c10d store picks a free port on 127.0.0.1 (no bind-close-reuse race)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(n_ranks), str(Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


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
                      f"{threads} threads (the {usable}-core share of the {total}-core host this process may use: "
                      f"affinity mask and cgroup quota), {wall:.1f} s wall"}


def sustained_leg(cols, forcing, psi0, members, days, seed, device, label):
    """What the stepper sustains over a whole simulated year: `members` members through the first `days` days of
    `forcing`, in 30-day requests that the library splits into launches of its own choosing, with the failed-attempt
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
           "workload": f"{members} members x D={cols.dim_d}, days 1..{days} of {label} ({forcing.dim_t} rows)"}
    sim.close()
    return out


def n1e6_leg(cols, forcing, psi0, members, days, seed, device):
    """north_star's N = 1e6 on ONE GPU: `members` (default 1 048 576) x D members from the shared initial condition, one
    warm-up day + `days` timed days (one 48-row launch each).  2.5 GB of state; the same kernel, a 4x longer launch."""
    from hydromodel_amd.ensemble import EnsembleSimulation
    sim = EnsembleSimulation(cols, forcing, members, seed=seed, device=device, psi0=psi0)
    sim.advance(ROWS_PER_DAY)
    sim.kernel_ms, sim.launches = 0.0, 0
    t0 = time.perf_counter()
    per_launch = []
    for _ in range(days):
        per_launch.append(sim.advance(ROWS_PER_DAY)["kernel_ms"])
    wall = time.perf_counter() - t0
    m = sim.moments()
    last = sim.next_row - 1
    out = {"value": members * days / wall, "unit": "column-days/s", "members": members, "days": days, "wall_s": wall,
           "launch_ms": sim.kernel_ms / max(sim.launches, 1), "launch_ms_min": min(per_launch),
           "launch_ms_max": max(per_launch), "launches": sim.launches,
           "members_counted_last_row": int(m[0][last]),
           "workload": f"{members} members x D={cols.dim_d}, days 2..{1 + days} of the same digest, one GPU"}
    sim.close()
    return out


def valu_object(column_steps_per_s, pmc, source):
    """The fp64 vector unit, two ways: flop against the FMA peak, and issue slots -- the kernel's VALU wave instructions (fp64
    arithmetic or not) against one instruction per SIMD per 4 cycles, the bound that binds here."""
    if not pmc:
        return None
    tflops = column_steps_per_s * pmc["f64_flop_per_column_step"] / 1e12
    out = {"achieved": tflops, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s per GPU", "frac": tflops / FP64_VALU_PEAK_TFLOPS,
           "source": source}
    insts = pmc.get("valu_wave_instructions_per_column_step")
    if insts:
        arith = sum(pmc["f64_wave_instructions_per_column_step"].values())
        out["issue"] = {"valu_wave_instructions_per_column_step": insts, "fp64_arithmetic_share": arith / insts,
                        "frac_of_issue_slots": column_steps_per_s * insts / VALU_ISSUE_SLOTS_PER_S,
                        "what": "SQ_INSTS_VALU per column-step x column-steps/s over 1024 SIMDs x 2.4 GHz / 4 cycles; every VALU "
                                "instruction priced at 4 cycles (fp64 transcendentals take 16), the clock at its peak"}
    return out


def roofline_object(bytes_per_launch, launch_ms, per_launch_ms, launches, pmc, pmc_why, members, rows_per_launch, kernel,
                    note):
    """The `roofline` entry of the JSON line.  `achieved` = algorithmic bytes / mean launch time against the HBM peak (the
    contract's figure).  `traffic` = L2 <-> fabric bytes per launch by the committed counter passes of THIS kernel build
    (null + the reason otherwise); `fabric` prices that traffic: it is Infinity-Cache and HBM traffic together -- the
    counters cannot tell them apart -- so its peak is the guide's measured Infinity-Cache rate, with the HBM streaming
    rate beside it."""
    achieved = bytes_per_launch / (launch_ms * 1e-3) / 1e9
    traffic = (pmc["fabric_bytes_per_member_launch"] * members
               if (pmc and abs(rows_per_launch - 48) < 1e-9) else None)
    fabric = None
    if traffic is not None:
        gbps = traffic / (launch_ms * 1e-3) / 1e9
        fabric = {"achieved": gbps, "peak": FABRIC_PEAK_GBPS, "unit": "GB/s", "frac": gbps / FABRIC_PEAK_GBPS,
                  "hbm_achievable": HBM_ACHIEVABLE_GBPS, "frac_of_hbm_achievable": gbps / HBM_ACHIEVABLE_GBPS,
                  "traffic_over_algorithmic": traffic / bytes_per_launch,
                  "what": "FETCH_SIZE x 2 + WRITE_SIZE: requests between the L2s and the fabric, Infinity-Cache hits "
                          "included (the integrator's per-wave region cycling through the cache hierarchy, not psi "
                          "re-read from HBM: profiles/r05_fabric_residency.txt)"}
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic, "fabric": fabric,
            "traffic_source": (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on the same launch shape ({pmc['source']}), "
                               "scaled by members") if pmc else pmc_why,
            "kernel": kernel, "launch_ms": launch_ms,
            "launch_ms_min": min(per_launch_ms) if per_launch_ms else None,
            "launch_ms_max": max(per_launch_ms) if per_launch_ms else None, "launches": launches,
            "algorithmic_bytes_per_launch": bytes_per_launch, "note": note}


def probe_ranks(args, rank, world):
    """--probe-ranks: what did the launcher start?  Every rank joins the process group and adds a one."""
    import torch
    import torch.distributed as dist
    if world > 1:
        if args.backend == "nccl":
            dev_index = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(dev_index)
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
            t = torch.ones(1, dtype=torch.int64, device=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)
            t = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(t)
        seen, ranks, backend = int(t.item()), dist.get_world_size(), dist.get_backend()
        dist.destroy_process_group()
    else:
        seen, ranks, backend = 1, 1, None
    if rank == 0:
        print(json.dumps({"probe": True, "n_gpus": world, "ranks": ranks, "ranks_counted": seen, "backend": backend}))


def sweep_grid(n_points):
    """n in linspace(1.5, 3.0, k), a0 in geomspace(0.003, 0.03, k), psi_sat in -geomspace(1e-3, 1.0, k)."""
    k = round(n_points ** (1.0 / 3.0))
    if k ** 3 != n_points:
        raise SystemExit(f"--points {n_points} is not a cube (n x a0 x psi_sat grid)")
    return [(float(n), float(a0), float(ps)) for n in np.linspace(1.5, 3.0, k) for a0 in np.geomspace(0.003, 0.03, k)
            for ps in -np.geomspace(1e-3, 1.0, k)]


def run_sweep(args, rank, world, dev, dist):
    import torch
    from hydromodel_amd.digest import ColumnTables, ForcingDigest
    from hydromodel_amd.ensemble import SweepSimulation, check_sweep_points, deal_points
    from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
    params = default_parameters()
    grid = sweep_grid(args.points)
    pts = [{"Soil_Properties": {"n": n, "a0": a0, "psi_sat": ps}} for n, a0, ps in grid]
    merged = check_sweep_points(params, pts)
    mine = deal_points(len(pts), rank, world)
    well = synthetic_well(args.depth)
    cols_list = [ColumnTables(merged[k], well) for k in mine]
    forcing = ForcingDigest(params, synthetic_forcing_frame(args.years), cols_list[0])
    D, M, P = cols_list[0].dim_d, args.members, len(pts)
    need_rows = 1 + (args.warmup + args.steps) * ROWS_PER_DAY
    if need_rows > forcing.dim_t:
        raise SystemExit(f"forcing has {forcing.dim_t} rows, need {need_rows}")
    t_spin = time.perf_counter()
    sim = SweepSimulation(cols_list, forcing, M, seed=args.seed, device=dev.index, point_ids=mine)
    spin_s = time.perf_counter() - t_spin
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
    per_launch_ms = []
    for _ in range(args.steps):
        step = sim.advance(ROWS_PER_DAY)
        per_launch_ms.append(step["kernel_ms"] / max(step["launches"], 1))
    sync()
    elapsed = time.perf_counter() - t0
    t_el = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
    elapsed = float(t_el.item())

    n_local = len(mine) * M
    cnt = sim.stepper.counters()
    cost = sim.stepper.point_costs().astype(np.float64)
    m = sim.moments()
    last = sim.next_row - 1
    top = np.argsort(-cost)[:5]
    col_days = float(P) * M * args.steps
    rows_per_launch = ROWS_PER_DAY * args.steps / max(sim.launches, 1)
    bytes_per_launch = float(n_local) * rows_per_launch * (16 * D + 16)
    launch_ms = sim.kernel_ms / max(sim.launches, 1)
    pmc, pmc_why = pmc_constants(D, "generic")
    # every rank's points into one [P][3][T] table (the product's own assembly: multigpu.assemble_points); its count row
    # says that every point arrived complete
    from hydromodel_amd import multigpu

    class _Group:       # bench.py's process group behind the product's interface
        rank, world, backend, dist = 0, 1, None, None

        def __init__(self):
            self.rank, self.world = rank, world
            if world > 1:
                self.dist, self.backend = dist, dist.get_backend()

        def device_index(self):
            return dev.index
        allreduce_sum = multigpu.Ranks.allreduce_sum

    t_asm = time.perf_counter()
    whole, psi0_all, _ = multigpu.assemble_points(_Group(), P, {k: {"moments": m[j], "psi0": sim.psi0[j],
                                                                 "spinup_iterations": int(sim.spinup_iters[j])}
                                                             for j, k in enumerate(mine)}, forcing.dim_t, D)
    assemble_s = time.perf_counter() - t_asm
    counts_last = whole[:, 0, last]
    out = {
        "metric": "ensemble column-days/sec", "value": col_days / elapsed, "unit": "column-days/s",
        "n_gpus": world, "ranks": world, "backend": (dist.get_backend() if world > 1 else None),
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": (("BASELINE.json configs[4]: " if (P == 512 and M == 4096 and D == 300) else "")
                                + f"{P}-point (n, a0, psi_sat) grid x {M} members each, D={D}, {args.years}-yr half-hourly "
                                f"synthetic forcing ({forcing.dim_t} rows), vrettas_fung + Stratified, ET+LF on, one "
                                f"spin-up per point; timed prefix = days {args.warmup + 1}..{args.warmup + args.steps}"),
                   "points": P, "members_per_point": M, "points_per_gpu": len(mine), "depth_nodes": D,
                   "rows_per_step": ROWS_PER_DAY, "noise": "philox4x32-10 in-kernel",
                   "parallelism": f"whole points dealt round-robin x{world}, no data-path collective; one all-reduce "
                                  f"assembles the [P][3][T] result"},
        "sweep_assembled": {"points": int(P), "points_complete_last_row": int((counts_last == M).sum()),
                            "members_per_point_last_row_min_max": [int(counts_last.min()), int(counts_last.max())],
                            "table_shape": list(whole.shape), "initial_cond_shape": list(psi0_all.shape),
                            "assemble_s": assemble_s,
                            # identity of the assembled statistics: equal at any rank count (integer sums meeting zeros)
                            "sha1": __import__("hashlib").sha1(np.ascontiguousarray(whole[:, :, :last + 1]).tobytes()).hexdigest()},
        "roofline": roofline_object(bytes_per_launch, launch_ms, per_launch_ms, sim.launches, pmc, pmc_why, n_local,
                                    rows_per_launch, "hc::step_kernel<5, generic exponents> (rank 0)",
                                    "fp64-VALU/recurrence bound (SURVEY.md §8d); the costliest points set the pace; "
                                    "counter constants are those of the n = 1.7 point"),
        "valu_f64": valu_object((col_days / elapsed) * ROWS_PER_DAY / world, pmc,
                                "fp64 instruction mix per column-step of the n = 1.7 point from rocprofv3 PMC (profiles/README.md); "
                                "costlier points do more"),
        "sweep_rank0": {"spinup_s": spin_s, "spinup_iterations_max": int(np.max(np.abs(sim.spinup_iters))),
                        "spinup_capped": int((np.asarray(sim.spinup_iters) < 0).sum()),
                        "failed_attempts": cnt["failed_attempts"], "guard_trips": cnt["guard_trips"],
                        "rhs_evaluations_per_column_step_min_median_max": [
                            float(v) / (M * (args.warmup + args.steps) * ROWS_PER_DAY)
                            for v in (cost.min(), np.median(cost), cost.max())],
                        "costliest_points": [{"point": int(mine[j]), "n": grid[mine[j]][0], "a0": grid[mine[j]][1],
                                              "psi_sat": grid[mine[j]][2],
                                              "rhs_evaluations_per_column_step":
                                                  float(cost[j]) / (M * (args.warmup + args.steps) * ROWS_PER_DAY)}
                                             for j in top],
                        "wtd_mean_cm_last_row_min_max": [float(v) for v in (
                            cols_list[0].z[0] + 5.0 * (m[:, 1, last] / m[:, 0, last]).min(),
                            cols_list[0].z[0] + 5.0 * (m[:, 1, last] / m[:, 0, last]).max())]},
        "sustained": None, "sustained_heavy": None, "cpu_baseline": None,
    }
    sim.close()
    if int((counts_last == M).sum()) != P:
        if world > 1:
            dist.destroy_process_group()
        raise SystemExit(f"the assembled sweep table holds {int((counts_last == M).sum())} complete points of {P}")
    return out


def run_ensemble(args, rank, world, dev, dist):
    import torch
    from hydromodel_amd.digest import ColumnTables, ForcingDigest
    from hydromodel_amd.ensemble import PHILOX_DRAW_SPINUP, EnsembleSimulation, allreduce_stepper_moments, spinup_on_gpu
    from hydromodel_amd.stepper import EnsembleStepper
    from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
    local_rank = dev.index
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
    per_launch_ms = []
    for _ in range(args.steps):
        step = sim.advance(ROWS_PER_DAY)          # hc_step_rows synchronises the library's stream
        per_launch_ms.append(step["kernel_ms"] / max(step["launches"], 1))
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
    members_seen = int(moments[0][last_row])                   # count row of the reduced table: every rank's members

    col_days = float(N) * world * args.steps
    value = col_days / elapsed
    rows_per_launch = ROWS_PER_DAY * args.steps / max(sim.launches, 1)
    bytes_per_launch = float(N) * rows_per_launch * (16 * D + 16)
    launch_ms = sim.kernel_ms / max(sim.launches, 1)
    pmc, pmc_why = pmc_constants(D, "special")

    out = {
        "metric": "ensemble column-days/sec", "value": value, "unit": "column-days/s",
        "n_gpus": world, "ranks": world, "backend": (dist.get_backend() if world > 1 else None),
        "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": ("BASELINE.json configs[2] (x8 GPUs = configs[3]): " if (N == 262144 and D == 300) else "")
                               + f"{N} members/GPU x D={D}, {args.years}-yr half-hourly synthetic forcing "
                               f"({forcing.dim_t} rows), vrettas_fung + Stratified, ET+LF on; timed prefix = "
                               f"days {args.warmup + 1}..{args.warmup + args.steps}",
                   "members_per_gpu": N, "depth_nodes": D, "rows_per_step": ROWS_PER_DAY,
                   "noise": "philox4x32-10 in-kernel", "parallelism": f"members sharded x{world}, no data-path collective"},
        "roofline": roofline_object(bytes_per_launch, launch_ms, per_launch_ms, sim.launches, pmc, pmc_why, N,
                                    rows_per_launch, "hc::step_kernel",
                                    "the path is fp64-VALU/recurrence bound (SURVEY.md §8d, `valu_f64` below): ~19 RHS "
                                    "evaluations per column-step at ~10^2 flop per byte of state; psi stays on chip "
                                    "for the 48 rows of a launch"),
        "valu_f64": valu_object(value * ROWS_PER_DAY / world, pmc,
                                "fp64 instruction mix per column-step from rocprofv3 PMC (profiles/README.md)"),
        "moments_allreduce_s": allreduce_s, "members_in_reduced_moments": members_seen,
        # identity of the reduced statistics: equal at any rank count (integer sums keyed by global member ids)
        "moments_sha1": __import__("hashlib").sha1(np.ascontiguousarray(np.asarray(moments)[:, :last_row + 1]).tobytes()).hexdigest(),
        "wtd_mean_cm_last_row": float(mean_cm[last_row]), "wtd_std_cm_last_row": float(std_cm[last_row]),
        "spinup_iterations": spin_iters,
    }
    sim.close()
    if members_seen != N * world:
        if world > 1:
            dist.destroy_process_group()
        raise SystemExit(f"the reduced moment table counts {members_seen} members, expected {N} x {world}")
    solo = rank == 0 and world == 1
    out["sustained"] = (sustained_leg(cols, forcing, psi0, args.sustained_members, args.sustained_days, args.seed,
                                      local_rank, f"the same {args.years}-year digest")
                        if solo and not args.no_sustained else None)
    if solo and not args.no_heavy:
        # the heavy regime: the reference spreads the annual demand over the file length (simulation.py:320-325), so the
        # 1-year forcing carries ten times the evapo-transpiration per row of the 10-year digest
        forcing1 = ForcingDigest(params, synthetic_forcing_frame(1), cols)
        out["sustained_heavy"] = sustained_leg(cols, forcing1, psi0, args.heavy_members, args.heavy_days, args.seed,
                                               local_rank, "the 1-year forcing")
    else:
        out["sustained_heavy"] = None
    out["n1e6"] = (n1e6_leg(cols, forcing, psi0, args.n1e6_members, args.n1e6_days, args.seed, local_rank)
                   if solo and not args.no_n1e6 else None)
    if solo and not args.no_cpu_baseline:
        threads = args.cpu_threads or host_cores()[0]
        out["cpu_baseline"] = cpu_baseline(cols, forcing, psi0, threads, args.cpu_seconds,
                                           first_row=1 + args.warmup * ROWS_PER_DAY)
    else:
        out["cpu_baseline"] = None
    return out


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))            # before torch / the GPU are touched by this process
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.workload == "sweep" and world > args.points:
        # (before any collective: a rank without a point would otherwise die alone and leave the others in a barrier)
        raise SystemExit(f"bench.py: --gpus {world} ranks for --points {args.points}: a rank would have no parameter point")
    if args.probe_ranks:
        return probe_ranks(args, rank, world)
    import torch
    import torch.distributed as dist
    n_dev = max(torch.cuda.device_count(), 1)
    if args.backend == "nccl" and world > n_dev:
        raise SystemExit(f"bench.py: {world} RCCL ranks need {world} GPUs, {n_dev} visible (use --backend gloo to rehearse)")
    dev_index = local_rank % n_dev                 # one rank per GPU; a gloo rehearsal may share a card
    if world > 1:
        torch.cuda.set_device(dev_index)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: the process group has {dist.get_world_size()} ranks, --gpus says {args.gpus}")
    dev = torch.device("cuda", dev_index)
    out = (run_sweep if args.workload == "sweep" else run_ensemble)(args, rank, world, dev, dist)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
