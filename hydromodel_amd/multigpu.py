"""One process per GPU for the two workloads that shard: ensemble members and whole parameter points.

The reference runs ONE column in ONE process (``/root/reference/code/berkeley_hydro_main.py:128-137``: ``sim.run();
sim.saveResults()``).  Here ``berkeley_hydro_main.py --gpus N`` (or ``"Ensemble": {"GPUs": N}``) starts N ranks of the
same command line -- child processes of ``torch.distributed.run`` on 127.0.0.1, started BEFORE this process has touched
a GPU, never a re-exec -- and every rank runs its share with no communication while stepping:

* ensemble: rank r owns the contiguous member block ``shard(N, r, world)``; the Philox stream is keyed by the GLOBAL
  member id, the shared initial condition is global member 0's spin-up on every rank, and ONE int64 all-reduce of the
  per-row water-table moments ``[3][T]`` ends the run (RCCL over xGMI with the nccl backend);
* sweep (BASELINE config 5): whole points are dealt round-robin (``ensemble.deal_points``); every rank writes ITS points
  into zeroed ``[P][3][T]`` / ``[P][D]`` / ``[P]`` tables and one all-reduce each assembles them (x + 0 is exact, so the
  assembled file holds every point's bits exactly as the rank that ran it produced them).

Rank 0 owns the output file.  Integer moment sums are order-independent: the file is bit-identical at any rank count.
"""
import os
import subprocess
import sys

import numpy as np


def requested_gpus(cli_gpus, params):
    """--gpus wins over "Ensemble": {"GPUs": N}; 1 when neither is given."""
    if cli_gpus:
        return int(cli_gpus)
    ens = params.get("Ensemble") or {}
    return int(ens.get("GPUs", 1) or 1)


def in_rank():
    """True inside a rank started by a launcher (torch.distributed.run sets these)."""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def launch_ranks(n_ranks, script, argv):
    """Start ``n_ranks`` ranks of ``python script argv...`` as children of torch.distributed.run and return the launcher's
    exit code.  Must run before anything in this process has initialised a GPU."""
    # --standalone: the launcher's own c10d store picks a free port on 127.0.0.1 and hands it to the ranks (no
    # bind-close-reuse of a port another process can take in between)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               HYDROCOL_EXPECT_WORLD=str(n_ranks))
    for k in ("MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(n_ranks), str(script)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


class Ranks:
    """The process group of a multi-GPU run (or a stand-in with one rank when there is none)."""

    def __init__(self, expect=None):
        self.rank, self.world, self.local_rank = 0, 1, 0
        self.dist = None
        self.backend = None
        if not in_rank():
            if expect and int(expect) > 1:
                raise RuntimeError(f" {expect} GPUs requested but this process is not a rank of a launcher.")
            return
        import torch.distributed as dist
        self.rank, self.world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        self.local_rank = int(os.environ.get("LOCAL_RANK", self.rank))
        want = expect or os.environ.get("HYDROCOL_EXPECT_WORLD")
        if want and int(want) != self.world:
            raise RuntimeError(f" {want} GPUs requested but the launcher started WORLD_SIZE={self.world} ranks.")
        # nccl (= RCCL) unless a rehearsal asks for gloo (CPU tensors; e.g. several ranks sharing one card in a test)
        self.backend = os.environ.get("HYDROCOL_DIST_BACKEND", "nccl")
        if self.world > 1 or os.environ.get("HYDROCOL_DIST_FORCE"):
            if self.backend == "nccl":
                import torch
                torch.cuda.set_device(self.device_index())
            dist.init_process_group(self.backend, rank=self.rank, world_size=self.world)
            self.dist = dist

    def device_index(self):
        """GPU ordinal of this rank: its local rank, or 0 for every rank when a rehearsal shares one card."""
        return 0 if os.environ.get("HYDROCOL_SHARE_DEVICE") else self.local_rank

    def allreduce_sum(self, array):
        """Element-wise sum over the ranks (int64 or float64 NumPy array); identity with one rank."""
        a = np.ascontiguousarray(array)
        if self.dist is None:
            return a
        import torch
        t = torch.from_numpy(a.copy())
        if self.backend == "nccl":
            t = t.to(torch.device("cuda", self.device_index()))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None


def shard(n_members, rank, world):
    """[lo, hi): the contiguous block of global member ids rank `rank` owns (sizes differ by at most one)."""
    n, r, w = int(n_members), int(rank), int(world)
    base, extra = divmod(n, w)
    lo = r * base + min(r, extra)
    return lo, lo + base + (1 if r < extra else 0)


def assemble_points(ranks, n_points, local, T, D):
    """Sweep result of ALL ranks from each rank's own points.

    ``local`` = {global point index: {"moments" int64 [3][T], "psi0" float64 [D], "spinup_iterations" int}} for the points
    this rank ran (possibly none).  Every rank fills zeroed [P][3][T] / [P][D] / [P] tables with its points and the tables
    are summed over the ranks: a point's entries meet only zeros, so its bits survive unchanged.  ``owners`` counts how
    many ranks delivered each point -- exactly one each, or the sweep was dealt wrongly."""
    P = int(n_points)
    moments = np.zeros((P, 3, T), dtype=np.int64)
    psi0 = np.zeros((P, D), dtype=np.float64)
    spin = np.zeros(P, dtype=np.int64)
    owners = np.zeros(P, dtype=np.int64)
    for k, rec in local.items():
        moments[k] = rec["moments"]
        psi0[k] = rec["psi0"]
        spin[k] = 0 if rec.get("spinup_iterations") is None else int(rec["spinup_iterations"])
        owners[k] = 1
    moments = ranks.allreduce_sum(moments)
    psi0 = ranks.allreduce_sum(psi0)
    spin = ranks.allreduce_sum(spin)
    owners = ranks.allreduce_sum(owners)
    if not np.array_equal(owners, np.ones(P, dtype=np.int64)):
        bad = np.flatnonzero(owners != 1)
        raise RuntimeError(f" Sweep: points {bad[:8].tolist()} were delivered by {owners[bad[:8]].tolist()} ranks (expected one each).")
    return moments, psi0, spin
