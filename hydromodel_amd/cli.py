"""Command line of the reference (``/root/reference/code/berkeley_hydro_main.py``), GPU-backed.

    python3 berkeley_hydro_main.py --params P.json [--data D.csv] [--seed S] [--device K] [--gpus N]

Same 12 required JSON keys (:40-43), same header-less 4-column CSV (:115-116), same messages and
exit codes (any error -> message + exit status 1, :138-142).  Additions (the reference tolerates
unknown keys, only membership of the 12 is checked):

* ``--seed`` / JSON ``"Seed"``: seeds ``Simulation(name, seed)``; the reference's CLI cannot be seeded.
* JSON ``"Ensemble": {"Members": N, "Seed": s, "Days": d, "Noise": "philox"|"numpy", "Spinup": "shared"|"member"}``:
  run N stochastic members (in-kernel Philox noise, or the reference's NumPy streams with member 0 on
  ``SeedSequence(seed)`` and member k on ``spawn_key=(k,)``; one shared spin-up or one per member) and write the
  per-row water-table mean / sigma to ``<Output_Name>_ensemble.h5``.
* ``"Ensemble": {..., "Points": [{"Soil_Properties": {"n": 1.7, "a0": 0.012}}, {...}, ...]}``: a parameter sweep
  (BASELINE config 5) -- every entry is merged over the file's own sections to give one parameter point; all points run
  with ``Members`` realisations each in ONE launch per batch of rows, each from its own spin-up, and the output holds
  ``moments [P][3][T]``, ``wtd_mean_cm`` / ``wtd_std_cm [P][T]`` and ``initial_cond [P][D]``.
* ``--gpus N`` / ``"Ensemble": {"GPUs": N}``: one process per GPU (``multigpu.py``).  The command starts its N ranks itself
  (children of ``torch.distributed.run`` on 127.0.0.1) before it touches a GPU; ensemble members shard by contiguous
  blocks, sweep points are dealt round-robin, nothing is exchanged while stepping, ONE all-reduce ends the run (the int64
  moment table over RCCL; for a sweep also the per-point initial conditions) and rank 0 writes
  ``<Output_Name>_ensemble.h5`` once -- the reference's single ``sim.run(); sim.saveResults()``
  (``berkeley_hydro_main.py:128-137``), bit-identical at any N.
* ``"Ensemble": {"repair_predict": true}`` with ``Simulation_Flags.PREDICT``: run the repaired predictive lateral flow
   instead of raising the reference's ``TypeError``.
"""
import sys
from pathlib import Path

REQUIRED_KEYS = ("Trees", "Well_No", "Output_Name", "IC_Filename",
                 "Data_Filename", "Water_Content", "Environmental",
                 "Soil_Properties", "Site_Information", "Simulation_Flags",
                 "Hydrological_Model", "Hydraulic_Conductivity")


def validateInputParametersFile(filename, quiet=False):
    """berkeley_hydro_main.py:13-61: key membership only, values are not validated here."""
    import json
    with open(filename, "r") as fh:
        settings = json.load(fh)
    missing = [key for key in REQUIRED_KEYS if key not in settings]
    if missing:
        raise ValueError(f" Key: {missing[0]}, is not given.")
    if not quiet:
        print(" Model parameters are given correctly.")
    return settings


def _speaks():
    """One voice per run: the single process, or rank 0 of a multi-GPU run."""
    import os
    return int(os.environ.get("RANK", "0")) == 0


def _read_parameters(params_file):
    """The first half of berkeley_hydro_main.py:65-100: no file -> message + exit 1; a bad file -> its message + exit 1.
    (The ranks of a run this command started itself stay quiet: their parent has said it.)"""
    import os
    if params_file is None:
        print(" The simulation can't run without input parameters.")
        sys.exit(1)
    try:
        return validateInputParametersFile(Path(params_file), quiet=("HYDROCOL_EXPECT_WORLD" in os.environ) or not _speaks())
    except ValueError as bad_key:
        print(bad_key)
        sys.exit(1)


def main(params_file=None, data_file=None, seed=None, device=0, gpus=None, _settings=None):
    """berkeley_hydro_main.py:65-145.  Inside a rank of a multi-GPU run (RANK / WORLD_SIZE set by the launcher) the rank
    takes its own GPU and its share of the members / points; only rank 0 reports and writes."""
    import pandas as pd
    from . import multigpu
    params = _settings if _settings is not None else _read_parameters(params_file)
    csv_path = Path(data_file) if data_file is not None else Path(params["Data_Filename"])
    ranks = None
    try:
        n_gpus = multigpu.requested_gpus(gpus, params)
        ranks = multigpu.Ranks(expect=n_gpus if (n_gpus > 1 or multigpu.in_rank()) else None)
        if ranks.world > 1:
            device = ranks.device_index()
        if ranks.rank == 0:
            print(f" Simulation water data file: {csv_path}")
        with open(csv_path, "r") as fh:
            forcing_table = pd.read_csv(fh, names=["ID", "Datenum", "Precipitation_cm", "WTD_m"])
        run_name = params["Output_Name"] if params["Output_Name"] is not None else "Sim_01"
        if seed is None:
            seed = params.get("Seed")
        ens = params.get("Ensemble")
        if ens:
            _run_ensemble(params, forcing_table, run_name, ens, device, ranks)
        elif ranks.world > 1:
            raise ValueError(" --gpus / Ensemble.GPUs needs an \"Ensemble\" block: one column is one GPU's work.")
        else:
            from .simulation import Simulation
            column_run = Simulation(run_name, seed=seed, device=device)
            column_run.setupModel(params, forcing_table)
            column_run.run()
            column_run.saveResults()
    except Exception as failure:  # noqa: BLE001 - the reference converts every failure to exit status 1
        import os
        print(failure if _speaks() else f" [rank {os.environ.get('RANK')}] {failure}")
        # A failure may belong to this rank alone (its GPU, its shard): the peers are then inside, or heading for, the
        # run's one all-reduce, and tearing the communicator down here can block on them.  Leave it alone and end the
        # process with status 1 -- the launcher stops the other ranks and reports the failure.
        sys.stdout.flush()
        if ranks is not None and ranks.world > 1:
            os._exit(1)
        sys.exit(1)
    ranks.close()


def _save(stem, arrays, what, ranks):
    """<stem>.h5 through libhdf5 (same container as Simulation.saveResults, simulation.py:697-706) or .npz; rank 0 only."""
    import numpy as np
    from . import hdf5io
    if ranks.rank != 0:
        return None
    if hdf5io.available():
        out = Path(stem + ".h5")
        hdf5io.write(out, arrays)
    else:
        out = Path(stem + ".npz")
        np.savez_compressed(out, **arrays)
    print(f" Saving the {what} to: {out}")
    return out


def _run_ensemble(params, water_data, output_name, ens, device, ranks):
    import numpy as np
    from . import multigpu
    from .digest import ColumnTables, ForcingDigest, load_site_well
    from .ensemble import EnsembleSimulation
    cols = ColumnTables(params, load_site_well(params))
    forcing = ForcingDigest(params, water_data, cols)
    if cols.flags["PREDICT"] and not ens.get("repair_predict"):
        raise TypeError("'numpy.float64' object cannot be interpreted as an integer")     # richards_pde.py:327-330
    n_members = int(ens.get("Members", 4096))
    days = int(ens.get("Days", (forcing.dim_t - 1) // 48))
    rows = min(days * 48, forcing.dim_t - 1)
    if ens.get("Points"):
        return _run_sweep(params, forcing, output_name, ens, n_members, rows, device, ranks)
    lo, hi = multigpu.shard(n_members, ranks.rank, ranks.world)
    if hi <= lo:
        raise ValueError(f" Ensemble: {n_members} members do not shard over {ranks.world} GPUs (a rank would be empty).")
    sim = EnsembleSimulation(cols, forcing, hi - lo, seed=int(ens.get("Seed", 0)), device=device, member_offset=lo,
                             noise=str(ens.get("Noise", "philox")).lower(),
                             spinup=str(ens.get("Spinup", "shared")).lower())
    done = 0
    while done < rows:
        n = min(48 * 30, rows - done)
        sim.advance(n)
        done += n
        if ranks.rank == 0:
            print(f" [Ensemble x{n_members}{'' if ranks.world == 1 else f' on {ranks.world} GPUs'}] {done} rows done")
    # the run's one collective: int64 (count, sum idx, sum idx^2) per row, exact and order-independent
    moments = ranks.allreduce_sum(np.asarray(sim.moments(), dtype=np.int64))
    mean_cm, std_cm = sim.wtd_mean_std(moments)
    # completeness, row by row: a solved row (its observation lies on the grid) must count every member of every shard,
    # a skipped row nobody (simulation.py:582-588); a run whose rows are all skipped has nothing to check
    expect = np.where(np.asarray(forcing.wtd_obs[1:rows + 1]) >= 0, n_members, 0).astype(np.int64)
    got = np.asarray(moments[0][1:rows + 1], dtype=np.int64)
    if not np.array_equal(got, expect):
        bad = int(np.flatnonzero(got != expect)[0])
        raise RuntimeError(f" Ensemble: the reduced moments hold {int(got[bad])} members on row {bad + 1}, expected "
                           f"{int(expect[bad])} ({int((got != expect).sum())} rows differ).")
    psi0 = np.asarray(sim.psi0)
    extra = {}
    if psi0.ndim == 2 and ranks.world > 1:
        # one spin-up per member: the shards' initial conditions, assembled like a sweep's (zeros elsewhere, summed) when
        # the table is small enough to travel; otherwise rank 0's block, with its member range
        if n_members * psi0.shape[1] * 8 <= 256 * 1024 * 1024:
            table = np.zeros((n_members, psi0.shape[1]))
            table[lo:hi] = psi0
            psi0 = ranks.allreduce_sum(table)
        else:
            extra["initial_cond_members"] = np.array([lo, hi])
    arrays = dict(moments=moments, wtd_mean_cm=mean_cm, wtd_std_cm=std_cm, rows=np.array(rows),
                  members=np.array(n_members), gpus=np.array(ranks.world), initial_cond=psi0, **extra)
    _save(output_name.strip().replace(" ", "_") + "_ensemble", arrays, "ensemble water-table statistics", ranks)
    sim.close()


def _run_sweep(params, forcing, output_name, ens, n_members, rows, device, ranks):
    """Parameter points x members: this rank's points in one handle (ensemble.SweepSimulation), the whole table assembled
    over the ranks (multigpu.assemble_points)."""
    import numpy as np
    from . import multigpu
    from .digest import ColumnTables, load_site_well
    from .ensemble import SweepSimulation, check_sweep_points, deal_points
    from .stepper import moments_to_mean_std
    # a sweep runs in-kernel Philox noise from one spin-up per point; anything else is refused, not ignored
    if str(ens.get("Noise", "philox")).lower() != "philox":
        raise ValueError(f" Sweep: Ensemble.Noise = {ens.get('Noise')!r} is not supported with Points (philox only).")
    if str(ens.get("Spinup", "point")).lower() not in ("point", "shared"):
        raise ValueError(f" Sweep: Ensemble.Spinup = {ens.get('Spinup')!r} is not supported with Points "
                         f"(one spin-up per parameter point).")
    well = load_site_well(params)
    merged = check_sweep_points(params, ens["Points"])
    P = len(merged)
    mine = deal_points(P, ranks.rank, ranks.world)
    points = [ColumnTables(merged[k], well) for k in mine]
    D, T = None, forcing.dim_t
    local = {}
    if mine:
        sim = SweepSimulation(points, forcing, n_members, seed=int(ens.get("Seed", 0)), device=device, point_ids=mine)
        done = 0
        while done < rows:
            n = min(48 * 30, rows - done)
            sim.advance(n)
            done += n
            if ranks.rank == 0:
                print(f" [Sweep {P} points x{n_members}{'' if ranks.world == 1 else f' on {ranks.world} GPUs'}] {done} rows done")
        table = sim.moments()
        for j, k in enumerate(mine):
            local[k] = {"moments": table[j], "psi0": sim.psi0[j],
                        "spinup_iterations": None if sim.spinup_iters is None else int(sim.spinup_iters[j])}
        D = points[0].dim_d
        sim.close()
    if D is None:        # a rank without points still joins the collectives: the grid is the well's, whoever owns it
        D = ColumnTables(merged[0], well).dim_d
    moments, psi0, spin = multigpu.assemble_points(ranks, P, local, T, D)
    ref = points[0] if points else ColumnTables(merged[0], well)
    mean_cm, std_cm = moments_to_mean_std(moments, ref.dz, ref.z[0])
    arrays = dict(moments=moments, wtd_mean_cm=mean_cm, wtd_std_cm=std_cm, rows=np.array(rows),
                  members=np.array(n_members), points=np.array(P), gpus=np.array(ranks.world), initial_cond=psi0,
                  spinup_iterations=spin)
    _save(output_name.strip().replace(" ", "_") + "_ensemble", arrays, "sweep's water-table statistics", ranks)


def run_cli(argv=None):
    """berkeley_hydro_main.py:149-177, plus the self-launch of a multi-GPU run."""
    argv = sys.argv if argv is None else argv
    if len(argv) > 1:
        import argparse
        parser = argparse.ArgumentParser(description=" Berkeley Hydrological Simulation ")
        parser.add_argument("--params", help=" Input file (.json) with simulation parameters.")
        parser.add_argument("--data", help=" Input file (.csv) with simulation data (e.g.: precipitation, wtd).")
        parser.add_argument("--seed", type=int, default=None, help=" Seed of the noise stream (reproducible runs).")
        parser.add_argument("--device", type=int, default=0, help=" GPU ordinal.")
        parser.add_argument("--gpus", type=int, default=None,
                            help=" GPUs of this node to run an \"Ensemble\" block on (one process each; default: Ensemble.GPUs or 1).")
        args = parser.parse_args(argv[1:])
        from . import multigpu
        settings = _read_parameters(args.params)
        n_gpus = multigpu.requested_gpus(args.gpus, settings)
        if n_gpus > 1 and not multigpu.in_rank():
            # the parent only starts the ranks (nothing here has touched a GPU) and hands their exit status on
            script = Path(argv[0]).resolve()
            status = multigpu.launch_ranks(n_gpus, script, argv[1:])
            if status != 0:
                sys.exit(1)
            return                                  # (rank 0 has said " Simulation completed.")
        main(args.params, args.data, args.seed, args.device, args.gpus, _settings=settings)
        if _speaks():
            print(' Simulation completed.')
    else:
        sys.exit('Error: Not enough input parameters.')
