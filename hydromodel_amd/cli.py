"""Command line of the reference (``/root/reference/code/berkeley_hydro_main.py``), GPU-backed.

    python3 berkeley_hydro_main.py --params P.json [--data D.csv] [--seed S] [--device K]

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
* ``"Ensemble": {"repair_predict": true}`` with ``Simulation_Flags.PREDICT``: run the repaired predictive lateral flow
   instead of raising the reference's ``TypeError``.
"""
import sys
from pathlib import Path

REQUIRED_KEYS = ("Trees", "Well_No", "Output_Name", "IC_Filename",
                 "Data_Filename", "Water_Content", "Environmental",
                 "Soil_Properties", "Site_Information", "Simulation_Flags",
                 "Hydrological_Model", "Hydraulic_Conductivity")


def validateInputParametersFile(filename):
    """berkeley_hydro_main.py:13-61: key membership only, values are not validated here."""
    import json
    with open(filename, "r") as input_file:
        model_params = json.load(input_file)
        for k in REQUIRED_KEYS:
            if k not in model_params:
                raise ValueError(f" Key: {k}, is not given.")
        print(" Model parameters are given correctly.")
    return model_params


def main(params_file=None, data_file=None, seed=None, device=0):
    """berkeley_hydro_main.py:65-145."""
    import pandas as pd
    if params_file is not None:
        try:
            params_file = Path(params_file)
            params = validateInputParametersFile(params_file)
        except ValueError as e0:
            print(e0)
            sys.exit(1)
    else:
        print(" The simulation can't run without input parameters.")
        sys.exit(1)
    data_file = Path(data_file) if data_file is not None else Path(params["Data_Filename"])
    print(f" Simulation water data file: {data_file}")
    try:
        with open(data_file, "r") as input_file:
            water_data = pd.read_csv(input_file, names=["ID", "Datenum", "Precipitation_cm", "WTD_m"])
        output_name = params["Output_Name"]
        if output_name is None:
            output_name = "Sim_01"
        if seed is None:
            seed = params.get("Seed")
        ens = params.get("Ensemble")
        if ens:
            _run_ensemble(params, water_data, output_name, ens, device)
        else:
            from .simulation import Simulation
            sim_01 = Simulation(output_name, seed=seed, device=device)
            sim_01.setupModel(params, water_data)
            sim_01.run()
            sim_01.saveResults()
    except Exception as e1:  # noqa: BLE001 - the reference converts every failure to exit status 1
        print(e1)
        sys.exit(1)


def _run_ensemble(params, water_data, output_name, ens, device):
    import numpy as np
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
        return _run_sweep(params, forcing, output_name, ens, n_members, rows, device)
    sim = EnsembleSimulation(cols, forcing, n_members, seed=int(ens.get("Seed", 0)), device=device,
                             noise=str(ens.get("Noise", "philox")).lower(),
                             spinup=str(ens.get("Spinup", "shared")).lower())
    done = 0
    while done < rows:
        n = min(48 * 30, rows - done)
        sim.advance(n)
        done += n
        print(f" [Ensemble x{n_members}] {done} rows done")
    moments = sim.moments()
    mean_cm, std_cm = sim.wtd_mean_std(moments)
    from . import hdf5io
    arrays = dict(moments=moments, wtd_mean_cm=mean_cm, wtd_std_cm=std_cm, rows=np.array(rows),
                  members=np.array(n_members), initial_cond=sim.psi0)
    stem = output_name.strip().replace(" ", "_") + "_ensemble"
    if hdf5io.available():      # same container as Simulation.saveResults (simulation.py:697-706)
        out = Path(stem + ".h5")
        hdf5io.write(out, arrays)
    else:
        out = Path(stem + ".npz")
        np.savez_compressed(out, **arrays)
    print(f" Saving the ensemble water-table statistics to: {out}")
    sim.close()


def _run_sweep(params, forcing, output_name, ens, n_members, rows, device):
    """Parameter points x members, one handle (ensemble.SweepSimulation)."""
    import numpy as np
    from . import hdf5io
    from .digest import ColumnTables, load_site_well
    from .ensemble import SweepSimulation, check_sweep_points
    from .stepper import moments_to_mean_std
    # a sweep runs in-kernel Philox noise from one spin-up per point; anything else is refused, not ignored
    if str(ens.get("Noise", "philox")).lower() != "philox":
        raise ValueError(f" Sweep: Ensemble.Noise = {ens.get('Noise')!r} is not supported with Points (philox only).")
    if str(ens.get("Spinup", "point")).lower() not in ("point", "shared"):
        raise ValueError(f" Sweep: Ensemble.Spinup = {ens.get('Spinup')!r} is not supported with Points "
                         f"(one spin-up per parameter point).")
    well = load_site_well(params)
    points = [ColumnTables(mp, well) for mp in check_sweep_points(params, ens["Points"])]
    sim = SweepSimulation(points, forcing, n_members, seed=int(ens.get("Seed", 0)), device=device)
    done = 0
    while done < rows:
        n = min(48 * 30, rows - done)
        sim.advance(n)
        done += n
        print(f" [Sweep {len(points)} points x{n_members}] {done} rows done")
    moments = sim.moments()
    mean_cm, std_cm = moments_to_mean_std(moments, points[0].dz, points[0].z[0])
    arrays = dict(moments=moments, wtd_mean_cm=mean_cm, wtd_std_cm=std_cm, rows=np.array(rows),
                  members=np.array(n_members), points=np.array(len(points)), initial_cond=sim.psi0,
                  spinup_iterations=np.asarray(sim.spinup_iters))
    stem = output_name.strip().replace(" ", "_") + "_ensemble"
    if hdf5io.available():
        out = Path(stem + ".h5")
        hdf5io.write(out, arrays)
    else:
        out = Path(stem + ".npz")
        np.savez_compressed(out, **arrays)
    print(f" Saving the sweep's water-table statistics to: {out}")
    sim.close()


def run_cli(argv=None):
    """berkeley_hydro_main.py:149-177."""
    argv = sys.argv if argv is None else argv
    if len(argv) > 1:
        import argparse
        parser = argparse.ArgumentParser(description=" Berkeley Hydrological Simulation ")
        parser.add_argument("--params", help=" Input file (.json) with simulation parameters.")
        parser.add_argument("--data", help=" Input file (.csv) with simulation data (e.g.: precipitation, wtd).")
        parser.add_argument("--seed", type=int, default=None, help=" Seed of the noise stream (reproducible runs).")
        parser.add_argument("--device", type=int, default=0, help=" GPU ordinal.")
        args = parser.parse_args(argv[1:])
        main(args.params, args.data, args.seed, args.device)
        print(' Simulation completed.')
    else:
        sys.exit('Error: Not enough input parameters.')
