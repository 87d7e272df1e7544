import os, sys, pathlib
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
from hydromodel_amd import _lib
_lib.LIB_PATH = pathlib.Path(sys.argv[1]).resolve()
from hydromodel_amd.digest import ColumnTables, ForcingDigest
from hydromodel_amd.ensemble import EnsembleSimulation
from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
D, rows, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
params = default_parameters()
cols = ColumnTables(params, synthetic_well(D))
forcing = ForcingDigest(params, synthetic_forcing_frame(1), cols)
sim = EnsembleSimulation(cols, forcing, 256, seed=1)
o = sim.stepper.step_rows(1, rows, want_stats=True, want_wtd=True)
np.savez(out, psi=sim.stepper.get_state(), stats=o["stats"], wtd=o["wtd"], psi0=sim.psi0)
print(out, sim.stepper.counters())
