"""dev: compare Simulation's transpiration / lateral_flow series with the golden year run."""
import sys, tempfile, pathlib
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from helpers import WELLS, forcing_frame, golden
from hydromodel_amd.synthetic import default_parameters, write_site_information
from hydromodel_amd.simulation import Simulation
tmp = pathlib.Path(tempfile.mkdtemp())
params = default_parameters()
params["Site_Information"] = str(write_site_information(tmp / "site.json", {1: WELLS[1]}))
params["Well_No"] = 1
sim = Simulation("golden_1", seed=911)
sim.setupModel(params, forcing_frame(1))
sim.run()
g = golden("g5_traj_1.npz")
for key in ("transpiration", "lateral_flow"):
    a, b = sim.output[key], g[key]
    print(key, "sum", a.sum(), b.sum(), "day1 max rel", np.max(np.abs(a[:48] - b[:48]) / (1e-6 + np.abs(b[:48]))),
          "month mean abs", np.abs(a[:1440] - b[:1440]).mean(), "mean", b[:1440].mean(),
          "year mean abs", np.abs(a - b).mean(), "corr", np.corrcoef(a, b)[0, 1])
