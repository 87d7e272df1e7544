"""Throughput of the generic-exponent kernel (n != 2 and/or lambda != 1): python tools/prof_generic.py [n] [lam] [N] [D]
(HC_LIB=<lib.so> selects a development build)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
if os.environ.get("HC_LIB"):
    import pathlib
    from hydromodel_amd import _lib
    _lib.LIB_PATH = pathlib.Path(os.environ["HC_LIB"]).resolve()
from hydromodel_amd.digest import ColumnTables, ForcingDigest
from hydromodel_amd.ensemble import EnsembleSimulation
from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
n = float(sys.argv[1]) if len(sys.argv) > 1 else 1.7
lam = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
N = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
D = int(sys.argv[4]) if len(sys.argv) > 4 else 300
params = default_parameters()
params["Soil_Properties"]["n"] = n
params["Hydraulic_Conductivity"]["Lambda_Exponent"] = lam
cols = ColumnTables(params, synthetic_well(D))
forcing = ForcingDigest(params, synthetic_forcing_frame(1), cols)
sim = EnsembleSimulation(cols, forcing, N, seed=1)
sim.advance(48)
sim.kernel_ms = 0.0
sim.advance(96)
print(f"n={n} lambda={lam} N={N} D={D}: {N * 2 / (sim.kernel_ms * 1e-3):.0f} column-days/s "
      f"(kernel {sim.kernel_ms:.1f} ms for 2 days), counters {sim.stepper.counters()}")
sim.close()
