"""Throughput by depth for a given build: python tools/prof_depth.py <lib.so> D [D ...]  (HC_PROF_MEMBERS or 8 192 members, 2 days)"""
import os, sys, pathlib, hashlib
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from hydromodel_amd import _lib
_lib.LIB_PATH = pathlib.Path(sys.argv[1]).resolve()
import ctypes
_have = ctypes.CDLL(str(_lib.LIB_PATH))          # an older build lacks the newer entry points: bind what it has
_lib.EXPORTS = {k: v for k, v in _lib.EXPORTS.items() if hasattr(_have, k)}
OLD = "hc_add_point" not in _lib.EXPORTS
from hydromodel_amd.digest import ColumnTables, ForcingDigest
from hydromodel_amd.ensemble import EnsembleSimulation
from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
params = default_parameters()
if os.environ.get("HC_PROF_SOIL_N"):                 # a generic-exponent kernel instead (n = 2 is the specialised one)
    params["Soil_Properties"]["n"] = float(os.environ["HC_PROF_SOIL_N"])
frame = synthetic_forcing_frame(1)
for D in map(int, sys.argv[2:]):
    cols = ColumnTables(params, synthetic_well(D))
    forcing = ForcingDigest(params, frame, cols)
    if OLD:
        forcing.wet_season = forcing.wet_season * 0      # the round-1 library reads the whole byte as its daylight flag
    NM = int(os.environ.get("HC_PROF_MEMBERS", "8192"))
    sim = EnsembleSimulation(cols, forcing, NM, seed=1)
    sim.advance(48)
    sim.kernel_ms = 0.0
    sim.advance(96)
    print(f"{_lib.LIB_PATH.name} D={D}: {NM * 2 / (sim.kernel_ms * 1e-3):.0f} column-days/s ({NM} members), "
          f"sha {hashlib.sha1(sim.stepper.get_state().tobytes()).hexdigest()[:10]}", flush=True)
    sim.close()
