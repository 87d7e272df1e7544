"""Time the step kernel of a development build: python tools/prof_variant.py <lib.so> [N]"""
import os, sys, pathlib
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
from hydromodel_amd import _lib
_lib.LIB_PATH = pathlib.Path(sys.argv[1]).resolve()
import ctypes
_have = ctypes.CDLL(str(_lib.LIB_PATH))          # an older build lacks the newer entry points: bind what it has
_lib.EXPORTS = {k: v for k, v in _lib.EXPORTS.items() if hasattr(_have, k)}
from helpers import digest, golden
from hydromodel_amd.stepper import EnsembleStepper
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
_, cols, forcing = digest(300)
g = golden("g1_tables_300.npz")
if "hc_add_point" not in _lib.EXPORTS:            # round-1 library: the whole daylight byte is its daylight flag
    import copy
    forcing = copy.copy(forcing)
    forcing.wet_season = forcing.wet_season * 0
st = EnsembleStepper(cols, forcing, N)
st.set_state(g["initial_cond"]); st.set_noise_philox(42, 0)
out = st.step_rows(1, 48)
import hashlib
print(sys.argv[1], "kernel_ms", round(out["kernel_ms"], 2), "col-days/s", round(N / (out["kernel_ms"] * 1e-3)),
      "state sha", hashlib.sha1(st.get_state().tobytes()).hexdigest()[:12])
st.close()
