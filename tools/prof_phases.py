"""Diagnostic build (python tools/build_dev.py prof -DHC_PROFILE --cpl 5): cycles per phase of the step kernel's
state machine.  python tools/prof_phases.py [tools/dev/_ab/lib_prof.so]"""
import ctypes as C, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np
so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(R, "tools", "dev", "_ab", "lib_prof.so")
from hydromodel_amd import _lib
_lib.LIB_PATH = __import__("pathlib").Path(so).resolve()
from helpers import digest, golden
from hydromodel_amd.stepper import EnsembleStepper
_, cols, forcing = digest(300)
g = golden("g1_tables_300.npz")
N = 8192
st = EnsembleStepper(cols, forcing, N)
st.set_state(g["initial_cond"]); st.set_noise_philox(42, 0)
out = st.step_rows(1, 48)
prof = (C.c_uint64 * 32)()
st.lib.hc_debug_profile.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
assert st.lib.hc_debug_profile(st.h, prof) == 0
names = ["PH_F0", "PH_F1", "PH_JAC", "PH_JAC_REDO", "PH_NEWTON", "C_JAC_FIN", "C_STEP_BEGIN", "C_STEP_TRY",
         "C_NEWTON_BEGIN", "C_NEWTON_FAIL", "C_ERR_TEST", "C_ACCEPT", "C_SUCCESS", "C_FAIL", "-", "-", "RHS_EVAL"]
tot = sum(prof)
print("kernel_ms", out["kernel_ms"], "total cycles/row/wave", tot / N / 48)
for k, v in enumerate(prof):
    if v:
        print(f"{(names[k] if k < len(names) else str(k)):16s} {100.0 * v / tot:6.2f} %   {v / N / 48:10.0f} cycles/row")
st.close()
