"""Diagnostic build (python tools/build_dev.py prof -DHC_PROFILE --cpl 5): cycles spent in and entries into each region of
the step kernel (HC_STAMP / HC_RSTAMP sites).  python tools/prof_phases.py [lib_prof.so] [N] [first_row] [--json out.json]"""
import ctypes as C, json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np
args = [a for a in sys.argv[1:] if not a.startswith("--")]
so = args[0] if len(args) > 0 else os.path.join(R, "tools", "dev", "_ab", "lib_prof.so")
N = int(args[1]) if len(args) > 1 else 8192
row0 = int(args[2]) if len(args) > 2 else 1
from hydromodel_amd import _lib
_lib.LIB_PATH = __import__("pathlib").Path(so).resolve()
from helpers import digest, golden
from hydromodel_amd.stepper import EnsembleStepper
DEPTH = int(os.environ.get("HC_PROF_D", "300"))
if DEPTH == 300:
    _, cols, forcing = digest(300)
    ic = golden("g1_tables_300.npz")["initial_cond"]
else:       # any depth: synthetic well, spin-up on the GPU (HC_PROF_D=581: the split-column kernel)
    from hydromodel_amd.digest import ColumnTables, ForcingDigest
    from hydromodel_amd.ensemble import PHILOX_DRAW_SPINUP, spinup_on_gpu
    from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
    params = default_parameters()
    cols = ColumnTables(params, synthetic_well(DEPTH))
    forcing = ForcingDigest(params, synthetic_forcing_frame(1), cols)
    probe = EnsembleStepper(cols, forcing, 1); probe.set_noise_philox(1, 0)
    n0 = probe.philox_normals(0, PHILOX_DRAW_SPINUP); probe.close()
    ic, _, _ = spinup_on_gpu(cols, forcing, n0)
st = EnsembleStepper(cols, forcing, N)
st.set_state(ic); st.set_noise_philox(42, 0)
if row0 > 1:
    st.step_rows(1, row0 - 1)
prof0 = (C.c_uint64 * 32)(); cnt0 = (C.c_uint64 * 32)()
st.lib.hc_debug_profile.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
st.lib.hc_debug_profile_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
st.lib.hc_debug_profile(st.h, prof0); st.lib.hc_debug_profile_counts(st.h, cnt0)
sub0 = (C.c_uint64 * 32)()
st.lib.hc_debug_profile_subcounts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
st.lib.hc_debug_profile_subcounts(st.h, sub0)
out = st.step_rows(row0, 48)
prof = (C.c_uint64 * 32)(); cnt = (C.c_uint64 * 32)(); sub = (C.c_uint64 * 32)()
assert st.lib.hc_debug_profile(st.h, prof) == 0 and st.lib.hc_debug_profile_counts(st.h, cnt) == 0
st.lib.hc_debug_profile_subcounts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
assert st.lib.hc_debug_profile_subcounts(st.h, sub) == 0
prof = [a - b for a, b in zip(prof, prof0)]; cnt = [a - b for a, b in zip(cnt, cnt0)]
sub = [a - b for a, b in zip(sub, sub0)]
names = {0: "PH_F0", 1: "PH_F1", 2: "PH_JAC", 3: "PH_JAC_REDO", 4: "PH_NEWTON", 5: "C_JAC_FIN", 6: "C_STEP_BEGIN",
         7: "C_STEP_TRY", 8: "C_NEWTON_BEGIN", 9: "C_NEWTON_FAIL", 10: "C_ERR_TEST", 11: "C_ACCEPT", 12: "PH_FBASE", 16: "RHS prologue",
         17: "post-RHS dispatch", 20: "newton: residual", 21: "newton: lu_solve", 22: "newton: norm+decide", 23: "lu_factor",
         24: "RHS cell model", 25: "RHS flux/hlift", 26: "RHS ET", 27: "RHS lateral flow", 28: "RHS top BC",
         29: "RHS assembly", 31: "loop top / outside"}
tot = sum(prof)
steps = N * 48.0
print(so, "rows", row0, "..", row0 + 47, "kernel_ms", round(out["kernel_ms"], 2), "col-days/s", round(N / (out["kernel_ms"] * 1e-3)),
      "cycles per column-step", round(tot / steps))
print(f"{'region':24s} {'share':>7s} {'cycles/step':>12s} {'entries/step':>13s} {'cycles/entry':>13s}")
for k, v in enumerate(prof):
    if v or cnt[k]:
        print(f"{names.get(k, str(k)):24s} {100.0 * v / tot:6.2f}% {v / steps:12.0f} {cnt[k] / steps:13.2f} "
              f"{(v / cnt[k] if cnt[k] else 0):13.0f}")
rhs = sum(prof[k] for k in (16, 24, 25, 26, 27, 28, 29))
SUBS = {33: "change_D order 1", 34: "change_D order 2", 35: "change_D order 3", 36: "change_D order 4", 37: "change_D order 5",
        41: "accept order 1", 42: "accept order 2", 43: "accept order 3", 44: "accept order 4", 45: "accept order 5",
        49: "predict order 1", 50: "predict order 2", 51: "predict order 3", 52: "predict order 4", 53: "predict order 5",
        56: "ET interior call", 57: "ET water_k > 0", 58: "ET renormalise", 59: "ET first midpoint", 60: "LF sink",
        61: "hydraulic lift", 62: "num_jac step sizes", 63: "num_jac group scatter"}
print("sub-region entries per column-step: " + ", ".join(f"{SUBS.get(32 + k, 32 + k)} {v / steps:.2f}" for k, v in enumerate(sub) if v))
print(f"RHS evaluations: {cnt[16] / steps:.2f} per column-step, {rhs / max(cnt[16], 1):.0f} cycles each, {100.0 * rhs / tot:.1f} % of the cycles")
if "--json" in sys.argv:
    path = sys.argv[sys.argv.index("--json") + 1]
    json.dump({"source": f"{os.path.basename(so)}, {N} members x D={DEPTH}, rows {row0}..{row0 + 47} (tools/prof_phases.py)",
               "cycles_per_column_step": {str(k): prof[k] / steps for k in range(32) if prof[k] or cnt[k]},
               "entries_per_column_step": dict({str(k): cnt[k] / steps for k in range(32) if prof[k] or cnt[k]},
                                               **{str(32 + k): sub[k] / steps for k in range(32) if sub[k]})}, open(path, "w"))
st.close()
