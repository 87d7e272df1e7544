"""Diagnostic build: cost of one RHS evaluation per wave at different occupancies (LDS request per workgroup decides
how many workgroups share a CU).  python tools/prof_rhs_occ.py <lib.so> <D> [N] [reps]"""
import os, sys, pathlib, time, subprocess
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
if len(sys.argv) > 5:           # child: one measurement
    from hydromodel_amd import _lib
    _lib.LIB_PATH = pathlib.Path(sys.argv[1]).resolve()
    import numpy as np
    from hydromodel_amd.digest import ColumnTables, ForcingDigest
    from hydromodel_amd.ensemble import pressure_head
    from hydromodel_amd.stepper import EnsembleStepper
    from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
    D, N, reps, kb = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    params = default_parameters()
    cols = ColumnTables(params, synthetic_well(D))
    forcing = ForcingDigest(params, synthetic_forcing_frame(1), cols)
    y0, _ = pressure_head(cols, cols.por_raw)
    st = EnsembleStepper(cols, forcing, N)
    st.set_state(y0); st.set_noise_host(np.random.default_rng(0).standard_normal((N, D)))
    if kb:
        os.environ["HYDROCOL_RHS_LDS_KB"] = str(kb)
    for row in (2, 24):
        os.environ["HYDROCOL_RHS_REPEAT"] = "1"; st.rhs(row)
        t0 = time.perf_counter(); st.rhs(row); t1 = time.perf_counter()
        os.environ["HYDROCOL_RHS_REPEAT"] = str(reps)
        t2 = time.perf_counter(); st.rhs(row); t3 = time.perf_counter()
        dt = (t3 - t2) - (t1 - t0)
        print(f"D={D} lds_kb={kb or 'step'} row {row}: {dt / (reps - 1) / N * 1e9 * 1024:.1f} ns per evaluation per SIMD "
              f"(= {dt / (reps - 1) / (N / 1024) * 2.4e9:.0f} cycles at 2.4 GHz)")
    st.close()
else:
    D = int(sys.argv[2]); N = int(sys.argv[3]) if len(sys.argv) > 3 else 8192; reps = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
    for kb in (0, 80, 52, 40):
        subprocess.run([sys.executable, __file__, sys.argv[1], str(D), str(N), str(reps), str(kb)], check=True)
