"""dev: does the CPU oracle ever fall into the same chattering on tiny perturbations of the stuck row's input?"""
import os, sys, time, multiprocessing as mp
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np

def trial(seed):
    from hydromodel_amd.digest import ColumnTables, ForcingDigest
    from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
    from oracle.oracle import Oracle
    g = np.load(os.path.join(R, "tools", "dev", "guard_case2.npz"))
    row = int(g["row"])
    params = default_parameters()
    cols = ColumnTables(params, synthetic_well(300))
    forcing = ForcingDigest(params, synthetic_forcing_frame(1), cols)
    o = Oracle(cols, forcing.surface_evap)
    r = Oracle.row(forcing.precip[row], forcing.atm[row], forcing.daylight[row], forcing.wtd_obs[row])
    rng = np.random.default_rng(seed)
    y = g["y_before"] * (1.0 + 1e-13 * rng.standard_normal(300)) if seed else g["y_before"]
    t0 = time.perf_counter()
    _, so, _, _ = o.solve_row(r, row - 1, row, y, g["z"].copy())
    return seed, so["nfev"], so["attempts"], time.perf_counter() - t0

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    res, hung = [], 0
    with mp.Pool(8) as pool:
        jobs = [pool.apply_async(trial, (s,)) for s in range(n)]
        deadline = time.time() + float(sys.argv[2]) if len(sys.argv) > 2 else time.time() + 120.0
        for j in jobs:
            try:
                res.append(j.get(timeout=max(0.1, deadline - time.time())))
            except mp.TimeoutError:
                hung += 1
        pool.terminate()
    nf = np.array([r[1] for r in res]); at = np.array([r[2] for r in res])
    print(f"{len(res)} finished, {hung} did not finish before the deadline; nfev median {np.median(nf):.0f} max {nf.max()}, "
          f"attempts > 1: {(at > 1).sum()}, nfev > 5000: {(nf > 5000).sum()}")
