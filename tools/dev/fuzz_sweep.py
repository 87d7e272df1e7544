"""Random parameter sweeps, one handle against stand-alone handles: for each case a random depth (all cells-per-lane
classes incl. the two-waves-per-SIMD ones), 2 .. 7 random points (exponents, a0, psi_sat), a random member count per point
(1 .. 130, so that chunks are ragged and some waves of a workgroup find no member), random launch length and chunk size,
round-robin point ids as a rank of a multi-GPU sweep would get them; a third of the cases on a starved iteration budget
(abandoned attempts, the x0.8 retry rule).  Property: every point's states and moments are
bit-equal to a handle that runs that point alone with the same global member ids; per-point counts are complete.
    python tools/dev/fuzz_sweep.py [n_cases=30] [seed=1]
Exit status 1 on the first difference."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
from hydromodel_amd.digest import ColumnTables, ForcingDigest
from hydromodel_amd.ensemble import SweepSimulation, check_sweep_points
from hydromodel_amd.synthetic import default_parameters, synthetic_forcing_frame, synthetic_well
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
# (round 4: sweeps and lone points take the same kernel at every depth -- the split column from 577 nodes on for both --
#  so no kernel is forced here any more; FUZZ_ONE_WAVE=1 restores the one-wave kernels everywhere)
if os.environ.get("FUZZ_ONE_WAVE"):
    os.environ["HYDROCOL_SPLIT_COLUMN"] = "0"
frame = synthetic_forcing_frame(1)
bad = 0
t0 = time.time()
for case in range(n_cases):
    D = int(rng.choice([101, 121, 128, 150, 192, 193, 200, 241, 300, 321, 401, 470, 541, 581, 640]))
    P = int(rng.integers(2, 8))
    M = int(rng.choice([1, 2, 3, 7, 8, 9, 16, 17, 40, 130]))
    rows = int(rng.choice([5, 24, 50]))
    first_row = int(rng.choice([1, 20, 40]))
    params = default_parameters()
    pts = []
    for _ in range(P):
        sp = {}
        if rng.random() < 0.8:
            sp["n"] = float(rng.choice([1.5, 1.7, 2.0, 2.3, 2.8]))
        if rng.random() < 0.6:
            sp["a0"] = float(rng.choice([0.004, 0.009, 0.015, 0.025]))
        if rng.random() < 0.4:
            sp["psi_sat"] = float(rng.choice([-0.001, -0.0047, -0.05, -0.5]))
        pts.append({"Soil_Properties": sp} if sp else {})
    cols_all = [ColumnTables(mp, synthetic_well(D)) for mp in check_sweep_points(params, pts)]
    forcing = ForcingDigest(params, frame, cols_all[0])
    # a cheap common start (no spin-up: the property is about the launch, not the physics): a hydrostatic-like profile
    table = 300.0
    psi0 = np.stack([c.z - table for c in cols_all]) + 0.3 * rng.standard_normal((P, D))
    # a third of the cases run on a starved iteration budget: attempts are abandoned, the x0.8 retry rule and the
    # failure accounting run inside the multi-point launch too
    budget = int(rng.choice([0, 0, 30, 60]))
    ids = np.sort(rng.choice(64, size=P, replace=False))         # this "rank"'s share of a 64-point sweep
    rpl = int(rng.choice([0, 7, 24]))
    chunk = int(rng.choice([0, 0, 4, 8, 13]))
    if chunk:
        os.environ["HYDROCOL_CHUNK_MEMBERS"] = str(chunk)
    else:
        os.environ.pop("HYDROCOL_CHUNK_MEMBERS", None)
    big = SweepSimulation(cols_all, forcing, M, seed=seed + case, psi0=psi0, point_ids=ids)
    if rpl:
        big.stepper.set_rows_per_launch(rpl)
    if budget:
        big.stepper.set_iteration_budget(budget)
    if first_row > 1:
        big.next_row = first_row
    big.advance(rows)
    state = big.stepper.get_state().reshape(P, M, D)
    mom = big.moments()
    cnt = big.stepper.counters()
    big.close()
    os.environ.pop("HYDROCOL_CHUNK_MEMBERS", None)
    ok = bool(np.array_equal(mom[:, 0, first_row:first_row + rows], np.full((P, rows), M)))
    worst = ""
    for j in range(P):
        one = SweepSimulation([cols_all[j]], forcing, M, seed=seed + case, psi0=psi0[j], point_ids=[int(ids[j])])
        if first_row > 1:
            one.next_row = first_row
        if budget:
            one.stepper.set_iteration_budget(budget)
        one.advance(rows)
        s1 = one.stepper.get_state().reshape(M, D)
        m1 = one.moments()[0]
        one.close()
        if not (np.array_equal(s1, state[j]) and np.array_equal(m1, mom[j])):
            ok = False
            worst += f" point {j} (id {int(ids[j])}): {int((s1 != state[j]).any(axis=1).sum())} of {M} members differ;"
    bad += not ok
    print(f"case {case}: D={D} P={P} M={M} rows {first_row}+{rows} rows/launch {rpl or 'auto'} chunk {chunk or 'auto'} budget {budget or 'default'} ids {ids.tolist()} "
          f"failed attempts {cnt['failed_attempts']} guard trips {cnt['guard_trips']}: {'ok' if ok else 'DIFFERENT' + worst}", flush=True)
    if not ok:
        break
print(f"{case + 1} cases, {bad} different, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
