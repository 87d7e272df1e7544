"""Noise vector seen by the per-row diagnostics call of ``Simulation.run`` (simulation.py:126).

``h_model(y_i, z, args_i)`` (simulation.py:623) reads ``args_i["n_rnd"]`` as the solve left it: every failed BDF attempt
scaled it by 0.8 in place (richards_pde.py:522) -- the shared base array on ordinary rows, the row's own vector on
refresh rows.  ``simulation.rows_noise`` rebuilds that per row from the kernel's failed-attempt counts.
"""
import numpy as np

from helpers import golden
from hydromodel_amd.simulation import rows_noise


def test_damping_rule_reproduces_the_reference_vectors_bit_for_bit():
    """G5: on every row the reference needed several attempts for, noise-after = noise-before * 0.8 * 0.8 ..."""
    g = golden("g5_traj_1.npz")
    stats, rows = g["per_row_stats"], g["rec_rows"]
    seen = 0
    for k, i in enumerate(rows):
        if stats[i, 4] <= 1:
            assert np.array_equal(g["rec_nrnd_out"][k], g["rec_nrnd_in"][k])
            continue
        nin, nout = g["rec_nrnd_in"][k], g["rec_nrnd_out"][k]
        failed = int(round(np.log(np.median(nout / nin)) / np.log(0.8)))
        assert failed in (stats[i, 4] - 1, stats[i, 4])          # the last attempt may have failed as well
        for is_fresh in (0, 1):                                  # the rule is the same for a base and a fresh vector
            base = nin.copy()
            got = rows_noise(base, [nin.copy()], [is_fresh], [failed])
            assert np.array_equal(got[0], nout), (i, failed, is_fresh)
            assert np.array_equal(base, nin if is_fresh else nout)    # only a base vector is damped in place
        seen += 1
    assert seen >= 10


def test_base_damping_carries_forward_and_fresh_vectors_live_one_row():
    rng = np.random.default_rng(0)
    D = 7
    base0 = rng.standard_normal(D)
    fresh = rng.standard_normal((2, D))
    refresh = [0, 1, 0, 0, 1, 0]
    failed = [0, 2, 1, 0, 0, 3]
    base = base0.copy()
    out = rows_noise(base, fresh, refresh, failed)
    assert np.array_equal(out[0], base0)                         # rows before a retried row see the undamped base
    assert np.array_equal(out[1], fresh[0] * 0.8 * 0.8)          # a refresh row damps its own vector ...
    assert np.array_equal(out[2], base0 * 0.8)                   # ... not the base; this row's failure does
    assert np.array_equal(out[3], base0 * 0.8)
    assert np.array_equal(out[4], fresh[1])
    assert np.array_equal(out[5], base0 * 0.8 * 0.8 * 0.8 * 0.8)
    assert np.array_equal(base, out[5])                          # the caller's base array ends as the library's does
