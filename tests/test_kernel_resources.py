"""Compiler-drift gate: what hipcc makes of every step kernel -- registers, scratch,
spills, occupancy -- is compared with the committed table profiles/kernel_resources.json on the CPU box.

Round 2 shipped two builds that passed host-noise parity and were wrong with in-kernel noise; the cause was narrowed to the
way the compiler placed a kernel's spills, never found.  Nothing on the GPU side notices when a new hipcc (or an edit)
moves a kernel to another spill picture -- this test does: a difference fails with what to re-run."""
import json
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
KEYS = ("vgpr", "agpr", "scratch", "occupancy", "sgpr_spill", "vgpr_spill")


def test_every_step_kernel_compiles_to_the_committed_resource_picture():
    import sys
    sys.path.insert(0, str(REPO / "tools"))
    import kernel_resources
    want = json.loads((REPO / "profiles" / "kernel_resources.json").read_text())
    try:
        have = kernel_resources.table()
    except (OSError, RuntimeError) as e:
        pytest.skip(f"hipcc cannot report kernel resources here: {e}")
    assert sorted(have) == sorted(want), "the set of step-kernel instantiations changed"
    diffs = []
    for name in sorted(want):
        for k in KEYS:
            if have[name].get(k) != want[name].get(k):
                diffs.append(f"{name}: {k} {want[name].get(k)} -> {have[name].get(k)}")
    assert not diffs, ("hipcc now compiles these kernels to another register / spill picture:\n  " + "\n  ".join(diffs) +
                       "\nre-run tools/dev/partition_check.py on every depth x build and the GPU suite, then refresh the table "
                       "with `python tools/kernel_resources.py --json profiles/kernel_resources.json`")
    # the layouts the host code sizes LDS and the global region for: two waves per SIMD exactly where hc_step.h says so
    for name, v in have.items():
        cpl, model, wpb = name[5:-1].split(",")[:3]
        # two waves per SIMD: 2 ... 6 cells per lane (round 5: the generic-exponent kernel of 6 cells too; late round 5: 7 cells with
        # the default exponents) and the split column (",2>" / ",2,points>": 5 cells per lane and half, four pairs per CU since round 5)
        two = int(cpl) <= 6 or (int(cpl) == 7 and model == "special")
        assert v["occupancy"] == (2 if two else 1), name
        assert int(wpb) == (8 if two else 4), name
