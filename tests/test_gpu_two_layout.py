"""The TWO layout (two waves per SIMD with hand-placed state) changes WHERE the integrator's state lives, not one
arithmetic operation: a build of the same source with the layout switched off (`-DHC_TWO_MASK=0`: the round-3 register
layout, one wave per SIMD) must end in the same states to the bit.  Both are compiled with `-ffp-contract=on` (the unit's
own flag), which is what makes the bits a property of the source rather than of the optimiser's view of it.

VERDICT r3 item 1 "done means": A/B with identical state digests -- kept as a test, not only as a development record
(`profiles/r04_two_layout_ab.txt`)."""
import os
import re
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent


def _digests(lib, depths, members):
    env = dict(os.environ, HC_PROF_MEMBERS=str(members))
    p = subprocess.run([sys.executable, str(REPO / "tools" / "prof_depth.py"), str(lib), *map(str, depths)], cwd=REPO, env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    out = dict((int(d), s) for d, s in re.findall(r"D=(\d+): .*? sha ([0-9a-f]{10})", p.stdout))
    assert sorted(out) == sorted(depths), p.stdout
    return out


def test_one_wave_and_two_wave_builds_of_the_same_source_agree_to_the_bit(tmp_path):
    import __graft_entry__ as ge
    ge.build()
    # the kernels of 5 cells per lane (D = 257..320: the bench's depth) and of 7 (D = 385..448: the reference's default well; on
    # the TWO layout since late round 5) with the layout switched off
    one_wave = ge.build_library(tmp_path / "lib_one_wave.so", cpls=(5, 7), obj_dir=tmp_path / "obj", force=True,
                                defines=("-DHC_CPL_MASK=160", "-DHC_TWO_MASK=0", "-DHC_TWO_MASK_GENERIC=0"))
    depths, members = [261, 300, 320, 401], 3000     # 3 000 members: the 1-ulp difference of `fast` builds showed in ~1 member of 1 000
    a = _digests(ge.CSRC / "libhydrocol.so", depths, members)
    b = _digests(one_wave, depths, members)
    assert a == b, (a, b)


def test_the_compile_settings_of_late_round_5_do_not_change_a_bit(tmp_path):
    """`__graft_entry__.UNIT_FLAGS` compiles the two-wave units with machine LICM off, MachineSink's sink-to-avoid-spills and
    (some) the AMDGPU register-pressure trackers: options that act after instruction selection, i.e. on placement and schedule
    only (profiles/r05_compile_flags_by_unit.txt: identical digests in every A/B block).  Kept as a test: a build of the same
    units with NONE of the tuning options -- only `-ffp-contract=on`, which is part of the semantics -- must end in the same
    states to the bit at the bench's depth, the default well's and on the split column."""
    import __graft_entry__ as ge
    ge.build()
    plain = ge.build_library(tmp_path / "lib_plain.so", cpls=(5, 7), obj_dir=tmp_path / "obj", force=True, unit_flags=False,
                             defines=("-DHC_CPL_MASK=160", "-ffp-contract=on"))
    depths, members = [300, 401, 581], 3000
    a = _digests(ge.CSRC / "libhydrocol.so", depths, members)
    b = _digests(plain, depths, members)
    assert a == b, (a, b)
