"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (a committed target, not a
one-off).  `make -C oracle sanitize` builds oracle/hydro_oracle.c with -fsanitize=address,undefined and runs the oracle's
own test files against that library in a child process that preloads the sanitizer runtime; any report aborts the child."""
import shutil
import subprocess
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent


def test_oracle_is_clean_under_asan_and_ubsan():
    if not shutil.which("gcc") or not shutil.which("make"):
        pytest.skip("no gcc / make")
    r = subprocess.run(["make", "-C", str(REPO / "oracle"), "sanitize"], capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
