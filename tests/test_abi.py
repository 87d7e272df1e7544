"""The C-ABI library loads and exports every symbol include/hydrocol.h declares (no GPU needed)."""
import ctypes as C
import re
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent


def _declared():
    text = (REPO / "include" / "hydrocol.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hc_[a-z_0-9]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from hydromodel_amd import _lib
    return _lib.load()


def test_header_and_binding_agree(lib):
    from hydromodel_amd import _lib
    declared = _declared()
    assert len(declared) >= 20
    assert sorted(_lib.EXPORTS) == declared


def test_every_declared_symbol_is_exported(lib):
    for name in _declared():
        assert hasattr(lib, name), name


def test_struct_sizes():
    from hydromodel_amd import _lib
    assert C.sizeof(_lib.ColumnParams) == 8 * 4 + 15 * 8 + 2 * 4          # static_assert'ed on the C side
    assert C.sizeof(_lib.StepArgs) == 8 + 8 + 4 + 4 + 5 * 8 + 8 + 8
    assert C.sizeof(_lib.SpinupArgs) == 8 + 4 + 4 + 8 + 8 + 8 + 8          # int32 + padding before the doubles


def test_version_string(lib):
    assert b"gfx950" in lib.hc_version()


def test_fails_loudly_without_a_device(lib):
    """No CPU path: on a box without a GPU hc_create must fail with a message, never fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible; the no-device path cannot be exercised here")
    from hydromodel_amd import _lib
    h = C.c_void_p()
    rc = lib.hc_create(0, C.byref(h))
    assert rc in (-2, -3) and not h.value
    assert b"no CPU path" in lib.hc_last_error() or b"failed" in lib.hc_last_error()
    with pytest.raises(_lib.HcError):
        _lib.check(rc)


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    from hydromodel_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(_lib.HcError, match="no CPU fallback"):
        _lib.load()


def test_product_never_imports_the_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline may touch oracle/ (it is the checker)."""
    pat = re.compile(r"import\s+oracle|from\s+oracle|oracle[/.]|hydro_oracle|ho_[a-z_]+\(")
    for path in (REPO / "hydromodel_amd").rglob("*"):
        if path.suffix in (".py", ".hip", ".h", ".cpp"):
            assert not pat.search(path.read_text()), path


def test_every_kernel_unit_is_built_and_the_scheduler_table_names_real_units():
    """One translation unit per cells-per-lane count (2..10) and cell model, one for the split column; the per-unit
    settings (`UNIT_FLAGS`) may only name units that exist, as groups of command-line words: `-mllvm <option>` pairs
    (tuning: droppable), `-ffp-contract=on` (the TWO-layout units: part of the semantics, never dropped) or the cell
    model's batch size of one unit (same bits under `-ffp-contract=on`)."""
    import __graft_entry__ as ge
    assert tuple(ge.ALL_CPL) == tuple(range(2, 11))
    units = {(n, sp) for n in ge.ALL_CPL for sp in (0, 1)} | {"pair"}
    assert set(ge.UNIT_FLAGS) <= units
    for groups in ge.UNIT_FLAGS.values():
        for group in groups:
            assert group in (("-ffp-contract=on",), ("-DHC_MODEL_BATCH=4",), ("-DHC_GENERIC_BATCH=4",)) or \
                (len(group) == 2 and group[0] == "-mllvm" and group[1].startswith("-amdgpu-")) or \
                group == ("-mllvm", "-disable-machine-licm", "-mllvm", "-sink-insts-to-avoid-spills")   # one group: alone, the first loses 16 %
    # the units compiled as TWO-layout kernels are exactly the ones with source-determined contraction
    two = {k for k, groups in ge.UNIT_FLAGS.items() if ("-ffp-contract=on",) in groups}
    assert two == {(4, 1), (5, 1), (6, 1), (7, 1), (4, 0), (5, 0), (6, 0), "pair"}    # (round 5: + generic 6 cells, the split column, special 7 cells)
    src = (ge.CSRC / "hc_inst.hip").read_text()
    assert "HC_INST_SPECIAL" in src and "HC_INST_PAIR" in src


def test_a_hipcc_without_a_tuning_option_still_builds_the_unit(tmp_path, monkeypatch):
    """VERDICT r3 item 4b / ADVICE r3: the per-unit `-mllvm -amdgpu-...` options are LLVM internals.  A hipcc that rejects
    one must cost speed, not the library: the probe drops it (and says so), the unit compiles with the defaults."""
    import __graft_entry__ as ge
    monkeypatch.setenv("HYDROCOL_REJECT_MLLVM", "-amdgpu-use-amdgpu-trackers,-sink-insts-to-avoid-spills")
    monkeypatch.setattr(ge, "_FLAG_OK", {})
    assert ge.unit_flags_for((9, 0)) == [] and ge.unit_tuning_flags((9, 0)) == []
    assert ge.unit_flags_for((8, 0)) == ["-mllvm", "-amdgpu-sched-strategy=iterative-maxocc"]
    assert ge.unit_flags_for((8, 1)) == []                          # (machine LICM off goes with the sink option or not at all)
    assert ge.unit_flags_for((3, 1)) == ["-mllvm", "-amdgpu-sched-strategy=iterative-minreg"]       # others untouched
    assert ge.unit_flags_for((5, 1)) == ["-ffp-contract=on"]                                        # never dropped
    assert ge.unit_flags_for((6, 1)) == ["-ffp-contract=on", "-DHC_MODEL_BATCH=4"]
    # and a rejection that only shows at the compile itself: compile_one retries without the tuning words
    monkeypatch.setattr(ge, "_FLAG_OK", {("-mllvm", "-amdgpu-sched-strategy=iterative-minreg"): True})
    monkeypatch.setitem(ge.UNIT_FLAGS, (2, 1), [("-mllvm", "-amdgpu-sched-strategy=iterative-minreg"), ("-mllvm", "-not-an-option")])
    monkeypatch.setattr(ge, "flag_supported", lambda group: True)
    lib = ge.build_library(tmp_path / "lib_fallback.so", cpls=(2,), defines=("-DHC_CPL_MASK=4",), obj_dir=tmp_path / "obj", force=True)
    assert lib.exists() and lib.stat().st_size > 100_000
    # the library's device-code identity says how its units were REALLY built: the unit that fell back is hashed
    # without its tuning words, the stamp next to the host object holds that identity, and hc_version() carries it
    assert (tmp_path / "obj" / "hc_inst_cpl2s.fellback").exists()
    stamp = (tmp_path / "obj" / "hydrocol.kernel_hash").read_text().strip()
    assert stamp == ge.kernel_hash(("-DHC_CPL_MASK=4",), fell_back=((2, 1),)) != ge.kernel_hash(("-DHC_CPL_MASK=4",))
    assert stamp.encode() in lib.read_bytes()
