"""profiles/pmc_constants.json -- what bench.py reports as `roofline.traffic` and `valu_f64` -- must follow from the counter
CSVs committed next to it (the roofline reproducible from this round's profiles/), by the formulas of
MI355X_MICROARCH.md: traffic = (2 x FETCH_SIZE + WRITE_SIZE) KiB (gfx950 counts half of a streamed read), fp64 flop =
(ADD + MUL + TRANS + 2 FMA) wave instructions x 64 lanes."""
import csv
import json
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
PROF = REPO / "profiles"


def _mean_of_timed_launches(path):
    per = {}
    with open(path, newline="") as fh:
        for r in csv.DictReader(fh):
            assert "step_kernel" in r["Kernel_Name"]
            per.setdefault(int(r["Dispatch_Id"]), {}).setdefault(r["Counter_Name"], 0.0)
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    ids = sorted(per)
    timed = ids[1:] if len(ids) > 1 else ids            # the first launch is bench.py's warm-up day
    return {k: sum(per[i][k] for i in timed) / len(timed) for k in per[ids[0]]}, len(timed)


def test_counter_constants_follow_from_the_committed_csvs():
    table = json.loads((PROF / "pmc_constants.json").read_text())
    rec = table["kernels"]["300/special"]
    for name in rec["source"].split(", "):
        assert (REPO / name).exists(), name
    fetch, n_f = _mean_of_timed_launches(PROF / "r04_pmc_fetch_cpl5.csv")
    write, n_w = _mean_of_timed_launches(PROF / "r04_pmc_write_cpl5.csv")
    f64, _ = _mean_of_timed_launches(PROF / "r04_pmc_f64_cpl5.csv")
    assert n_f == n_w == 2                               # bench.py --steps 2 --warmup 1
    members = 262144
    hbm = (2.0 * fetch["FETCH_SIZE"] + write["WRITE_SIZE"]) * 1024.0 / members
    assert abs(hbm - rec["hbm_bytes_per_member_launch"]) < 1e-6 * hbm
    flop = (f64["SQ_INSTS_VALU_ADD_F64"] + f64["SQ_INSTS_VALU_MUL_F64"] + f64["SQ_INSTS_VALU_TRANS_F64"]
            + 2.0 * f64["SQ_INSTS_VALU_FMA_F64"]) * 64.0 / (members * 48.0)
    assert abs(flop - rec["f64_flop_per_column_step"]) < 1e-6 * flop
    # the calibration dispatch (a launch that only moves psi) shows the x2 of FETCH_SIZE on gfx950
    assert 0.45 < rec["calibration"]["fetch_size_over_state_bytes"] < 0.60
    assert len(table["kernel_hash"]) == 12


def test_constants_are_of_the_kernel_source_in_this_tree():
    """The committed counters belong to the device code this tree builds (hash of kernel sources + flags + compiler); bench.py
    would otherwise report traffic: null.  If this fails after a kernel edit: tools/gpu_r4_pmc.sh + tools/pmc_constants.py."""
    import __graft_entry__ as ge
    table = json.loads((PROF / "pmc_constants.json").read_text())
    assert table["kernel_hash"] == ge.kernel_hash(), "counter constants are of another kernel build: re-run the counter passes"
