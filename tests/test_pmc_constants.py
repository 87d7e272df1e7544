"""profiles/pmc_constants.json -- what bench.py reports as `roofline.traffic` / `roofline.fabric` and `valu_f64` -- must follow
from the counter CSVs committed next to it (the roofline reproducible from this round's profiles/): fabric bytes =
(fetch_factor x FETCH_SIZE + write_factor x WRITE_SIZE) KiB, the factors measured on known byte counts in the kernel's own
access shapes (tools/pmc_calib.hip -> profiles/r05_pmc_calib.txt; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts half of a
streamed read); fp64 flop = (ADD + MUL + TRANS + 2 FMA) wave instructions x 64 lanes.  Every kernel a BASELINE config or a
reference well uses has its passes."""
import csv
import json
import re
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
PROF = REPO / "profiles"
KERNELS = ("300/special", "200/special", "300/generic", "401/special", "581/special")


def _one_launch(path):
    vals, ids = {}, set()
    with open(path, newline="") as fh:
        for r in csv.DictReader(fh):
            assert "step_kernel" in r["Kernel_Name"]
            ids.add(int(r["Dispatch_Id"]))
            vals[r["Counter_Name"]] = vals.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    assert len(ids) == 1, path                              # tools/prof_kernel.py: ONE 48-row launch per pass
    return vals


def test_counter_constants_follow_from_the_committed_csvs():
    table = json.loads((PROF / "pmc_constants.json").read_text())
    assert len(table["kernel_hash"]) == 12
    cal = table["calibration"]
    # raw_buffer_load_b64 / raw_buffer_store_b64 at 512 B per wave instruction: FETCH_SIZE reads half, WRITE_SIZE all of it
    assert 1.95 < cal["fetch_factor"] < 2.05 and 0.98 < cal["write_factor"] < 1.02
    assert abs(cal["fetch_size_per_byte_loaded"]["cal_psi_load"] - 0.5) < 0.02         # the state's strided layout: the same
    assert abs(cal["write_size_per_byte_stored"]["cal_psi_store"] - 1.0) < 0.02
    assert (PROF / "r05_pmc_calib.txt").exists()
    assert set(KERNELS) <= set(table["kernels"])
    for key in KERNELS:
        rec = table["kernels"][key]
        members = int(re.match(r"(\d+) members", rec["launch_shape"]).group(1))
        csvs = {Path(n).stem.rsplit("_", 1)[1]: REPO / n for n in rec["source"].split(", ")}
        for f in csvs.values():
            assert f.exists(), f
        fetch, write, f64 = _one_launch(csvs["fetch"]), _one_launch(csvs["write"]), _one_launch(csvs["f64"])
        fabric = (cal["fetch_factor"] * fetch["FETCH_SIZE"] + cal["write_factor"] * write["WRITE_SIZE"]) * 1024.0 / members
        assert abs(fabric - rec["fabric_bytes_per_member_launch"]) < 1e-6 * fabric, key
        assert abs(rec["fabric_bytes_per_column_step"] * 48 - fabric) < 1e-6 * fabric
        flop = (f64["SQ_INSTS_VALU_ADD_F64"] + f64["SQ_INSTS_VALU_MUL_F64"] + f64["SQ_INSTS_VALU_TRANS_F64"]
                + 2.0 * f64["SQ_INSTS_VALU_FMA_F64"]) * 64.0 / (members * 48.0)
        assert abs(flop - rec["f64_flop_per_column_step"]) < 1e-6 * flop, key
        assert 0.0 < rec["l2_hit_rate"] < 1.0


def test_constants_are_of_the_kernel_source_in_this_tree():
    """The committed counters belong to the device code this tree builds (hash of kernel sources + flags + compiler); bench.py
    would otherwise report traffic: null.  If this fails after a kernel edit: tools/gpu_r5_pmc.sh + tools/pmc_constants.py."""
    import __graft_entry__ as ge
    table = json.loads((PROF / "pmc_constants.json").read_text())
    assert table["kernel_hash"] == ge.kernel_hash(), "counter constants are of another kernel build: re-run the counter passes"
