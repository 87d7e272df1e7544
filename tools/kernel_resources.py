"""Register / scratch / occupancy picture of the step kernels, read off hipcc's own report on the CPU box.

    python tools/kernel_resources.py                       # every unit of the shipped build -> table on stdout
    python tools/kernel_resources.py --json out.json       # ... and as JSON (profiles/kernel_resources.json is this file)
    python tools/kernel_resources.py --cpl 5 --special 1 -- -DHC_WAVES_PER_BLOCK=8     # one unit with extra flags (A/B builds)

hipcc --cuda-device-only -Rpass-analysis=kernel-resource-usage prints, per kernel: SGPRs, VGPRs, AGPRs, scratch bytes per
lane, occupancy, SGPR / VGPR spills and LDS size.  tests/test_kernel_resources.py compares this table with the committed
one: a compiler that places a kernel's spills differently has to be looked at (tools/dev/partition_check.py on every
depth x build) before its build ships.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import __graft_entry__ as entry  # noqa: E402

FIELDS = {"TotalSGPRs": "sgpr", "VGPRs": "vgpr", "AGPRs": "agpr", "ScratchSize [bytes/lane]": "scratch",
          "Occupancy [waves/SIMD]": "occupancy", "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill",
          "LDS Size [bytes/block]": "lds"}


def parse_report(text):
    """{demangled-ish kernel name: {field: int}} from the -Rpass-analysis remarks"""
    out, cur = {}, None
    for line in text.splitlines():
        m = re.search(r"remark: .*?Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z][A-Za-z \[\]/]+?): (\d+)", line)
        if m and cur is not None and m.group(1).strip() in FIELDS:
            cur[FIELDS[m.group(1).strip()]] = int(m.group(2))
    return out


def short_name(mangled):
    """_ZN2hc11step_kernelILi5ELb1ELi4ELb0ELi1EEEvNS_8StepArgsE -> step<5,s,4,mon,1>"""
    m = re.search(r"step_kernelILi(\d+)ELb([01])ELi(\d+)ELb([01])ELi(\d+)ELb([01])E", mangled)
    if m:
        cpl, sp, wpb, pr, hv, pm = m.groups()
        return (f"step<{cpl},{'special' if sp == '1' else 'generic'},{wpb},{'predict' if pr == '1' else 'monitor'},{hv}"
                f"{',points' if pm == '1' else ''}>")
    return None


def unit_report(defs, extra=()):
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [entry._hipcc(), *entry.HIPCC_FLAGS, *defs, *extra, "--cuda-device-only",
               "-Rpass-analysis=kernel-resource-usage", "-c", "-o", os.path.join(tmp, "x.o"), str(entry.CSRC / "hc_inst.hip")]
        p = subprocess.run(cmd, cwd=str(entry.CSRC), capture_output=True, text=True)
        if p.returncode != 0:
            raise RuntimeError(p.stderr[-2000:])
        rep = parse_report(p.stderr)
    return {short_name(k): v for k, v in rep.items() if short_name(k)}


def shipped_units():
    units = []
    for n in entry.ALL_CPL:
        for sp in (1, 0):
            units.append(([f"-DHC_INST_CPL={n}", f"-DHC_INST_SPECIAL={sp}"] + entry.unit_flags_for((n, sp))))
    units.append(["-DHC_INST_PAIR"] + entry.unit_flags_for("pair"))
    return units


def table(jobs=None):
    units = shipped_units()
    with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as ex:
        parts = list(ex.map(unit_report, units))
    out = {}
    for p in parts:
        out.update(p)
    return dict(sorted(out.items()))


def show(tab):
    print(f"{'kernel':44s} vgpr agpr sgpr scratch occ sgpr_spill vgpr_spill    lds")
    for k, v in tab.items():
        print(f"{k:44s} {v.get('vgpr', -1):4d} {v.get('agpr', -1):4d} {v.get('sgpr', -1):4d} {v.get('scratch', -1):7d} "
              f"{v.get('occupancy', -1):3d} {v.get('sgpr_spill', -1):10d} {v.get('vgpr_spill', -1):10d} {v.get('lds', -1):6d}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--json")
    ap.add_argument("--cpl", type=int)
    ap.add_argument("--special", type=int, default=1)
    ap.add_argument("--pair", action="store_true")
    ap.add_argument("extra", nargs="*")
    a = ap.parse_args()
    if a.cpl or a.pair:
        defs = ["-DHC_INST_PAIR"] if a.pair else [f"-DHC_INST_CPL={a.cpl}", f"-DHC_INST_SPECIAL={a.special}"]
        tab = unit_report(defs, a.extra)
    else:
        tab = table()
    show(tab)
    if a.json:
        Path(a.json).write_text(json.dumps(tab, indent=1, sort_keys=True) + "\n")
