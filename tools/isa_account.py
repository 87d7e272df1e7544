"""Static instruction mix of the step kernel BY REGION, read off the assembly of a -DHC_MARKS build, and -- joined with
the region entry counts and cycles of a -DHC_PROFILE run (tools/prof_phases.py --json) -- an estimate of the DYNAMIC
instruction account per column-step.  rocprofv3's thread trace (--att) would give this directly, but its decoder
library (librocprof-trace-decoder) is not part of this image, so a trace cannot be decoded here.

    python tools/isa_account.py [--cpl 5] [--kernel step_kernelILi5ELb1ELi4ELb0] [--dynamic gpurun_out/phases.json]

Every HC_STAMP / HC_RSTAMP site of the kernel emits "; HCMARK <region>" in a marks build (no instruction).  A region's
static code = the instructions that follow its mark up to the next mark in layout order; each instruction is put in
one class.  Inside a region there are branches (switch(order), rare paths), so static counts are an upper bound of one
pass; the dynamic estimate scales the per-region instruction counts so that their total matches the measured
SQ_INSTS_* totals where those are given.
"""
import argparse
import json
import os
import re
import subprocess
import sys
from collections import Counter, defaultdict

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(R, "hydromodel_amd", "csrc")

REGION_NAMES = {0: "PH_F0", 1: "PH_F1", 2: "PH_JAC", 3: "PH_JAC_REDO", 4: "PH_NEWTON", 5: "C_JAC_FIN", 6: "C_STEP_BEGIN",
                7: "C_STEP_TRY", 8: "C_NEWTON_BEGIN", 9: "C_NEWTON_FAIL", 10: "C_ERR_TEST", 11: "C_ACCEPT", 12: "PH_FBASE",
                16: "RHS prologue", 17: "post-RHS dispatch", 20: "newton: residual", 21: "newton: lu_solve",
                22: "newton: norm+decide", 23: "lu_factor", 24: "RHS cell model", 25: "RHS flux/hlift", 26: "RHS ET",
                27: "RHS lateral flow", 28: "RHS top BC", 29: "RHS assembly", 31: "loop top", -1: "outside the phase loop",
                33: " change_D order 1", 34: " change_D order 2", 35: " change_D order 3", 36: " change_D order 4",
                37: " change_D order 5", 41: " accept order 1", 42: " accept order 2", 43: " accept order 3",
                44: " accept order 4", 45: " accept order 5", 49: " predict order 1", 50: " predict order 2",
                51: " predict order 3", 52: " predict order 4", 53: " predict order 5", 56: " ET interior call",
                57: " ET water_k > 0", 58: " ET renormalise", 59: " ET first midpoint", 60: " LF sink",
                61: " hydraulic lift", 62: " num_jac step sizes", 63: " num_jac group scatter"}


def classify(op):
    if op.startswith(("v_fma_f64", "v_fmac_f64", "v_add_f64", "v_mul_f64")):
        return "f64 arith"
    if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")):
        return "f64 trans"
    if op.startswith(("v_max_f64", "v_min_f64", "v_ldexp_f64", "v_frexp", "v_rndne_f64", "v_cvt_", "v_div_", "v_trunc_f64",
                      "v_floor_f64", "v_fract_f64", "v_ceil_f64")):
        return "f64 other (min/max/ldexp/frexp/cvt)"
    if op.startswith("v_cmp") or op.startswith("v_cmpx"):
        return "v_cmp"
    if op.startswith("v_cndmask"):
        return "v_cndmask"
    if op.startswith("v_accvgpr"):
        return "v_accvgpr (AGPR<->VGPR)"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "v_read/writelane"
    if op.endswith("_dpp") or "_dpp" in op or op.startswith(("v_permlane",)):
        return "DPP moves"
    if op.startswith(("v_mov", "v_swap")):
        return "v_mov"
    if op.startswith("v_"):
        return "VALU int/other"
    if op.startswith("ds_bpermute") or op.startswith("ds_permute"):
        return "LDS bpermute"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith("scratch_load"):
        return "scratch load (spill)"
    if op.startswith("scratch_store"):
        return "scratch store (spill)"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "VMEM"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_call")):
        return "branch"
    if op.startswith(("s_load", "s_buffer_load", "s_store")):
        return "SMEM"
    if op.startswith("s_"):
        return "SALU"
    return "other"


VALU_CLASSES = ("f64 arith", "f64 trans", "f64 other (min/max/ldexp/frexp/cvt)", "v_cmp", "v_cndmask",
                "v_accvgpr (AGPR<->VGPR)", "v_read/writelane", "DPP moves", "v_mov", "VALU int/other")


def assemble(cpl, extra):
    out = f"/tmp/hc_marks_cpl{cpl}.s"
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-DHC_INST_CPL={cpl}", "-DHC_INST_SPECIAL=1", "-DHC_MARKS",
           *extra, "--cuda-device-only", "-S", "-o", out, "hc_inst.hip"]
    subprocess.run(cmd, check=True, cwd=CSRC, stderr=subprocess.DEVNULL)
    return out


def regions_of(path, kernel):
    text = open(path).read().splitlines()
    start = next(i for i, l in enumerate(text) if l.startswith("_ZN2hc11" + kernel) and l.rstrip().endswith(
        ("E:", "E")) or (l.startswith("_ZN2hc11" + kernel) and ":" in l))
    end = next(i for i in range(start, len(text)) if text[i].strip().startswith("s_endpgm"))
    region = parent = -1
    per = defaultdict(Counter)
    sites = Counter()                        # how often a sub-region's code was inlined (one mark per copy)
    for line in text[start + 1:end + 1]:
        m = re.search(r";\s*HCMARK\s+(-?\d+)", line)
        if m and not line.strip().startswith("v_") and not line.strip().startswith("s_"):
            k = int(m.group(1))
            if k == -2:
                region = parent              # end of a sub-region: back to the enclosing region
            elif k >= 32:
                region = k                   # sub-region: entries counted on their own
                sites[k] += 1
            else:
                region = parent = k
            continue
        t = line.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        if not re.match(r"^[a-z_0-9]+$", op):
            continue
        per[region][classify(op)] += 1
    for k, n in sites.items():               # per-entry cost of a sub-region = its copies' average
        if n > 1:
            for c in per[k]:
                per[k][c] = per[k][c] / n
    return per


def phases_text(path):
    """Entries and cycles per region out of the table tools/prof_phases.py prints (the committed profiles/r0N_phases_*.txt)."""
    by_name = {v.strip(): k for k, v in REGION_NAMES.items()}
    by_name["loop top / outside"] = 31
    entries, cycles = {}, {}
    lines = open(path).read().splitlines()
    for line in lines:
        m = re.match(r"^(.+?)\s+([0-9.]+)%\s+(\d+)\s+([0-9.]+)\s+(\d+)\s*$", line)
        if m and m.group(1).strip() in by_name:
            k = by_name[m.group(1).strip()]
            entries[str(k)], cycles[str(k)] = float(m.group(4)), float(m.group(3))
        if line.startswith("sub-region entries per column-step:"):
            for part in line.split(":", 1)[1].split(","):
                name, val = part.strip().rsplit(" ", 1)
                if name in by_name:
                    entries[str(by_name[name])] = float(val)
    return {"entries_per_column_step": entries, "cycles_per_column_step": cycles, "source": lines[0].strip() + " -- " + path}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cpl", type=int, default=5)
    ap.add_argument("--kernel", default=None, help="mangled-name prefix after _ZN2hc11 (default: the special monitoring kernel)")
    ap.add_argument("--dynamic", default=None, help="JSON of tools/prof_phases.py --json (entries and cycles per region)")
    ap.add_argument("--phases", default=None, help="text output of tools/prof_phases.py (profiles/r05_phases_d300.txt) instead of --dynamic")
    ap.add_argument("--asm", default=None, help="use this .s file instead of compiling")
    ap.add_argument("-D", action="append", default=[], help="extra -D for the compile")
    args = ap.parse_args()
    kernel = args.kernel or f"step_kernelILi{args.cpl}ELb1ELi4ELb0ELi1EE"
    path = args.asm or assemble(args.cpl, [f"-D{d}" for d in args.D])
    per = regions_of(path, kernel)
    classes = sorted({c for r in per.values() for c in r})
    total = Counter()
    for r in per.values():
        total.update(r)
    print(f"# static instruction mix of {kernel} by region ({path})")
    print(f"{'region':28s} {'all':>6s} {'VALU':>6s} {'f64':>6s} | " + " ".join(f"{c[:14]:>14s}" for c in classes))
    for reg in sorted(per):
        r = per[reg]
        valu = sum(r[c] for c in VALU_CLASSES)
        f64 = r["f64 arith"] + r["f64 trans"]
        print(f"{REGION_NAMES.get(reg, str(reg)):28s} {sum(r.values()):6.0f} {valu:6.0f} {f64:6.0f} | "
              + " ".join(f"{r[c]:14.0f}" for c in classes))
    valu = sum(total[c] for c in VALU_CLASSES)
    print(f"{'TOTAL':28s} {sum(total.values()):6.0f} {valu:6.0f} {total['f64 arith'] + total['f64 trans']:6.0f} | "
          + " ".join(f"{total[c]:14.0f}" for c in classes))
    if args.phases:
        dyn = phases_text(args.phases)
    elif args.dynamic:
        dyn = json.load(open(args.dynamic))
    else:
        return
    entries = {int(k): v for k, v in dyn["entries_per_column_step"].items()}
    cycles = {int(k): v for k, v in dyn["cycles_per_column_step"].items()}
    print(f"\n# dynamic estimate per column-step = static instructions of a region x its measured entries "
          f"({dyn.get('source', args.dynamic)})")
    print(f"{'region':28s} {'entries':>8s} {'instr':>8s} {'f64':>8s} {'non-f64 VALU':>12s} {'other':>8s} {'slots':>8s} "
          f"{'measured quad-cycles':>20s}")
    acc = Counter()
    tot_slots = tot_cyc = 0.0
    for reg in sorted(per):
        if reg not in entries:
            continue
        e = entries[reg]
        r = per[reg]
        n_all = sum(r.values()) * e
        f64 = (r["f64 arith"] + r["f64 trans"]) * e
        valu = sum(r[c] for c in VALU_CLASSES) * e
        for c in r:
            acc[c] += r[c] * e
        qc = cycles.get(reg, 0.0) / 4.0
        tot_slots += n_all
        tot_cyc += qc
        meas = f"{qc:20.0f}" if reg < 32 else f"{'(in its region)':>20s}"
        print(f"{REGION_NAMES.get(reg, str(reg)):28s} {e:8.2f} {n_all:8.0f} {f64:8.0f} {valu - f64:12.0f} {n_all - valu:8.0f} "
              f"{n_all:8.0f} {meas}")
    print(f"{'TOTAL':28s} {'':8s} {tot_slots:8.0f} {'':8s} {'':12s} {'':8s} {tot_slots:8.0f} {tot_cyc:20.0f}   waiting "
          f"{tot_cyc - tot_slots:.0f} (the profile build's own stamps included)")
    print("\n# dynamic estimate by instruction class (per column-step)")
    for c, v in acc.most_common():
        print(f"  {c:40s} {v:9.0f}  {100.0 * v / max(tot_slots, 1):5.1f} %")


if __name__ == "__main__":
    main()
