"""profiles/pmc_constants.json from the counter passes of tools/gpu_r4_pmc.sh.

    python tools/pmc_constants.py gpurun_out/<dir> [--tag r04] [--depth 300] [--model special]

Reads <dir>/{fetch,write,f64,fetch_cal,write_cal,kt}/*/*_counter_collection.csv (rocprofv3 --pmc, one pass per counter
group as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass), keeps the step-kernel rows as
profiles/<tag>_pmc_<pass>_cpl5.csv, and writes the constants bench.py reads:

  hbm_bytes_per_member_launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 / members      (gfx950: FETCH_SIZE counts half of a
      streamed read -- the calibration dispatch, a launch that only loads and stores psi, is checked against that here)
  f64_flop_per_column_step    = (ADD + MUL + TRANS + 2 FMA) wave instructions x 64 lanes / column-steps

keyed by the kernel hash in <dir>/library_hash.txt (hc_version() of the library the passes ran on).
"""
import argparse
import csv
import glob
import json
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rows_of(d, what):
    files = glob.glob(os.path.join(d, what, "*", "*_counter_collection.csv"))
    if not files:
        raise SystemExit(f"no counter CSV under {d}/{what}")
    with open(files[0], newline="") as fh:
        return [r for r in csv.DictReader(fh) if "step_kernel" in r["Kernel_Name"]], files[0]


def by_dispatch(rows):
    out = {}
    for r in rows:
        out.setdefault(int(r["Dispatch_Id"]), {}).setdefault(r["Counter_Name"], 0.0)
        out[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    return out


def keep(rows, src, dst):
    with open(src, newline="") as fh:
        header = next(csv.reader(fh))
    with open(dst, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=header)
        w.writeheader()
        w.writerows(rows)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--tag", default="r04")
    ap.add_argument("--depth", type=int, default=300)
    ap.add_argument("--model", default="special")
    ap.add_argument("--members", type=int, default=262144, help="members of the bench launches the passes profiled")
    ap.add_argument("--cal-members", type=int, default=65536)
    a = ap.parse_args()
    prof = os.path.join(R, "profiles")
    khash = open(os.path.join(a.dir, "library_hash.txt")).read().split()[-1]
    sources, vals = [], {}
    for what in ("fetch", "write", "f64"):
        rows, src = rows_of(a.dir, what)
        dst = os.path.join(prof, f"{a.tag}_pmc_{what}_cpl5.csv")
        keep(rows, src, dst)
        sources.append(os.path.relpath(dst, R))
        d = by_dispatch(rows)
        ids = sorted(d)
        timed = ids[1:] if len(ids) > 1 else ids            # the first launch is bench.py's warm-up day
        for name in d[ids[0]]:
            vals[name] = sum(d[i][name] for i in timed) / len(timed)
        vals[f"_{what}_launches"] = len(timed)
    # calibration: dispatch 1 of prof_kernel.py --calibrate only loads and stores psi (D x 8 bytes per member each way)
    cal = {}
    for what in ("fetch_cal", "write_cal"):
        rows, src = rows_of(a.dir, what)
        dst = os.path.join(prof, f"{a.tag}_pmc_{what}_cpl5.csv")
        keep(rows, src, dst)
        sources.append(os.path.relpath(dst, R))
        d = by_dispatch(rows)
        first = d[sorted(d)[0]]
        cal.update(first)
    state_kib = a.cal_members * a.depth * 8 / 1024.0
    fetch_ratio = cal["FETCH_SIZE"] / state_kib
    write_ratio = cal["WRITE_SIZE"] / state_kib
    if not (0.45 < fetch_ratio < 0.60):
        print(f"WARNING: the calibration dispatch fetched {fetch_ratio:.3f} of the state bytes by FETCH_SIZE (expected ~0.5: the "
              f"gfx950 x2 correction)", file=sys.stderr)
    col_steps = a.members * 48.0
    hbm = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0 / a.members
    flop = (vals["SQ_INSTS_VALU_ADD_F64"] + vals["SQ_INSTS_VALU_MUL_F64"] + vals["SQ_INSTS_VALU_TRANS_F64"]
            + 2.0 * vals["SQ_INSTS_VALU_FMA_F64"]) * 64.0 / col_steps
    rec = {"hbm_bytes_per_member_launch": hbm, "f64_flop_per_column_step": flop,
           "fetch_size_kib_per_launch": vals["FETCH_SIZE"], "write_size_kib_per_launch": vals["WRITE_SIZE"],
           "valu_wave_instructions_per_column_step": vals["SQ_INSTS_VALU"] / col_steps,
           "f64_wave_instructions_per_column_step": {k[14:-4].lower(): vals[k] / col_steps for k in vals if k.startswith("SQ_INSTS_VALU_") and k.endswith("_F64")},
           "wave_quad_cycles_per_column_step": vals["SQ_WAVE_CYCLES"] / col_steps,
           "calibration": {"fetch_size_over_state_bytes": fetch_ratio, "write_size_over_state_bytes": write_ratio},
           "launch_shape": f"{a.members} members x 48 rows x D = {a.depth} (bench.py --steps 2 --warmup 1: the mean of the timed launches)",
           "source": ", ".join(sources)}
    path = os.path.join(prof, "pmc_constants.json")
    table = {"kernel_hash": khash, "kernels": {}}
    if os.path.exists(path):
        old = json.load(open(path))
        if old.get("kernel_hash") == khash:
            table = old
    table["kernels"][f"{a.depth}/{a.model}"] = rec
    json.dump(table, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(rec, indent=1))
    # the kernel-trace summary of the same command
    ks = glob.glob(os.path.join(a.dir, "kt", "*", "*_kernel_stats.csv"))
    if ks:
        dst = os.path.join(prof, f"{a.tag}_kernel_stats_pmc_run.csv")
        open(dst, "w").write(open(ks[0]).read())
        print("kernel stats ->", os.path.relpath(dst, R))


if __name__ == "__main__":
    main()
