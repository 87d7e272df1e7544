"""profiles/pmc_constants.json from the counter passes of tools/gpu_r5_pmc.sh.

    python tools/pmc_constants.py gpurun_out/<dir> [--tag r05]

For every kernel key of the script (<depth>/<model>: 300/special, 200/special, 300/generic, 401/special, 581/special) reads
<dir>/<key>_{fetch,write,f64,tcc}/*/*_counter_collection.csv (rocprofv3 --pmc, one pass per counter group as
MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass; ONE 48-row step-kernel launch per pass),
keeps the step-kernel rows as profiles/<tag>_pmc_<key>_<pass>.csv, and writes the constants bench.py reads:

  fabric_bytes_per_member_launch = (fetch_factor x FETCH_SIZE + write_factor x WRITE_SIZE) KiB x 1024 / members
      -- requests between the L2s and the fabric, Infinity-Cache hits and HBM accesses alike (MI355X_MICROARCH.md § HBM);
      the factors come from the calibration kernels of tools/pmc_calib.hip profiled in the same call (bytes moved are KNOWN
      there): 1 / (FETCH_SIZE per byte loaded) of cal_buf_load and 1 / (WRITE_SIZE per byte stored) of cal_buf_store --
      the step kernel's own instruction shapes (raw_buffer_load/store_b64, 512 B per wave instruction);
  f64_flop_per_column_step       = (ADD + MUL + TRANS + 2 FMA) wave instructions x 64 lanes / column-steps;
  l2_hit_rate                    = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)

keyed by the kernel hash in <dir>/library_hash.txt (hc_version() of the library the passes ran on).  Also writes
profiles/<tag>_pmc_calib.txt: every calibration kernel's counters against its known bytes.
"""
import argparse
import csv
import glob
import json
import os

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"300/special": 262144, "200/special": 65536, "300/generic": 65536, "401/special": 65536, "581/special": 32768}
ROWS = 48


def counter_rows(d, what, match):
    files = glob.glob(os.path.join(d, what, "*", "*_counter_collection.csv"))
    if not files:
        return None, None
    with open(files[0], newline="") as fh:
        rows = [r for r in csv.DictReader(fh) if match in r["Kernel_Name"]]
    return rows, files[0]


def by_dispatch(rows):
    out = {}
    for r in rows:
        k = (int(r["Dispatch_Id"]), r["Kernel_Name"])
        out.setdefault(k, {}).setdefault(r["Counter_Name"], 0.0)
        out[k][r["Counter_Name"]] += float(r["Counter_Value"])
    return out


def keep(rows, src, dst):
    with open(src, newline="") as fh:
        header = next(csv.reader(fh))
    with open(dst, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=header)
        w.writeheader()
        w.writerows(rows)


def calibration(d, prof, tag):
    """{kernel: {bytes_stored, bytes_loaded, footprint, ms, FETCH_SIZE, WRITE_SIZE, hit, miss}} + the two factors."""
    plain = {r["kernel"]: r for r in csv.DictReader(open(os.path.join(d, "calib_plain.csv")))}
    table = {}
    for what in ("calib_fetch", "calib_write", "calib_tcc"):
        rows, _ = counter_rows(d, what, "cal_")
        for (_, name), vals in by_dispatch(rows or []).items():
            short = name.split("(")[0].replace("void ", "").replace("<", "_").replace(">", "")
            table.setdefault(short, {}).update(vals)
    lines = [f"{'kernel':16s} {'stored GiB':>10s} {'loaded GiB':>10s} {'footprint MiB':>13s} {'ms':>8s} {'TB/s':>6s} "
             f"{'WRITE_SIZE/stored':>17s} {'FETCH_SIZE/loaded':>17s} {'L2 hit rate':>11s}"]
    for name, p in plain.items():
        c = table.get(name, {})
        st, ld = float(p["bytes_stored"]), float(p["bytes_loaded"])
        ms = float(p["ms"])
        wr = c.get("WRITE_SIZE", 0.0) * 1024 / st if st else float("nan")
        fe = c.get("FETCH_SIZE", 0.0) * 1024 / ld if ld else float("nan")
        hit = c.get("TCC_HIT_sum", 0.0) / max(c.get("TCC_HIT_sum", 0.0) + c.get("TCC_MISS_sum", 0.0), 1.0)
        lines.append(f"{name:16s} {st / 2**30:10.2f} {ld / 2**30:10.2f} {float(p['footprint_bytes']) / 2**20:13.0f} {ms:8.3f} "
                     f"{(st + ld) / (ms * 1e-3) / 1e12:6.2f} {wr:17.3f} {fe:17.3f} {hit:11.3f}")
        table.setdefault(name, {}).update(write_ratio=wr, fetch_ratio=fe)
    head = ("# tools/pmc_calib.hip under rocprofv3 (tools/gpu_r5_pmc.sh): what FETCH_SIZE / WRITE_SIZE report for KNOWN byte counts in\n"
            "# the step kernel's access shapes, on its launch shape (256 workgroups x 512 threads).  cal_cycle_<KB>: 2 048 waves\n"
            "# each store and then load a region of <KB> KiB over and over (the TWO layout's pattern) -- plain timing (ms) from the\n"
            "# un-profiled run.\n")
    open(os.path.join(prof, f"{tag}_pmc_calib.txt"), "w").write(head + "\n".join(lines) + "\n")
    print("\n".join(lines))
    fetch_factor = 1.0 / table["cal_buf_load"]["fetch_ratio"]
    write_factor = 1.0 / table["cal_buf_store"]["write_ratio"]
    return {"fetch_factor": fetch_factor, "write_factor": write_factor,
            "fetch_size_per_byte_loaded": {k: v["fetch_ratio"] for k, v in table.items() if v.get("fetch_ratio") == v.get("fetch_ratio")},
            "write_size_per_byte_stored": {k: v["write_ratio"] for k, v in table.items() if v.get("write_ratio") == v.get("write_ratio")},
            "source": f"profiles/{tag}_pmc_calib.txt (tools/pmc_calib.hip)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--tag", default="r05")
    a = ap.parse_args()
    prof = os.path.join(R, "profiles")
    khash = open(os.path.join(a.dir, "library_hash.txt")).read().split()[-1]
    cal = calibration(a.dir, prof, a.tag)
    table = {"kernel_hash": khash, "calibration": cal, "kernels": {}}
    for key, members in KEYS.items():
        tagk = key.replace("/", "_")
        vals, sources = {}, []
        for what in ("fetch", "write", "f64", "tcc"):
            rows, src = counter_rows(a.dir, f"{tagk}_{what}", "step_kernel")
            if not rows:
                break
            dst = os.path.join(prof, f"{a.tag}_pmc_{tagk}_{what}.csv")
            keep(rows, src, dst)
            sources.append(os.path.relpath(dst, R))
            d = by_dispatch(rows)
            if len(d) != 1:
                raise SystemExit(f"{key}/{what}: {len(d)} step-kernel dispatches, expected one")
            vals.update(next(iter(d.values())))
        else:
            col_steps = members * float(ROWS)
            fabric = (cal["fetch_factor"] * vals["FETCH_SIZE"] + cal["write_factor"] * vals["WRITE_SIZE"]) * 1024.0 / members
            flop = (vals["SQ_INSTS_VALU_ADD_F64"] + vals["SQ_INSTS_VALU_MUL_F64"] + vals["SQ_INSTS_VALU_TRANS_F64"]
                    + 2.0 * vals["SQ_INSTS_VALU_FMA_F64"]) * 64.0 / col_steps
            ms = None
            ks = glob.glob(os.path.join(a.dir, f"{tagk}_kt", "*", "*_kernel_stats.csv"))
            if ks:
                dst = os.path.join(prof, f"{a.tag}_kernel_stats_{tagk}.csv")
                open(dst, "w").write(open(ks[0]).read())
                for r in csv.DictReader(open(ks[0])):
                    if "step_kernel" in r["Name"]:
                        ms = float(r["AverageNs"]) * 1e-6
            table["kernels"][key] = {
                "fabric_bytes_per_member_launch": fabric, "fabric_bytes_per_column_step": fabric / ROWS,
                "fetch_size_kib_per_launch": vals["FETCH_SIZE"], "write_size_kib_per_launch": vals["WRITE_SIZE"],
                "l2_hit_rate": vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]),
                "f64_flop_per_column_step": flop,
                "valu_wave_instructions_per_column_step": vals["SQ_INSTS_VALU"] / col_steps,
                "f64_wave_instructions_per_column_step": {k[14:-4].lower(): vals[k] / col_steps for k in vals
                                                          if k.startswith("SQ_INSTS_VALU_") and k.endswith("_F64")},
                "wave_quad_cycles_per_column_step": vals["SQ_WAVE_CYCLES"] / col_steps,
                "launch_ms_kernel_trace": ms,
                "launch_shape": f"{members} members x {ROWS} rows x D = {key.split('/')[0]}, one launch (tools/prof_kernel.py)",
                "source": ", ".join(sources)}
            continue
        print(f"(no complete counter passes for {key})")
    json.dump(table, open(os.path.join(prof, "pmc_constants.json"), "w"), indent=1, sort_keys=True)
    for key, rec in table["kernels"].items():
        print(f"{key:12s} fabric {rec['fabric_bytes_per_column_step'] / 1000:7.1f} KB per column-step, L2 hit rate "
              f"{rec['l2_hit_rate']:.3f}, {rec['f64_flop_per_column_step'] / 1e6:.3f} MFLOP and "
              f"{rec['valu_wave_instructions_per_column_step'] / 1e3:.1f} k VALU instructions per column-step, "
              f"launch {rec['launch_ms_kernel_trace']} ms")


if __name__ == "__main__":
    main()
