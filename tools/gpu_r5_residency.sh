#!/bin/bash
# Round 5: is the D = 300 kernel's fabric traffic a bandwidth bound, and does it live in the Infinity Cache?  
# -> gpurun_out/<dir>/residency.txt (tools/fabric_residency.py: grid-size sweep with counters, then the cache-evicting copy)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5res}
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
export HC_RES_IC=$OUT/ic300.npz
python3 tools/fabric_residency.py cus 256 > $OUT/ic.log 2>&1
run() { name=$1; shift; rocprofv3 "$@" > $OUT/$name.log 2>&1; echo "$name rc $?"; }
for n in 64 128 192 256; do
  python3 tools/fabric_residency.py cus $n >> $OUT/residency.txt 2>> $OUT/err.log
  run f$n --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f$n -- python3 tools/fabric_residency.py cus $n
  run w$n --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w$n -- python3 tools/fabric_residency.py cus $n
  run t$n --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/t$n -- python3 tools/fabric_residency.py cus $n
done
python3 tools/fabric_residency.py polluter 128 >> $OUT/residency.txt 2>> $OUT/err.log
python3 tools/fabric_residency.py polluter 192 >> $OUT/residency.txt 2>> $OUT/err.log
python3 - <<PY >> $OUT/residency.txt
import csv, glob
for n in (64, 128, 192, 256):
    vals = {}
    for tag in ("f", "w", "t"):
        for f in glob.glob(f"$OUT/{tag}{n}/*/*_counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if "step_kernel" in r["Kernel_Name"]:
                    vals[r["Counter_Name"]] = vals.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    if len(vals) < 4:
        print(f"grid {n}: counters incomplete {sorted(vals)}"); continue
    cs = 128 * n * 48.0
    print(f"grid {n:3d}: FETCH_SIZE x 2 = {2 * vals['FETCH_SIZE'] * 1024 / cs / 1e3:6.1f} KB, WRITE_SIZE = {vals['WRITE_SIZE'] * 1024 / cs / 1e3:6.1f} KB per "
          f"column-step; L2 hit rate {vals['TCC_HIT_sum'] / (vals['TCC_HIT_sum'] + vals['TCC_MISS_sum']):.3f}")
PY
find $OUT -name "*_agent_info.csv" -delete
find $OUT -type f -size +4M -delete
cat $OUT/residency.txt; tail -3 $OUT/err.log
