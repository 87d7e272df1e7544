#!/bin/bash
# Round 5, after the launch-length policy changed (host code only: same kernel hash): GPU suite, the driver's bench command, config 2 with the library's choice
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5fb}
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
timeout -k 10 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --no-sustained --no-heavy --no-n1e6 --no-cpu-baseline > $OUT/bench_kt.json 2> $OUT/bench_kt.err; echo "bench under rocprofv3 rc $?"
timeout -k 10 300 python3 tools/dev/cfg2_bench.py 0 48 17472 > $OUT/cfg2.json 2>&1; cat $OUT/cfg2.json
timeout -k 10 400 python3 tools/soak.py 65536 300 1 > $OUT/soak_65536_d300_1yr.txt 2>&1; tail -3 $OUT/soak_65536_d300_1yr.txt
find $OUT -name "*_agent_info.csv" -delete
find $OUT -type f -size +4M -delete
python3 - <<PY
import json
for name in ("bench", "bench_kt"):
    d = json.load(open("$OUT/%s.json" % name))
    r = d["roofline"]
    print(name, round(d["value"]), "col-days/s; launch ms", round(r["launch_ms"], 1), r["launch_ms_min"], r["launch_ms_max"], "frac", round(r["frac"], 5),
          "fabric", r["fabric"] and round(r["fabric"]["achieved"], 1), "valu", d.get("valu_f64") and round(d["valu_f64"]["frac"], 3))
    for k in ("sustained", "sustained_heavy", "n1e6", "cpu_baseline"):
        if d.get(k): print("   ", k, round(d[k]["value"]), {a: d[k][a] for a in ("members", "days", "cores", "launches") if a in d[k]})
PY
