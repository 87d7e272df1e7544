#!/bin/bash
# Round 5 final, part 2: the driver's bench command, the same command under rocprofv3 (kernel trace), region profiles
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5j2}
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
timeout -k 10 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --no-sustained --no-heavy --no-n1e6 --no-cpu-baseline > $OUT/bench_kt.json 2> $OUT/bench_kt.err; echo "bench under rocprofv3 rc $?"
timeout -k 10 600 python3 bench.py --workload sweep > $OUT/bench_sweep.json 2> $OUT/bench_sweep.err; echo "sweep rc $?"
timeout -k 10 300 python3 tools/prof_phases.py tools/dev/_ab/lib_prof_two.so 8192 > $OUT/phases_d300.txt 2>&1
HC_PROF_D=581 timeout -k 10 300 python3 tools/prof_phases.py tools/dev/_ab/lib_prof_two.so 4096 > $OUT/phases_d581.txt 2>&1
timeout -k 10 600 python3 -m pytest tests/test_gpu_bench.py -m gpu -q 2>&1 | tail -3
find $OUT -name "*_agent_info.csv" -delete
find $OUT -type f -size +4M -delete
python3 - <<PY
import json
for name in ("bench", "bench_kt", "bench_sweep"):
    try:
        d = json.load(open("$OUT/%s.json" % name))
    except Exception as e:
        print(name, "unreadable", e); continue
    r = d["roofline"]
    print(name, round(d["value"]), "col-days/s; launch ms", round(r["launch_ms"], 1), r["launch_ms_min"], r["launch_ms_max"], "frac", round(r["frac"], 5),
          "fabric", r["fabric"] and {k: round(v, 3) for k, v in r["fabric"].items() if isinstance(v, float)}, "valu", d.get("valu_f64") and round(d["valu_f64"]["frac"], 3))
    for k in ("sustained", "sustained_heavy", "n1e6", "cpu_baseline"):
        if d.get(k): print("   ", k, round(d[k]["value"]), {a: d[k][a] for a in ("members", "days", "cores") if a in d[k]})
PY
head -5 $OUT/phases_d581.txt
