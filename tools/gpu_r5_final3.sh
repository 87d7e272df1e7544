#!/bin/bash
# Round 5 final, part 3: whole-record soaks of the new kernels (heavy regime: failed attempts, budget trips, Jacobian refreshes)
# and BASELINE config 2 at its size
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5j3}
mkdir -p $OUT
cd $ROOT
timeout -k 10 500 python3 tools/dev/cfg2_bench.py 0 48 17472 > $OUT/cfg2.json 2>&1; cat $OUT/cfg2.json
timeout -k 10 400 python3 tools/soak.py 65536 300 1 > $OUT/soak_65536_d300_1yr.txt 2>&1; tail -3 $OUT/soak_65536_d300_1yr.txt
timeout -k 10 300 python3 tools/soak.py 16384 581 1 > $OUT/soak_16384_d581_1yr.txt 2>&1; tail -3 $OUT/soak_16384_d581_1yr.txt
timeout -k 10 300 python3 tools/soak.py 32768 200 1 > $OUT/soak_32768_d200_1yr.txt 2>&1; tail -3 $OUT/soak_32768_d200_1yr.txt
timeout -k 10 300 python3 tools/soak.py 16384 361 1 > $OUT/soak_16384_d361_1yr.txt 2>&1; tail -3 $OUT/soak_16384_d361_1yr.txt
timeout -k 10 300 python3 tools/soak.py 16384 401 1 > $OUT/soak_16384_d401_1yr.txt 2>&1; tail -3 $OUT/soak_16384_d401_1yr.txt
HC_PROF_D=401 timeout -k 10 300 python3 tools/prof_phases.py tools/dev/_ab/lib_prof_two.so 8192 > $OUT/phases_d401.txt 2>&1; head -3 $OUT/phases_d401.txt
