#!/bin/bash
# Round-3 closing measurements on the GPU box (run through gpurun): depth table, kernel trace of a split-column depth,
# the default bench line with its rocprofv3 kernel stats.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3m
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
python3 tools/prof_depth.py hydromodel_amd/csrc/libhydrocol.so 101 200 300 361 401 461 512 513 541 581 640 > $OUT/depths.txt 2>&1
echo "depths rc $?"
IC=$OUT/ic_cache.npz
python3 tools/prof_kernel.py 581 16384 48 --ic $IC > $OUT/ic.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_split_d581 -- python3 tools/prof_kernel.py 581 16384 48 --ic $IC > $OUT/kt_split_d581.log 2>&1
echo "kt split rc $?"
python3 bench.py --ic-file $OUT/ic_d300.npz > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench rc $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_bench -- python3 bench.py --no-cpu-baseline --no-sustained --no-heavy --ic-file $OUT/ic_d300.npz > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
echo "bench profiled rc $?"
find $OUT -name "*_agent_info.csv" -delete
find $OUT -name "*_kernel_trace.csv" -size +1M -delete
cat $OUT/depths.txt
grep step_kernel $OUT/kt_split_d581/*/*_kernel_stats.csv
grep step_kernel $OUT/kt_bench/*/*_kernel_stats.csv
cut -c1-400 $OUT/bench_default.json
