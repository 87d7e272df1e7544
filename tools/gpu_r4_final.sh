#!/bin/bash
# Round-4 closing measurements on the GPU box (run through gpurun): counter passes of the headline kernel
# (tools/gpu_r4_pmc.sh), the default bench line, the same command under rocprofv3 --kernel-trace --stats, the sweep bench.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r4final}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
bash tools/gpu_r4_pmc.sh ${TAG}_pmc > $OUT/pmc.log 2>&1; echo "pmc rc $?"
IC=$OUT/ic_d300.npz
python3 bench.py --ic-file $IC > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_bench -- python3 bench.py --no-cpu-baseline --no-sustained --no-heavy --ic-file $IC > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err; echo "bench profiled rc $?"
python3 bench.py --workload sweep > $OUT/bench_sweep.json 2> $OUT/bench_sweep.err; echo "sweep rc $?"
find $OUT -name "*_agent_info.csv" -delete
find $OUT -name "*_kernel_trace.csv" -size +1M -delete
grep step_kernel $OUT/kt_bench/*/*_kernel_stats.csv
cut -c1-250 $OUT/bench_default.json
