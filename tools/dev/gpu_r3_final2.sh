#!/bin/bash
# Round-3 closing measurements after the waves-per-SIMD change (run through gpurun): GPU tests, depth table, kernel traces
# of the shallow kernel (two waves per SIMD) and of a split-column depth, the default bench line with its rocprofv3
# kernel stats, the sweep bench line.  Every rocprofv3 run has the python program itself after `--`.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3z
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1
echo "gpu tests rc $?"; tail -2 $OUT/gpu_tests.log
HC_PROF_MEMBERS=16384 timeout -k 10 300 python3 tools/prof_depth.py hydromodel_amd/csrc/libhydrocol.so 101 121 128 192 200 241 261 300 361 401 421 461 512 513 541 581 640 > $OUT/depths.txt 2>&1
echo "depths rc $?"
IC=$OUT/ic_cache.npz
for w in "101 65536 48" "581 16384 48"; do timeout -k 10 200 python3 tools/prof_kernel.py $w --ic $IC >> $OUT/ic.log 2>&1; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_cpl2_d101 -- python3 tools/prof_kernel.py 101 65536 48 --ic $IC > $OUT/kt_cpl2_d101.log 2>&1
echo "kt cpl2 rc $?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_cpl2_d101 -- python3 tools/prof_kernel.py 101 65536 48 --ic $IC > $OUT/pmc_cpl2_d101.log 2>&1
echo "pmc cpl2 rc $?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_split_d581 -- python3 tools/prof_kernel.py 581 16384 48 --ic $IC > $OUT/kt_split_d581.log 2>&1
echo "kt split rc $?"
timeout -k 10 500 python3 bench.py --ic-file $OUT/ic_d300.npz > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench rc $?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_bench -- python3 bench.py --no-cpu-baseline --no-sustained --no-heavy --ic-file $OUT/ic_d300.npz > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
echo "bench profiled rc $?"
timeout -k 10 400 python3 bench.py --workload sweep --no-cpu-baseline > $OUT/bench_sweep.json 2> $OUT/bench_sweep.err
echo "bench sweep rc $?"
find $OUT -name "*_agent_info.csv" -delete
find $OUT -name "*_kernel_trace.csv" -size +1M -delete
find $OUT -type f -size +4M -delete
cat $OUT/depths.txt
grep step_kernel $OUT/kt_cpl2_d101/*/*_kernel_stats.csv
grep step_kernel $OUT/kt_split_d581/*/*_kernel_stats.csv
grep step_kernel $OUT/kt_bench/*/*_kernel_stats.csv
cut -c1-600 $OUT/bench_default.json
cut -c1-400 $OUT/bench_sweep.json
