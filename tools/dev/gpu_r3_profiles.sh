#!/bin/bash
# Round-3 profiling session on the GPU box (run through gpurun): kernel traces and PMC passes of the kernels that had
# none so far -- generic exponents at D = 300, CPL 4 (D = 200), CPL 7 (D = 401, the reference's default well) -- and the
# per-class instruction counters of the headline kernel.  Every rocprofv3 run has the python program itself after `--`.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3e
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
IC=$OUT/ic_cache.npz
W_GEN="tools/prof_kernel.py 300 65536 48 --generic --n 1.7 --ic $IC"
W_D200="tools/prof_kernel.py 200 65536 48 --ic $IC"
W_D401="tools/prof_kernel.py 401 16384 48 --ic $IC"
W_D300="tools/prof_kernel.py 300 65536 48 --ic $IC"
for w in "$W_GEN" "$W_D200" "$W_D401" "$W_D300"; do python3 $w >> $OUT/ic.log 2>&1; done     # cache the initial conditions
run() {   # name, rocprof args..., -- workload
  name=$1; shift
  rocprofv3 "$@" > $OUT/$name.log 2>&1
  echo "$name rc $?"
}
run kt_generic_d300 --kernel-trace --stats --output-format csv -d $OUT/kt_generic_d300 -- python3 $W_GEN
run kt_cpl4_d200    --kernel-trace --stats --output-format csv -d $OUT/kt_cpl4_d200 -- python3 $W_D200
run kt_cpl7_d401    --kernel-trace --stats --output-format csv -d $OUT/kt_cpl7_d401 -- python3 $W_D401
run pmc_fetch_generic --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_generic -- python3 $W_GEN --calibrate
run pmc_write_generic --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_generic -- python3 $W_GEN --calibrate
run pmc_fetch_cpl4 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_cpl4 -- python3 $W_D200 --calibrate
run pmc_write_cpl4 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_cpl4 -- python3 $W_D200 --calibrate
run pmc_f64_generic --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_f64_generic -- python3 $W_GEN
run pmc_f64_cpl4 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_f64_cpl4 -- python3 $W_D200
# the headline kernel: issue slots and instruction classes (two passes of <= 8 SQ counters)
run pmc_issue_d300 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d $OUT/pmc_issue_d300 -- python3 $W_D300
run pmc_class_d300 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $OUT/pmc_class_d300 -- python3 $W_D300
run pmc_wait_d300 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INST_CYCLES_SALU --output-format csv -d $OUT/pmc_wait_d300 -- python3 $W_D300
# keep only what is small: stats and counter CSVs
find $OUT -name "*_agent_info.csv" -delete
find $OUT -type f -size +4M -delete
ls -R $OUT | head -80
