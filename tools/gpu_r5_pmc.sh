#!/bin/bash
# Round-5 counter passes on the build that ships (run through gpurun): FETCH_SIZE / WRITE_SIZE / fp64 instruction classes /
# L2 hit rate of every step kernel a BASELINE config or a reference well uses, each as ONE 48-row launch from a cached
# initial condition (tools/prof_kernel.py), plus the calibration of the two memory counters on known byte counts
# (tools/pmc_calib.hip).  tools/pmc_constants.py turns the CSVs into profiles/pmc_constants.json + profiles/r05_pmc_*.csv.
#   key          depth members  extra arguments of prof_kernel.py
#   300/special  300   262144                      BASELINE configs[2]: the bench's own launch shape (5 cells per lane, TWO layout)
#   200/special  200   65536                       configs[1]'s kernel (4 cells per lane, TWO layout)
#   300/generic  300   65536   --generic --n 1.7   configs[4]'s kernel (generic exponents)
#   401/special  401   65536                       the reference's shipped well 10 (7 cells per lane, one wave per SIMD)
#   581/special  581   32768                       wells 14 / 15: the split column (two waves per member)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5pmc}
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
run() { name=$1; shift; rocprofv3 "$@" > $OUT/$name.log 2>&1; echo "$name rc $?"; }
if [ ! -x tools/dev/_ab/pmc_calib ]; then mkdir -p tools/dev/_ab; hipcc --offload-arch=gfx950 -O3 -o tools/dev/_ab/pmc_calib tools/pmc_calib.hip || exit 1; fi
tools/dev/_ab/pmc_calib > $OUT/calib_plain.csv 2> $OUT/calib_plain.err || exit 1
run calib_fetch --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -- tools/dev/_ab/pmc_calib
run calib_write --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -- tools/dev/_ab/pmc_calib
run calib_tcc   --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/calib_tcc -- tools/dev/_ab/pmc_calib
while read key depth members extra; do
  [ -z "$key" ] && continue
  tag=$(echo $key | tr '/' '_')
  W="tools/prof_kernel.py $depth $members 48 --ic $OUT/ic_cache.npz $extra"
  python3 $W > $OUT/${tag}_ic.log 2>&1           # (fills the initial-condition cache and exits)
  python3 $W > $OUT/${tag}_plain.log 2>&1 || { echo "$key: plain run failed"; continue; }
  run ${tag}_kt    --kernel-trace --stats --output-format csv -d $OUT/${tag}_kt -- python3 $W
  run ${tag}_fetch --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${tag}_fetch -- python3 $W
  run ${tag}_write --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${tag}_write -- python3 $W
  run ${tag}_f64   --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT/${tag}_f64 -- python3 $W
  run ${tag}_tcc   --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/${tag}_tcc -- python3 $W
  tail -1 $OUT/${tag}_plain.log
done <<SPECS
300/special 300 262144
200/special 200 65536
300/generic 300 65536 --generic --n 1.7
401/special 401 65536
581/special 581 32768
SPECS
find $OUT -name "*_agent_info.csv" -delete
find $OUT -type f -size +4M -delete
python3 -c "import sys; sys.path.insert(0, '.'); from hydromodel_amd import _lib; print(_lib.kernel_hash())" > $OUT/library_hash.txt 2>&1
cat $OUT/library_hash.txt
