#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5g}
mkdir -p $OUT
cd $ROOT
AB=tools/dev/_ab
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "spinup_on_gpu" > $OUT/pytest_s.log 2>&1
grep -h "spin-up:\|passed\|failed" $OUT/pytest_s.log
export HC_PROF_MEMBERS=32768
for d in 300 241 361; do
  timeout -k 10 900 python3 tools/dev/ab_interleaved.py $d 3 hydromodel_amd/csrc/libhydrocol.so $AB/lib_r5j.so $AB/lib_r5j0.so $AB/lib_r5rd.so | tee -a $OUT/ab.txt
done
