#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5k}
mkdir -p $OUT
cd $ROOT
AB=tools/dev/_ab
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -6 $OUT/pytest.log
timeout -k 10 300 python3 -m pytest tests/test_gpu_round5.py -m gpu -q -s 2>&1 | grep "recorded rows"
export HC_PROF_MEMBERS=32768
for d in 300 241 361; do
  timeout -k 10 900 python3 tools/dev/ab_interleaved.py $d 2 $AB/lib_r5base.so $AB/lib_r5diet.so $AB/lib_r5rd0.so hydromodel_amd/csrc/libhydrocol.so | tee -a $OUT/ab.txt
done
