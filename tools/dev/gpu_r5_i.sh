#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5i}
mkdir -p $OUT
cd $ROOT
AB=tools/dev/_ab
rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power" > $OUT/smi.txt
export HC_PROF_MEMBERS=32768
for d in 300 241 361; do
  timeout -k 10 900 python3 tools/dev/ab_interleaved.py $d 2 $AB/lib_r5commit.so hydromodel_amd/csrc/libhydrocol.so $AB/lib_r5rd0.so | tee -a $OUT/ab.txt
done
export HC_PROF_MEMBERS=16384
timeout -k 10 900 python3 tools/dev/ab_interleaved.py 581 2 $AB/lib_r5commit.so hydromodel_amd/csrc/libhydrocol.so | tee -a $OUT/ab.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $OUT/pytest.log
