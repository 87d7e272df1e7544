#!/bin/bash
# Round 5, GPU call D: the split column on the TWO layout (four pairs per CU) against the round-4 form (two pairs per CU).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5d}
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
AB=tools/dev/_ab
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -15 $OUT/pytest.log
export HC_PROF_MEMBERS=16384
bash tools/dev/ab_depths.sh $OUT/ab_pair.txt "581 640" $AB/lib_r5pair0.so hydromodel_amd/csrc/libhydrocol.so > /dev/null 2>&1
export HYDROCOL_SPLIT_COLUMN=1
bash tools/dev/ab_depths.sh $OUT/ab_pair541.txt "541" $AB/lib_r5pair0.so hydromodel_amd/csrc/libhydrocol.so > /dev/null 2>&1
export HYDROCOL_SPLIT_COLUMN=0
bash tools/dev/ab_depths.sh $OUT/ab_one541.txt "541 581" hydromodel_amd/csrc/libhydrocol.so > /dev/null 2>&1
unset HYDROCOL_SPLIT_COLUMN
cat $OUT/ab_pair.txt; echo "split forced:"; cat $OUT/ab_pair541.txt; echo "one wave:"; cat $OUT/ab_one541.txt
for lib in $AB/lib_r5pair0.so hydromodel_amd/csrc/libhydrocol.so; do
  timeout -k 10 300 python3 tools/prof_generic_lib.py $lib 1.7 1.0 16384 581 >> $OUT/gen581.txt 2>&1
done
grep column-days $OUT/gen581.txt
