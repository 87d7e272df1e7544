#!/bin/bash
# Round 5, GPU call C: GPU test suite on the new library, A/B by depth (digests), CPL-7 TWO and generic CPL-6 TWO trials.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5c}
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
AB=tools/dev/_ab
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -15 $OUT/pytest.log
export HC_PROF_MEMBERS=32768
bash tools/dev/ab_depths.sh $OUT/ab.txt "241 300 361" $AB/lib_r4.so hydromodel_amd/csrc/libhydrocol.so > /dev/null 2>&1
bash tools/dev/ab_depths.sh $OUT/ab_deep.txt "401 541" $AB/lib_r4.so hydromodel_amd/csrc/libhydrocol.so > /dev/null 2>&1
bash tools/dev/ab_depths.sh $OUT/ab_two7.txt "401" $AB/lib_r5two7.so > /dev/null 2>&1
cat $OUT/ab.txt $OUT/ab_deep.txt $OUT/ab_two7.txt
for lib in hydromodel_amd/csrc/libhydrocol.so $AB/lib_r5gen6.so; do
  timeout -k 10 300 python3 tools/prof_generic_lib.py $lib 1.7 1.0 32768 361 >> $OUT/gen6.txt 2>&1
done
cat $OUT/gen6.txt | grep column-days
