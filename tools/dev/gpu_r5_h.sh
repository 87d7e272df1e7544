#!/bin/bash
# clocks / power while the same kernel runs fast or slow
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5h}
mkdir -p $OUT
cd $ROOT
AB=tools/dev/_ab
export HC_PROF_MEMBERS=32768
rocm-smi --showclocks --showpower --showtemp > $OUT/smi_idle.txt 2>&1
n=0
for lib in hydromodel_amd/csrc/libhydrocol.so $AB/lib_r5j0.so $AB/lib_r5commit.so hydromodel_amd/csrc/libhydrocol.so $AB/lib_r5j0.so; do
  n=$((n+1)); name=${n}_$(basename $lib .so)
  ( for i in $(seq 1 60); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk" | tr '\n' ' '; echo; sleep 0.2; done ) > $OUT/smi_$name.txt 2>&1 &
  SMI=$!
  timeout -k 10 300 python3 tools/prof_depth.py $lib 300 2>&1 | tee -a $OUT/ab.txt
  kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
  echo "--- $name: samples"; sort $OUT/smi_$name.txt | uniq -c | sort -rn | head -4
done
