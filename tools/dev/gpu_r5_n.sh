#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5n}
mkdir -p $OUT
cd $ROOT
AB=tools/dev/_ab
for r in 1 2; do
for lib in g_none g_cur g_noalias g_nord g_nodiet g_nord_nodiet g_b3 r5commit; do
  timeout -k 10 200 python3 tools/prof_generic_lib.py $AB/lib_$lib.so 1.7 1.0 32768 300 2>&1 | grep column-days | sed 's/counters.*sha/sha/' | tee -a $OUT/gen.txt
done
done
export HC_PROF_MEMBERS=32768
timeout -k 10 600 python3 tools/dev/ab_interleaved.py 401 2 $AB/lib_r5one7.so $AB/lib_r5two7.so | tee -a $OUT/two7.txt
