#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5r}
mkdir -p $OUT
cd $ROOT
export HC_PROF_MEMBERS=32768
for d in 300 241 361; do
  timeout -k 10 600 python3 tools/dev/ab_interleaved.py $d 3 tools/dev/_ab/lib_r5cur.so tools/dev/_ab/lib_r5md.so tools/dev/_ab/lib_r5md2.so | tee -a $OUT/ab.txt
done
