#!/bin/bash
# generic-exponent kernel at 5 cells per lane: model batch size and scheduler settings (two interleaved rounds)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5q}
mkdir -p $OUT
cd $ROOT
for r in 1 2; do
for lib in gq_base gq_b1 gq_b2 gq_b4 gq_track gq_maxocc gq_nohrp; do
  timeout -k 10 200 python3 tools/prof_generic_lib.py tools/dev/_ab/lib_$lib.so 1.7 1.0 32768 300 2>&1 | grep column-days | sed 's/counters.*sha/sha/' | tee -a $OUT/gen.txt
done
done
