#!/bin/bash
# Round 5: region profile at 2 and 4 cells per lane (where does the shallow kernel's time go?)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5zh}
mkdir -p $OUT
cd $ROOT
HC_PROF_D=101 timeout -k 10 300 python3 tools/prof_phases.py tools/dev/_ab/lib_prof2.so 8192 > $OUT/phases_d101.txt 2>&1
HC_PROF_D=241 timeout -k 10 300 python3 tools/prof_phases.py tools/dev/_ab/lib_prof2.so 8192 > $OUT/phases_d241.txt 2>&1
cat $OUT/phases_d101.txt
