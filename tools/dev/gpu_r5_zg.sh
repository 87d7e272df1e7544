#!/bin/bash
# Round 5: non-temporal hint on the Jacobian's group vectors (written and read once per column-step)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5zg}
mkdir -p $OUT
cd $ROOT
export HC_PROF_MEMBERS=32768
AB=tools/dev/_ab
ab() { timeout -k 10 500 python3 tools/dev/ab_interleaved.py "$@" | tee -a $OUT/ab.txt; }
ab 401 3 $AB/lib_st0.so $AB/lib_st2.so &&
ab 300 2 $AB/lib_st0.so $AB/lib_st2.so &&
ab 581 2 $AB/lib_st0.so $AB/lib_st2.so
