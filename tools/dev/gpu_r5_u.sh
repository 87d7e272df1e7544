#!/bin/bash
# Round 5: machine-LICM off + sink-to-avoid-spills (+ trackers / iterative-maxocc) on every unit family, interleaved A/B
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5u}
mkdir -p $OUT
cd $ROOT
export HC_PROF_MEMBERS=32768
AB=tools/dev/_ab
ab() { timeout -k 10 500 python3 tools/dev/ab_interleaved.py "$@" | tee -a $OUT/ab.txt; }
ab 300 2 $AB/lib_s0.so $AB/lib_nls.so $AB/lib_nlst.so $AB/lib_nlsm.so &&
ab 241 2 $AB/lib_s0.so $AB/lib_nls.so $AB/lib_nlst.so $AB/lib_nlsm.so &&
ab 361 2 $AB/lib_s0.so $AB/lib_nls6.so $AB/lib_nlst6.so $AB/lib_nlsm6.so &&
ab 401 2 $AB/lib_s78.so $AB/lib_nls78.so $AB/lib_nlst78.so &&
ab 461 2 $AB/lib_s78.so $AB/lib_nls78.so $AB/lib_nlst78.so &&
ab 581 2 $AB/lib_s0.so $AB/lib_nls.so $AB/lib_nlst.so $AB/lib_nlsm.so &&
export HC_PROF_SOIL_N=1.7 && echo "generic exponents, n = 1.7" | tee -a $OUT/ab.txt &&
ab 300 2 $AB/lib_s0.so $AB/lib_gnls45.so &&
ab 241 2 $AB/lib_s0.so $AB/lib_gnls45.so &&
ab 361 2 $AB/lib_s0.so $AB/lib_gnls6.so
