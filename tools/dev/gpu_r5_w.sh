#!/bin/bash
# Round 5: cell-model batch sizes once the spills are gone (machine-LICM off + sink-to-avoid-spills [+ trackers]), other schedulers on top
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5w}
mkdir -p $OUT
cd $ROOT
export HC_PROF_MEMBERS=32768
AB=tools/dev/_ab
ab() { timeout -k 10 500 python3 tools/dev/ab_interleaved.py "$@" | tee -a $OUT/ab.txt; }
ab 300 2 $AB/lib_nlst.so $AB/lib_t5b3.so $AB/lib_t5b4.so $AB/lib_t5h.so $AB/lib_n5minreg.so $AB/lib_n5maxilp.so &&
ab 581 2 $AB/lib_nlst.so $AB/lib_t5b3.so $AB/lib_t5b4.so $AB/lib_t5h.so $AB/lib_n5minreg.so $AB/lib_n5maxilp.so &&
ab 361 2 $AB/lib_nlst6.so $AB/lib_t6b3.so $AB/lib_t6b6.so $AB/lib_n6g4.so &&
ab 401 2 $AB/lib_nls78.so $AB/lib_n7b4.so $AB/lib_n7b7.so &&
ab 461 2 $AB/lib_nls78.so $AB/lib_n8b4.so $AB/lib_n8b8.so &&
ab 241 2 $AB/lib_nls.so $AB/lib_n4t.so &&
export HC_PROF_SOIL_N=1.7 && echo "generic exponents, n = 1.7" | tee -a $OUT/ab.txt &&
ab 300 2 $AB/lib_gnlst45.so $AB/lib_t5b3.so $AB/lib_t5b4.so &&
ab 361 2 $AB/lib_gnls6.so $AB/lib_t6b3.so $AB/lib_t6b6.so $AB/lib_n6g4.so &&
ab 581 2 $AB/lib_s0.so $AB/lib_nlst.so $AB/lib_gnlst45.so $AB/lib_t5b3.so
