#!/bin/bash
# Round 5: second scan around machine-LICM off + sink-to-avoid-spills, the remaining unit families
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5v}
mkdir -p $OUT
cd $ROOT
export HC_PROF_MEMBERS=32768
AB=tools/dev/_ab
ab() { timeout -k 10 500 python3 tools/dev/ab_interleaved.py "$@" | tee -a $OUT/ab.txt; }
ab 300 2 $AB/lib_nlst.so $AB/lib_nlstg.so $AB/lib_nlsto2.so &&
ab 241 2 $AB/lib_nls.so $AB/lib_nlstg.so $AB/lib_nlsto2.so &&
ab 581 2 $AB/lib_nlst.so $AB/lib_nlstg.so $AB/lib_nlsto2.so &&
ab 401 2 $AB/lib_nls78.so $AB/lib_nlsh78.so &&
ab 461 2 $AB/lib_nls78.so $AB/lib_nlsh78.so &&
ab 101 2 $AB/lib_s23.so $AB/lib_nls23.so $AB/lib_nlst23.so &&
ab 192 2 $AB/lib_s23.so $AB/lib_nls23.so $AB/lib_nlst23.so &&
HYDROCOL_SPLIT_COLUMN=0 ab 541 1 $AB/lib_s910.so $AB/lib_nls910.so &&
HYDROCOL_SPLIT_COLUMN=0 ab 640 1 $AB/lib_s910.so $AB/lib_nls910.so &&
export HC_PROF_SOIL_N=1.7 && echo "generic exponents, n = 1.7" | tee -a $OUT/ab.txt &&
ab 300 2 $AB/lib_gnls45.so $AB/lib_gnlst45.so &&
ab 241 2 $AB/lib_gnls45.so $AB/lib_gnlst45.so &&
ab 361 2 $AB/lib_gnls6.so $AB/lib_gnlst6.so &&
ab 401 2 $AB/lib_s78.so $AB/lib_nls78.so $AB/lib_nlst78.so $AB/lib_nlsh78.so &&
ab 461 2 $AB/lib_s78.so $AB/lib_nls78.so $AB/lib_nlst78.so $AB/lib_nlsh78.so &&
ab 101 1 $AB/lib_s23.so $AB/lib_nls23.so $AB/lib_nlst23.so &&
ab 192 1 $AB/lib_s23.so $AB/lib_nls23.so $AB/lib_nlst23.so &&
ab 581 1 $AB/lib_s0.so $AB/lib_gnls45.so $AB/lib_gnlst45.so
