#!/bin/bash
# Round 5: control-flow / clustering options on top of the adopted settings (5 and 7 cells per lane, split column, generic 5)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5zz}
mkdir -p $OUT
cd $ROOT
export HC_PROF_MEMBERS=32768
AB=tools/dev/_ab
P=hydromodel_amd/csrc/libhydrocol.so
ab() { timeout -k 10 500 python3 tools/dev/ab_interleaved.py "$@" | tee -a $OUT/ab.txt; }
ab 300 2 $P $AB/lib_v5nocl.so $AB/lib_v5phi8.so $AB/lib_v5phi0.so $AB/lib_v5nobf.so $AB/lib_v5nobp.so &&
ab 581 2 $P $AB/lib_v5nocl.so $AB/lib_v5phi8.so $AB/lib_v5phi0.so $AB/lib_v5nobf.so $AB/lib_v5nobp.so &&
ab 401 2 $P $AB/lib_v7nocl.so $AB/lib_v7nobf.so $AB/lib_v7phi8.so &&
export HC_PROF_SOIL_N=1.7 && echo "generic exponents, n = 1.7" | tee -a $OUT/ab.txt &&
ab 300 2 $P $AB/lib_v5nocl.so $AB/lib_v5phi8.so $AB/lib_v5phi0.so $AB/lib_v5nobf.so $AB/lib_v5nobp.so
