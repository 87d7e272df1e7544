#!/bin/bash
# Round 5: the one-wave generic kernels (7 - 10 cells per lane) and the 9 / 10-cell fallbacks: the two options + a scheduler strategy
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5zc}
mkdir -p $OUT
cd $ROOT
export HC_PROF_MEMBERS=32768
AB=tools/dev/_ab
P=hydromodel_amd/csrc/libhydrocol.so
ab() { timeout -k 10 500 python3 tools/dev/ab_interleaved.py "$@" | tee -a $OUT/ab.txt; }
export HYDROCOL_SPLIT_COLUMN=0
ab 541 2 $P $AB/lib_x910nr.so $AB/lib_x910nt.so $AB/lib_x910nm.so &&
ab 640 2 $P $AB/lib_x910nr.so $AB/lib_x910nt.so $AB/lib_x910nm.so &&
export HC_PROF_SOIL_N=1.7 && echo "generic exponents, n = 1.7" | tee -a $OUT/ab.txt &&
ab 401 2 $P $AB/lib_x78nm.so &&
ab 461 2 $P $AB/lib_x78nm.so &&
ab 541 2 $P $AB/lib_x910nr.so $AB/lib_x910nt.so $AB/lib_x910nm.so &&
ab 640 2 $P $AB/lib_x910nr.so $AB/lib_x910nt.so $AB/lib_x910nm.so
