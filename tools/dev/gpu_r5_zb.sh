#!/bin/bash
# Round 5: 2 / 3 cells per lane, generic exponents, machine-LICM off + sink + the unit's scheduler strategy
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5zb}
mkdir -p $OUT
cd $ROOT
export HC_PROF_MEMBERS=32768
AB=tools/dev/_ab
P=hydromodel_amd/csrc/libhydrocol.so
ab() { timeout -k 10 500 python3 tools/dev/ab_interleaved.py "$@" | tee -a $OUT/ab.txt; }
export HC_PROF_SOIL_N=1.7 && echo "generic exponents, n = 1.7" | tee -a $OUT/ab.txt &&
ab 101 2 $P $AB/lib_w23nm.so $AB/lib_w23nr.so $AB/lib_w23s.so $AB/lib_w23h.so &&
ab 192 2 $P $AB/lib_w23nm.so $AB/lib_w23nr.so $AB/lib_w23s.so $AB/lib_w23h.so
