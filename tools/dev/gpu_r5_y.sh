#!/bin/bash
# Round 5: two waves per SIMD at 7 cells per lane, second scan (+ generic exponents)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5y}
mkdir -p $OUT
cd $ROOT
export HC_PROF_MEMBERS=32768
AB=tools/dev/_ab
ab() { timeout -k 10 500 python3 tools/dev/ab_interleaved.py "$@" | tee -a $OUT/ab.txt; }
ab 401 2 hydromodel_amd/csrc/libhydrocol.so $AB/lib_two7d.so $AB/lib_two7e.so $AB/lib_two7f.so $AB/lib_two7h.so &&
ab 448 1 hydromodel_amd/csrc/libhydrocol.so $AB/lib_two7d.so $AB/lib_two7f.so &&
export HC_PROF_SOIL_N=1.7 && echo "generic exponents, n = 1.7" | tee -a $OUT/ab.txt &&
ab 401 2 hydromodel_amd/csrc/libhydrocol.so $AB/lib_two7f.so $AB/lib_two7g.so $AB/lib_two7h.so
