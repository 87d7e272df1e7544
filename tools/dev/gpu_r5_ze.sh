#!/bin/bash
# Round 5: two source-level variants again, now that the spills are gone: Jacobian rows carried in registers into the factorisation; no reciprocal table
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5ze}
mkdir -p $OUT
cd $ROOT
export HC_PROF_MEMBERS=32768
AB=tools/dev/_ab
P=hydromodel_amd/csrc/libhydrocol.so
ab() { timeout -k 10 500 python3 tools/dev/ab_interleaved.py "$@" | tee -a $OUT/ab.txt; }
ab 300 2 $P $AB/lib_jf.so $AB/lib_rd0.so &&
ab 581 2 $P $AB/lib_jf.so $AB/lib_rd0.so &&
ab 401 2 $P $AB/lib_jf7.so
