#!/bin/bash
# Round 5: two waves per SIMD at 7 / 8 cells per lane once more, now that the spills are gone (machine-LICM off + sink [+ trackers])
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5x}
mkdir -p $OUT
cd $ROOT
export HC_PROF_MEMBERS=32768
AB=tools/dev/_ab
ab() { timeout -k 10 500 python3 tools/dev/ab_interleaved.py "$@" | tee -a $OUT/ab.txt; }
ab 401 2 hydromodel_amd/csrc/libhydrocol.so $AB/lib_two7a.so $AB/lib_two7b.so $AB/lib_two7c.so $AB/lib_two7d.so &&
ab 461 2 hydromodel_amd/csrc/libhydrocol.so $AB/lib_two8a.so $AB/lib_two8b.so &&
ab 300 1 hydromodel_amd/csrc/libhydrocol.so $AB/lib_nlst.so &&
ab 361 1 hydromodel_amd/csrc/libhydrocol.so $AB/lib_nlst6.so
