#!/bin/bash
# Round 5: ET's first-midpoint call skipped (wave-uniform) when the top cell takes nothing up
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5zf}
mkdir -p $OUT
cd $ROOT
export HC_PROF_MEMBERS=32768
AB=tools/dev/_ab
ab() { timeout -k 10 500 python3 tools/dev/ab_interleaved.py "$@" | tee -a $OUT/ab.txt; }
ab 300 3 $AB/lib_base0.so $AB/lib_etskip.so &&
ab 241 2 $AB/lib_base0.so $AB/lib_etskip.so &&
ab 361 2 $AB/lib_base0.so $AB/lib_etskip.so &&
ab 401 2 $AB/lib_base0.so $AB/lib_etskip.so &&
ab 581 2 $AB/lib_base0.so $AB/lib_etskip.so &&
HC_PROF_SOIL_N=1.7 ab 300 2 $AB/lib_base0.so $AB/lib_etskip.so
