#!/bin/bash
# Round 5: the one case of the second fuzz run that leaves the thresholds (seed 61 case 111, D = 385), on the product (TWO layout at 7 cells per lane)
# and on a build of the same source with the layout off at 7 cells (one wave per SIMD)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5fuzz2r}
mkdir -p $OUT
cd $ROOT
{
echo "== product"
timeout -k 10 300 python3 tools/dev/fuzz_vs_oracle.py 240 61 20 --only 111 2>&1 | grep -i "case 111\|cases"
echo "== the same source with -DHC_TWO_MASK=112 (7 cells per lane on the one-wave layout)"
HC_LIB=tools/dev/_ab/lib_one7.so timeout -k 10 300 python3 tools/dev/fuzz_vs_oracle.py 240 61 20 --only 111 2>&1 | grep -i "case 111\|cases"
} | tee $OUT/replay.txt
