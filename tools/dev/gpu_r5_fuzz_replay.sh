#!/bin/bash
# the two out-of-tier fuzz cases of round 5 on the shipped library and on the round-4 library
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5fuzz2}
mkdir -p $OUT
cd $ROOT
for spec in "51 133" "52 143"; do
  set -- $spec
  for lib in "" tools/dev/_ab/lib_r4.so; do
    echo "== seed $1 case $2 on ${lib:-the shipped library}"
    HC_LIB=$lib timeout -k 10 200 python3 tools/dev/fuzz_vs_oracle.py 240 $1 20 --only $2 2>&1 | grep "case $2"
    if [ "$2" = 143 ]; then echo "   (split column forced off:)"; HYDROCOL_SPLIT_COLUMN=0 HC_LIB=$lib timeout -k 10 200 python3 tools/dev/fuzz_vs_oracle.py 240 $1 20 --only $2 2>&1 | grep "case $2"; fi
  done
done 2>&1 | tee $OUT/replay.txt
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
