#!/bin/bash
# A/B on the GPU box: tools/dev/ab_depths.sh <out.txt> "<depths>" lib_a.so lib_b.so ...   (HC_PROF_MEMBERS members, 2 days each)
out=$1; depths=$2; shift 2
mkdir -p $(dirname $out)
for lib in "$@"; do
  timeout -k 10 300 python3 tools/prof_depth.py $lib $depths >> $out 2>&1 || echo "$lib FAILED rc $?" >> $out
done
cat $out
