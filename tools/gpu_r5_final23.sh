#!/bin/bash
# Round 5 final, parts 2 + 3 + the fuzzes in one call
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash tools/gpu_r5_final2.sh ${1:-r5f}2 && bash tools/gpu_r5_final3.sh ${1:-r5f}3 && bash tools/gpu_r5_fuzz.sh ${1:-r5f}fuzz
