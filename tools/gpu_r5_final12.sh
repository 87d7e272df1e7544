#!/bin/bash
# Round 5 final, parts 1 + 2 in one call
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash tools/gpu_r5_final1.sh ${1:-r5f}1 && bash tools/gpu_r5_final2.sh ${1:-r5f}2
