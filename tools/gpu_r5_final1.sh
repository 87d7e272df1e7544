#!/bin/bash
# Round 5 final, part 1: GPU suite, counter passes of the shipped build, throughput by depth
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5j1}
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -6 $OUT/pytest.log
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "spinup_on_gpu" 2>&1 | grep "spin-up:"
bash tools/gpu_r5_pmc.sh $(basename $OUT)/pmc 2>&1 | tail -12
export HC_PROF_MEMBERS=16384
timeout -k 10 600 python3 tools/prof_depth.py hydromodel_amd/csrc/libhydrocol.so 101 128 192 200 241 261 300 361 401 421 461 512 541 581 640 > $OUT/depths.txt 2>&1
cat $OUT/depths.txt
