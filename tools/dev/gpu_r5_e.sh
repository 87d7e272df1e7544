#!/bin/bash
# Round 5, GPU call E: GPU suite on the shipped build, then its counter passes and the residency experiment.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5e}
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -25 $OUT/pytest.log
bash tools/gpu_r5_pmc.sh $(basename $OUT)/pmc 2>&1 | tail -40
bash tools/gpu_r5_residency.sh $(basename $OUT)/res 2>&1 | tail -30
