#!/bin/bash
# Round 5: BASELINE config 5 at its full size through a whole month in one handle, on the final build
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5month}
mkdir -p $OUT
cd $ROOT
python3 -c "import sys; sys.path.insert(0, '.'); from hydromodel_amd import _lib; print('kernel hash', _lib.kernel_hash())" | tee $OUT/hash.txt
timeout -k 10 1100 python3 tools/sweep_soak.py 4096 300 30 2 2>&1 | tee $OUT/sweep512x4096_month.txt | cut -c1-200
