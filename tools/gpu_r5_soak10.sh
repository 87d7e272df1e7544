#!/bin/bash
# Round 5: the WHOLE 10-year record on the final build (every state checked finite every 73 days)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5soak10}
mkdir -p $OUT
cd $ROOT
python3 -c "import sys; sys.path.insert(0, '.'); from hydromodel_amd import _lib; print('kernel hash', _lib.kernel_hash())" | tee $OUT/hash.txt
timeout -k 10 500 python3 tools/soak.py 16384 300 10 2>&1 | tee $OUT/soak_16384_d300_10yr.txt | tail -4 &&
timeout -k 10 500 python3 tools/soak.py 8192 401 10 2>&1 | tee $OUT/soak_8192_d401_10yr.txt | tail -4
