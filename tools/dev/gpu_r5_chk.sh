#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5chk}
mkdir -p $OUT
cd $ROOT
date +%s.%N | tee $OUT/t0.txt
ls -la gpurun_out | head -5
timeout -k 10 1000 python3 -m pytest tests -m gpu -q --durations=15 > $OUT/pytest.log 2>&1; echo "pytest rc $?"
date +%s.%N | tee $OUT/t1.txt
tail -25 $OUT/pytest.log
