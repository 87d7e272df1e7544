#!/bin/bash
# Round 5: more fuzz seeds on the final build (the TWO layout at 7 cells per lane is new: --deep draws D = 385 ... 640)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5fuzz2}
mkdir -p $OUT
cd $ROOT
{
python3 -c "import sys; sys.path.insert(0, '.'); from hydromodel_amd import _lib; print('kernel hash', _lib.kernel_hash())"
for seed in 61 62; do
  echo "== python tools/dev/fuzz_vs_oracle.py 240 $seed 20 =="
  timeout -k 10 400 python3 tools/dev/fuzz_vs_oracle.py 240 $seed 20 > $OUT/fuzz_$seed.log 2>&1; echo "rc $?"; grep -i "out of tier\|cases" $OUT/fuzz_$seed.log | tail -6
done
for seed in 63 64 65; do
  echo "== python tools/dev/fuzz_vs_oracle.py 160 $seed 16 --deep =="
  timeout -k 10 400 python3 tools/dev/fuzz_vs_oracle.py 160 $seed 16 --deep > $OUT/fuzz_deep_$seed.log 2>&1; echo "rc $?"; grep -i "out of tier\|cases" $OUT/fuzz_deep_$seed.log | tail -4
done
echo "== python tools/dev/fuzz_sweep.py 300 61 =="
timeout -k 10 300 python3 tools/dev/fuzz_sweep.py 300 61 > $OUT/sweep_61.log 2>&1; echo "rc $?"; tail -1 $OUT/sweep_61.log
} 2>&1 | tee $OUT/summary.txt
