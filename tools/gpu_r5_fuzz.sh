#!/bin/bash
# Round 5 fuzzes on the final build (new seeds): random configurations against the oracle, random sweeps against stand-alone handles
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5fuzz}
mkdir -p $OUT
cd $ROOT
{
python3 -c "import sys; sys.path.insert(0, '.'); from hydromodel_amd import _lib; print('kernel hash', _lib.kernel_hash())"
for seed in 51 52 53; do
  echo "== python tools/dev/fuzz_vs_oracle.py 240 $seed 20 =="
  timeout -k 10 400 python3 tools/dev/fuzz_vs_oracle.py 240 $seed 20 > $OUT/fuzz_$seed.log 2>&1; echo "rc $?"; grep -i "out of tier\|OUT\|cases" $OUT/fuzz_$seed.log | tail -6
done
echo "== python tools/dev/fuzz_vs_oracle.py 160 54 16 --deep =="
timeout -k 10 400 python3 tools/dev/fuzz_vs_oracle.py 160 54 16 --deep > $OUT/fuzz_deep.log 2>&1; echo "rc $?"; grep -i "out of tier\|cases" $OUT/fuzz_deep.log | tail -4
for seed in 51 52; do
  echo "== python tools/dev/fuzz_sweep.py 300 $seed =="
  timeout -k 10 300 python3 tools/dev/fuzz_sweep.py 300 $seed > $OUT/sweep_$seed.log 2>&1; echo "rc $?"; tail -1 $OUT/sweep_$seed.log
done
} 2>&1 | tee $OUT/summary.txt
