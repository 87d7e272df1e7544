#!/bin/bash
# Round 5, GPU call B: GPU test suite on the new library, A/B (digests), counters of the D = 300 kernel r4 / r5.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5b}
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
AB=tools/dev/_ab
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
export HC_PROF_MEMBERS=32768
bash tools/dev/ab_depths.sh $OUT/ab.txt "241 300 361" $AB/lib_r4.so $AB/lib_r5a.so $AB/lib_r5d.so $AB/lib_r5e.so hydromodel_amd/csrc/libhydrocol.so > /dev/null 2>&1
echo "ab done"; cat $OUT/ab.txt
run() { name=$1; shift; rocprofv3 "$@" > $OUT/$name.log 2>&1; echo "$name rc $?"; }
W="tools/prof_kernel.py 300 65536 48 --ic $OUT/ic_cache.npz"
export HC_LIB=$ROOT/hydromodel_amd/csrc/libhydrocol.so
python3 $W > $OUT/ic.log 2>&1
for lib in r4 r5; do
  if [ $lib = r5 ]; then export HC_LIB=$ROOT/hydromodel_amd/csrc/libhydrocol.so; else export HC_LIB=$ROOT/$AB/lib_$lib.so; fi
  run ${lib}_fetch --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${lib}_fetch -- python3 $W
  run ${lib}_write --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${lib}_write -- python3 $W
  run ${lib}_tcc --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/${lib}_tcc -- python3 $W
done
find $OUT -name "*_agent_info.csv" -delete
find $OUT -type f -size +4M -delete
grep -h "col-days" $OUT/*.log | head -20
