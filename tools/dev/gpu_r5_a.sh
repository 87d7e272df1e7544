#!/bin/bash
# Round 5, GPU call A: (1) A/B of the traffic cuts (state digests must be identical), (2) counter calibration on known byte
# counts (tools/pmc_calib.hip), (3) FETCH / WRITE_SIZE + L2 hit rate of the D = 300 kernel before / after.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5a}
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
AB=tools/dev/_ab
export HC_PROF_MEMBERS=32768
bash tools/dev/ab_depths.sh $OUT/ab.txt "241 300 361" hydromodel_amd/csrc/libhydrocol.so $AB/lib_r5a.so $AB/lib_r5b.so $AB/lib_r5c.so > /dev/null 2>&1
echo "ab done"; cat $OUT/ab.txt
$AB/pmc_calib > $OUT/calib_plain.csv 2> $OUT/calib_plain.err && echo "calib plain ok"
run() { name=$1; shift; rocprofv3 "$@" > $OUT/$name.log 2>&1; echo "$name rc $?"; }
run calib_fetch --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -- $AB/pmc_calib
run calib_write --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -- $AB/pmc_calib
run calib_tcc --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/calib_tcc -- $AB/pmc_calib
W="tools/prof_kernel.py 300 65536 48 --ic $OUT/ic_cache.npz"
python3 $W > $OUT/ic.log 2>&1
for lib in base r5a; do
  if [ $lib = base ]; then export HC_LIB=$ROOT/hydromodel_amd/csrc/libhydrocol.so; else export HC_LIB=$ROOT/$AB/lib_$lib.so; fi
  run ${lib}_fetch --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${lib}_fetch -- python3 $W
  run ${lib}_write --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${lib}_write -- python3 $W
  run ${lib}_tcc --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/${lib}_tcc -- python3 $W
done
find $OUT -name "*_agent_info.csv" -delete
find $OUT -type f -size +4M -delete
ls -R $OUT | head -80
