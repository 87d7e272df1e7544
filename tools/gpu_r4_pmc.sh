#!/bin/bash
# Round-4 counter passes of the HEADLINE kernel on the build that ships (run through gpurun): FETCH_SIZE / WRITE_SIZE /
# fp64 instruction classes of hc::step_kernel<5, true, 4, false, 1> at bench.py's own launch shape (262 144 members x 48
# rows x D = 300), the program itself after `--`, the initial condition from --ic-file so that the profiled process
# launches nothing but ensemble steps.  tools/pmc_constants.py turns the CSVs into profiles/pmc_constants.json.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r4pmc}
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
IC=$OUT/ic_d300.npz
B="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sustained --no-heavy --ic-file $IC"
python3 $B > $OUT/bench_plain.json 2> $OUT/bench_plain.err || exit 1      # (writes the initial-condition cache)
run() { name=$1; shift; rocprofv3 "$@" > $OUT/$name.log 2>&1; echo "$name rc $?"; }
run kt     --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $B
run fetch  --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $B
run write  --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $B
run f64    --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT/f64 -- python3 $B
# the calibration dispatch of MI355X_MICROARCH.md's FETCH_SIZE note (a launch that only loads and stores psi)
W="tools/prof_kernel.py 300 65536 48 --ic $OUT/ic_cache.npz"
python3 $W > $OUT/ic.log 2>&1
run fetch_cal --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_cal -- python3 $W --calibrate
run write_cal --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_cal -- python3 $W --calibrate
find $OUT -name "*_agent_info.csv" -delete
find $OUT -type f -size +4M -delete
python3 -c "import sys; sys.path.insert(0, '.'); from hydromodel_amd import _lib; print(_lib.kernel_hash())" > $OUT/library_hash.txt 2>&1
ls -R $OUT | head -60
