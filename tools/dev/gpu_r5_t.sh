#!/bin/bash
# Round 5: -disable-machine-licm (+ trackers / + sink-insts-to-avoid-spills) on the two-wave units, interleaved A/B
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5t}
mkdir -p $OUT
cd $ROOT
export HC_PROF_MEMBERS=32768
AB=tools/dev/_ab
timeout -k 10 500 python3 tools/dev/ab_interleaved.py 300 3 $AB/lib_s0.so $AB/lib_nl.so $AB/lib_nlt.so $AB/lib_nls.so | tee -a $OUT/ab.txt &&
timeout -k 10 400 python3 tools/dev/ab_interleaved.py 241 2 $AB/lib_s0.so $AB/lib_nl.so $AB/lib_nlt.so $AB/lib_nls.so | tee -a $OUT/ab.txt &&
timeout -k 10 400 python3 tools/dev/ab_interleaved.py 361 2 $AB/lib_s0.so $AB/lib_nl6.so $AB/lib_nlt6.so $AB/lib_nls6.so | tee -a $OUT/ab.txt
