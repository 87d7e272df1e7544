#!/bin/bash
# Round 5: 2 / 3 cells per lane (the reference's shallow wells), other combinations of the new options
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5za}
mkdir -p $OUT
cd $ROOT
export HC_PROF_MEMBERS=32768
AB=tools/dev/_ab
P=hydromodel_amd/csrc/libhydrocol.so
ab() { timeout -k 10 500 python3 tools/dev/ab_interleaved.py "$@" | tee -a $OUT/ab.txt; }
ab 101 2 $P $AB/lib_w23t.so $AB/lib_w23s.so $AB/lib_w23st.so $AB/lib_w23nm.so $AB/lib_w23nr.so $AB/lib_w23h.so &&
ab 121 1 $P $AB/lib_w23t.so $AB/lib_w23s.so $AB/lib_w23st.so $AB/lib_w23nm.so $AB/lib_w23nr.so $AB/lib_w23h.so &&
ab 192 2 $P $AB/lib_w23t.so $AB/lib_w23s.so $AB/lib_w23st.so $AB/lib_w23nm.so $AB/lib_w23nr.so $AB/lib_w23h.so
