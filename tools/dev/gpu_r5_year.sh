#!/bin/bash
# Round 5: one year, member by member, GPU against the C oracle fed the same normals (the default well's depth runs on the new two-wave kernel)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5year}
mkdir -p $OUT
cd $ROOT
{
python3 -c "import sys; sys.path.insert(0, '.'); from hydromodel_amd import _lib; print('kernel hash', _lib.kernel_hash())"
for d in 401 300 101; do
  echo "== D = $d: python tools/dev/year_vs_oracle.py $d 0,5,1000,4000,77,31415"
  timeout -k 10 500 python3 tools/dev/year_vs_oracle.py $d 0,5,1000,4000,77,31415
done
} 2>&1 | tee $OUT/year_vs_oracle.txt
