#!/bin/bash
# Round 5: launch length (member-days per launch) for small and middle ensembles
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5zd}
mkdir -p $OUT
cd $ROOT
{
echo "== config 2 (4 096 members x D = 200, the year): rows per launch 768 (today's automatic choice) 1536 3072 6144"
timeout -k 10 300 python3 tools/dev/cfg2_bench.py 768 1536 3072 6144
for rpl in 48 192 768; do
  echo "== 65 536 members x D = 300 x 120 days of the 1-year forcing, HYDROCOL_ROWS_PER_LAUNCH=$rpl"
  HYDROCOL_ROWS_PER_LAUNCH=$rpl timeout -k 10 300 python3 bench.py --no-sustained --no-n1e6 --no-cpu-baseline --steps 2 --members 4096 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d.get('sustained_heavy'); print('sustained_heavy', h and round(h['value']), h and h.get('launches'))"
done
for rpl in 192 768 3072; do
  echo "== 16 384 members x D = 300, one year, HYDROCOL_ROWS_PER_LAUNCH=$rpl"
  HYDROCOL_ROWS_PER_LAUNCH=$rpl timeout -k 10 300 python3 tools/soak.py 16384 300 1 | tail -3 | head -1
done
} 2>&1 | tee $OUT/launch_length.txt
