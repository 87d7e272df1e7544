#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5f}
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $OUT/pytest.log
timeout -k 10 600 python3 -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -m gpu -q -s -k "whole_year or spinup_on_gpu or north_star" > $OUT/pytest_s.log 2>&1
grep -h "whole year\|year-long\|spin-up:\|1 048 576\|passed\|failed" $OUT/pytest_s.log
export HC_PROF_MEMBERS=32768
bash tools/dev/ab_depths.sh $OUT/ab_alias.txt "241 300 361" hydromodel_amd/csrc/libhydrocol.so tools/dev/_ab/lib_r5j.so tools/dev/_ab/lib_r5j0.so > /dev/null 2>&1
cat $OUT/ab_alias.txt
timeout -k 10 300 python3 tools/prof_phases.py tools/dev/_ab/lib_prof_two.so 8192 > $OUT/phases_d300.txt 2>&1; cat $OUT/phases_d300.txt
timeout -k 10 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; python3 -c "
import json
d=json.load(open('$OUT/bench.json'))
print({k:d[k] for k in ('value','ms_per_step')})
print('roofline', {k:v for k,v in d['roofline'].items() if k not in ('traffic_source','note')})
print('valu', d['valu_f64'])
for k in ('sustained','sustained_heavy','n1e6','cpu_baseline'):
    print(k, {a:b for a,b in d[k].items() if a in ('value','members','days','wall_s','launch_ms','launch_ms_min','launch_ms_max','failed_attempts_per_member_year','cores')})
"
