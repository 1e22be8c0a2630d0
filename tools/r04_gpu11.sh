#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04m
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_step_many_gpu.py tests/test_full_size_parity_gpu.py -q -x -k "rotation or hand_off or step_many" > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc" | tee -a $O/status.txt; tail -4 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/r04_many_ab.py 4096:20 4096:100 8192:20 8192:100 > $O/many_ab.json 2> $O/many_ab.err
rc=$?; echo "many_ab rc=$rc" | tee -a $O/status.txt; python -c "
import json;d=json.load(open('$O/many_ab.json'));print({k:min(v) for k,v in d['us_per_call'].items()}); print({k:min(v) for k,v in d['us_single_call_wall'].items()})"
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-a2c > $O/bench_driver_style.json 2> $O/bench_driver_style.err
python -c "
import json;d=json.loads(open('$O/bench_driver_style.json').read().strip().splitlines()[-1]);r=d['roofline'];print('20-step value %.4g'%d['value'], 'frac %.3f'%r['frac'], r['schedule'], 'traffic', r['traffic'], r.get('traffic_unavailable_because'), 'valu', r.get('valu_issue_frac'))"
timeout -k 10 600 python bench.py --no-a2c > $O/bench_default.json 2> $O/bench_default.err
python -c "
import json;d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]);r=d['roofline'];print('default value %.4g'%d['value'], 'frac %.3f'%r['frac'], r['schedule'], 'traffic', r['traffic'], r.get('traffic_unavailable_because'), 'valu', r.get('valu_issue_frac'), 'moved_frac', r.get('moved_frac'))"
