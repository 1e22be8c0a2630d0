#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04n
mkdir -p $O
cd $R
for i in 1 2 3; do
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-a2c --no-cpu-baseline > $O/bench_driver_style_$i.json 2> $O/bench_driver_style_$i.err
python -c "
import json;d=json.loads(open('$O/bench_driver_style_$i.json').read().strip().splitlines()[-1]);r=d['roofline'];print('20-step value %.4g'%d['value'], 'ms/step', d['ms_per_step'], 'gpu us/launch', r['avg_launch_us'], 'single-step %.4g'%d.get('single_step_launch_value',0), {k:'%.3g'%v['value'] for k,v in d['other_launch_forms'].items()})"
done
