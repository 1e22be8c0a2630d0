#!/bin/bash
# round 4, second GPU call: half-batch pipelined rollout (tests + A/B + a trace), step_range parity
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04b
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_learner_kernels_gpu.py tests/test_a2c_gpu.py tests/test_hip_parity.py tests/test_step_many_gpu.py -q -x > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc" | tee -a $O/status.txt; tail -8 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/ab_collect.py default unsplit > $O/ab_collect.json 2> $O/ab_collect.err
rc=$?; echo "ab_collect rc=$rc" | tee -a $O/status.txt; cat $O/ab_collect.json; tail -3 $O/ab_collect.err
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --mode a2c > $O/bench_a2c.json 2> $O/bench_a2c.err
rc=$?; echo "bench a2c rc=$rc" | tee -a $O/status.txt; python -c "
import json;d=json.loads(open('$O/bench_a2c.json').read().strip().splitlines()[-1])['a2c'];print({k:d[k] for k in ('value','ms_per_rollout','collect_ms_per_rollout','update_ms_per_rollout')})"
export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_a2c -- python3 $R/tools/prof_a2c_run.py > $O/trace_a2c.log 2>&1
echo "trace rc=$?" | tee -a $O/status.txt
cd $R
find $O -name "*agent_info.csv" -delete
ls -la $O/trace_a2c/*/ 2>/dev/null | head
