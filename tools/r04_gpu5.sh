#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04f
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_learner_kernels_gpu.py tests/test_step_many_gpu.py tests/test_hip_parity.py -q -x > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc" | tee -a $O/status.txt; tail -5 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/bench_head.py > $O/bench_head.json 2> $O/bench_head.err
rc=$?; echo "bench_head rc=$rc" | tee -a $O/status.txt; cat $O/bench_head.json
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/ab_collect.py default unsplit stagger > $O/ab_collect.json 2> $O/ab_collect.err
rc=$?; echo "ab_collect rc=$rc" | tee -a $O/status.txt; cat $O/ab_collect.json; tail -3 $O/ab_collect.err
