#!/bin/bash
# round 4, first GPU call: the one-launch rotation schedule (tests, then the A/B of the three launch forms), ADVICE fixes' tests
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04a
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_step_many_gpu.py -q -x > $O/tests_step_many.log 2>&1
rc=$?; echo "step_many tests rc=$rc" | tee -a $O/status.txt; tail -5 $O/tests_step_many.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/r04_many_ab.py 4096:20 4096:100 8192:20 8192:100 > $O/many_ab.json 2> $O/many_ab.err
rc=$?; echo "many_ab rc=$rc" | tee -a $O/status.txt; tail -c 1500 $O/many_ab.json
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python -m pytest tests/test_full_size_parity_gpu.py -q -x -k "step_many" > $O/tests_full_size.log 2>&1
rc=$?; echo "full size rc=$rc" | tee -a $O/status.txt; tail -5 $O/tests_full_size.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python -m pytest tests/test_learner_kernels_gpu.py -q -x -k "forward_reuse or validate or tuning" > $O/tests_advice.log 2>&1
rc=$?; echo "advice tests rc=$rc" | tee -a $O/status.txt; tail -5 $O/tests_advice.log
exit $rc
