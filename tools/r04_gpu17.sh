#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04w
mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests/test_hip_parity.py tests/test_launch_variants_gpu.py tests/test_full_size_parity_gpu.py tests/test_step_many_gpu.py -q -x > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc" | tee -a $O/status.txt; tail -4 $O/tests.log
[ $rc -eq 0 ] || exit $rc
bash tools/r04_regression_ab.sh 2>&1 | grep "c5\|seq"
