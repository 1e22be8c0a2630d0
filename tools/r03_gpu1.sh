#!/bin/bash
# round 3, first GPU call: the new parity tests, then the step_many A/B + rotation bound
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03c
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_launch_variants_gpu.py tests/test_step_many_gpu.py tests/test_full_size_parity_gpu.py tests/test_hip_parity.py tests/test_shim_dropin.py tests/test_capi_c_client_gpu.py -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" | tee -a $O/status.txt
tail -15 $O/tests.log
ROUNDS=3 timeout -k 10 300 python tools/r03_many_ab.py > $O/many_ab.json 2> $O/many_ab.err
echo "many_ab rc=$?" | tee -a $O/status.txt
cat $O/many_ab.json
