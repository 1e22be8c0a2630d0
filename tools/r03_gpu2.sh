#!/bin/bash
# round 3: the new / changed parity tests only
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03b
mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests/test_launch_variants_gpu.py tests/test_step_many_gpu.py tests/test_full_size_parity_gpu.py tests/test_hip_parity.py tests/test_shim_dropin.py tests/test_capi_c_client_gpu.py -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" | tee -a $O/status.txt
tail -25 $O/tests.log
