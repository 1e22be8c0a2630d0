#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03k
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_learner_kernels_gpu.py tests/test_a2c_gpu.py tests/test_agent_kernel_gpu.py -q -m gpu -x > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc" | tee -a $O/status.txt; tail -15 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python tools/bench_gemm_hip.py --tune --rounds 3 > $O/gemm.json 2> $O/gemm.err
echo "gemm rc=$?" | tee -a $O/status.txt
timeout -k 10 300 python bench.py --mode a2c --envs 8192 > $O/a2c.json 2> $O/a2c.err
echo "a2c rc=$?" | tee -a $O/status.txt
tail -c 1500 $O/a2c.json
