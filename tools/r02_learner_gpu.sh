#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02b
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_learner_kernels_gpu.py tests/test_agent_kernel_gpu.py tests/test_a2c_gpu.py tests/test_hardening_gpu.py -x -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" | tee -a $O/status.txt
tail -30 $O/tests.log
timeout -k 10 300 python tools/bench_gemm_shapes.py > $O/gemm_shapes.txt 2>&1
echo "gemm rc=$?" | tee -a $O/status.txt
timeout -k 10 300 python tools/profile_a2c.py > $O/profile_fused_graph.txt 2>&1
echo "profile rc=$?" | tee -a $O/status.txt
timeout -k 10 300 python tools/profile_a2c.py --eager-collect > $O/profile_fused_eager.txt 2>&1
echo "profile eager rc=$?" | tee -a $O/status.txt
timeout -k 10 300 python bench.py --mode a2c > $O/bench_a2c.json 2> $O/bench_a2c.err
echo "bench a2c rc=$?" | tee -a $O/status.txt
cat $O/bench_a2c.json
cat $O/status.txt
