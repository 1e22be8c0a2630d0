#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02g
mkdir -p $O
cd $R
timeout -k 10 600 python bench.py --mode a2c > $O/bench_a2c_tuned.json 2> $O/bench_a2c_tuned.err
echo "bench a2c rc=$?" | tee -a $O/status.txt
cat $O/bench_a2c_tuned.json
ls -la /tmp/uavagent_tunable_*/ 2>/dev/null
cat /tmp/uavagent_tunable_*/tunableop_results*.csv > $O/tunableop_after_a2c.csv 2>/dev/null
timeout -k 10 600 python -m pytest tests/test_learner_kernels_gpu.py tests/test_a2c_gpu.py tests/test_a2c_two_ranks_gpu.py -x -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" | tee -a $O/status.txt
tail -5 $O/tests.log
timeout -k 10 300 python tools/profile_a2c.py > $O/profile_tuned.txt 2>&1
cat $O/status.txt
