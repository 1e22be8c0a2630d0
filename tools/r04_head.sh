#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04hd}
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_learner_kernels_gpu.py -q -x -k "actor_head or graph_captured or index_lists" > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc" | tee -a $O/status.txt; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/bench_head.py > $O/bench_head.json 2> $O/bench_head.err; cat $O/bench_head.json
timeout -k 10 300 python tools/ab_collect.py default unsplit > $O/ab_collect.json 2> $O/ab_collect.err; cat $O/ab_collect.json
