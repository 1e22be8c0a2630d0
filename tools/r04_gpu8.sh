#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04i
mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_learner_kernels_gpu.py -q -x -k "graph_captured" > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc" | tee -a $O/status.txt; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tools/ab_collect.py default unsplit pipelined > $O/ab_collect.json 2> $O/ab_collect.err
rc=$?; echo "ab_collect rc=$rc" | tee -a $O/status.txt; python -c "
import json
for k,v in json.load(open('$O/ab_collect.json')).items(): print('%-50s'%k, v)"; tail -3 $O/ab_collect.err
