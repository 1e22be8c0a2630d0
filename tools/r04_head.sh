#!/bin/bash
# actor head: tests, the head alone by row count and tile height, s_memtime stamps (ab_build/libuavagent_stamps.so = a -DUAVGEMM_STAMPS build), collection A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04hf}
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_learner_kernels_gpu.py tests/test_a2c_gpu.py tests/test_agent_kernel_gpu.py -q -x > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc" | tee -a $O/status.txt; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/bench_head.py > $O/bench_head.json 2>> $O/err.log
python -c "
import json;d=json.load(open('$O/bench_head.json'));print({k:min(v) for k,v in d.items()})"
UAVAGENT_LIB=$R/ab_build/libuavagent_stamps.so timeout -k 10 300 python tools/head_stamps.py > $O/head_stamps.json 2>> $O/err.log
python -c "
import json;d=json.load(open('$O/head_stamps.json'))
for k,v in d.items(): print(k, {a:int(b) for a,b in v['median_cycles'].items()})"
timeout -k 10 300 python tools/ab_collect.py default unsplit > $O/ab_collect.json 2>> $O/err.log; cat $O/ab_collect.json
