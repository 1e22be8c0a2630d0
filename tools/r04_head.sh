#!/bin/bash
# actor head: correctness, then the head alone and the collection with / without the LDS-DMA stagger (UAVAGENT_HEAD_STAGGER=0: every wave issues
# at the chunk's start), processes alternated on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04hd}
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_learner_kernels_gpu.py -q -x -k "actor_head or graph_captured or index_lists" > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc" | tee -a $O/status.txt; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
  timeout -k 10 300 python tools/bench_head.py > $O/bench_head_stagger_$rep.json 2>> $O/err.log
  UAVAGENT_HEAD_STAGGER=0 timeout -k 10 300 python tools/bench_head.py > $O/bench_head_nostagger_$rep.json 2>> $O/err.log
done
python - <<'PY'
import json,os
R=os.environ.get("GRAFT_REPO_ROOT",".")
import glob
for f in sorted(glob.glob(R+"/gpurun_out/*/bench_head_*stagger_*.json"))[-4:]:
    d=json.load(open(f)); print(os.path.basename(f), {k:min(v) for k,v in d.items()})
PY
timeout -k 10 300 python tools/ab_collect.py default unsplit > $O/ab_collect_stagger.json 2>> $O/err.log; cat $O/ab_collect_stagger.json
UAVAGENT_HEAD_STAGGER=0 timeout -k 10 300 python tools/ab_collect.py default unsplit > $O/ab_collect_nostagger.json 2>> $O/err.log; cat $O/ab_collect_nostagger.json
