#!/bin/bash
# whole GPU suite + default bench (round 3, after the GEMM work)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03p
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc" | tee -a $O/status.txt; tail -6 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc=$?" | tee -a $O/status.txt
python - <<'PY'
import json,os
d=json.loads(open(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r03p/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], d["roofline"].get("moved_frac"), d["roofline"].get("valu_issue_frac"))
print({k:d["a2c"][k] for k in ("value","ms_per_rollout","collect_ms_per_rollout","update_ms_per_rollout")})
PY
