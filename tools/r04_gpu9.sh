#!/bin/bash
# round 4: whole GPU suite, then update A/B (loss kernel), then bench default + driver style
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04j
mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -q -m gpu -x > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc" | tee -a $O/status.txt; tail -6 $O/tests.log
[ $rc -eq 0 ] || exit $rc
UAVAGENT_LOSS_SCALAR=1 timeout -k 10 300 python bench.py --mode a2c > $O/bench_a2c_scalar_loss.json 2> $O/bench_a2c_scalar_loss.err
timeout -k 10 300 python bench.py --mode a2c > $O/bench_a2c.json 2> $O/bench_a2c.err
rc=$?; echo "bench a2c rc=$rc" | tee -a $O/status.txt
python - <<'PY'
import json,os
R=os.environ.get("GRAFT_REPO_ROOT",".")
for f in ("bench_a2c_scalar_loss","bench_a2c"):
    a=json.loads(open(R+"/gpurun_out/r04j/%s.json"%f).read().strip().splitlines()[-1])["a2c"]
    print(f,{k:a.get(k) for k in ("value","ms_per_rollout","collect_ms_per_rollout","update_ms_per_rollout","pipeline_halves")})
    rf=(a.get("roofline") or {})
    if "error" in rf: print("a2c roofline error", rf["error"])
    for k,v in (rf.get("kernels") or {}).items():
        print("  %-22s %8.1f us  %7.2f %s  frac %.3f  (%d launches)"%(k,v["avg_us"],v["achieved"],v["unit"],v["frac"],v["launches_timed"]))
PY
