#!/bin/bash
# round 4: RCCL at one rank, the bench's new blocks (a2c.roofline, schedule-keyed traffic), driver-style 20-step line
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04d
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_rccl_one_rank_gpu.py tests/test_a2c_two_ranks_gpu.py -q -x -rs > $O/tests_rccl.log 2>&1
rc=$?; echo "rccl tests rc=$rc" | tee -a $O/status.txt; tail -6 $O/tests_rccl.log
[ -f gpurun_out/rccl_one_rank_error.txt ] && cp gpurun_out/rccl_one_rank_error.txt $O/
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
rc=$?; echo "bench default rc=$rc" | tee -a $O/status.txt; tail -3 $O/bench_default.err
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err
rc=$?; echo "bench 20 rc=$rc" | tee -a $O/status.txt
python - <<'PY'
import json,os
R=os.environ.get("GRAFT_REPO_ROOT",".")
for f in ("bench_default","bench_driver_style"):
    d=json.loads(open(R+"/gpurun_out/r04d/%s.json"%f).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(f, "value %.4g"%d["value"], "frac %.3f"%r["frac"], r.get("schedule"), r.get("traffic"), r.get("traffic_unavailable_because"), "single-step %.4g"%d.get("single_step_launch_value",0))
    a=d.get("a2c") or {}
    print({k:a.get(k) for k in ("value","ms_per_rollout","collect_ms_per_rollout","update_ms_per_rollout")})
    rf=(a.get("roofline") or {})
    if "error" in rf: print("a2c roofline error", rf["error"])
    for k,v in (rf.get("kernels") or {}).items():
        print("  %-22s %8.1f us  %7.2f %s  frac %.3f  (%d launches)"%(k,v["avg_us"],v["achieved"],v["unit"],v["frac"],v["launches_timed"]))
    print(" covered ms", rf.get("ms_per_rollout_covered"))
PY
