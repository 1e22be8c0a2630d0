#!/bin/bash
# whole GPU suite, default bench line, driver-style 20-step line (round 4 evidence; copied to profiles/ by hand)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04p}
mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -q -m gpu -x -rs > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc" | tee -a $O/status.txt; tail -4 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc=$?" | tee -a $O/status.txt
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err
echo "bench 20 rc=$?" | tee -a $O/status.txt
python - <<'PY'
import json,os,sys
R=os.environ.get("GRAFT_REPO_ROOT",".")
O=[a for a in os.listdir(R+"/gpurun_out") if a.startswith("r04")]
for f in ("bench_default","bench_driver_style"):
    p=R+"/gpurun_out/%s/%s.json"%(os.environ.get("OUT_TAG","r04p"),f)
    d=json.loads(open(p).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f,"value %.4g"%d["value"],"frac %.3f"%r["frac"],"moved_frac",r.get("moved_frac"),"valu",r.get("valu_issue_frac"),"launch us",r["avg_launch_us"],"single %.4g"%d.get("single_step_launch_value",0))
    a=d.get("a2c") or {}
    print("  a2c",{k:a.get(k) for k in ("value","ms_per_rollout","collect_ms_per_rollout","update_ms_per_rollout")})
    print("  cpu",(d.get("cpu_baseline") or {}).get("value"), ((d.get("cpu_baseline") or {}).get("single_core_n1") or {}).get("value"))
    print("  other",{k:"%.3g"%v["value"] for k,v in (d.get("other_launch_forms") or {}).items()}, {k:"%.3g"%v["value"] for k,v in (d.get("other_grids") or {}).items()})
PY
