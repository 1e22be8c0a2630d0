#!/bin/bash
# what the release / acquire fences of a hand-off cost: the shipped library against a build without them (timing only: WRONG results)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04l
mkdir -p $O
cd $R
for rep in 1 2; do
  timeout -k 10 200 python tools/r04_many_ab.py 4096:20 4096:100 > $O/ab_shipped_$rep.json 2>> $O/err.log
  UAVENV_LIB=$R/ab_build/libuavenv_nofence.so timeout -k 10 200 python tools/r04_many_ab.py 4096:20 4096:100 > $O/ab_nofence_$rep.json 2>> $O/err.log
done
python - <<'PY'
import json,os
R=os.environ.get("GRAFT_REPO_ROOT",".")
for f in ("ab_shipped_1","ab_nofence_1","ab_shipped_2","ab_nofence_2"):
    d=json.load(open(R+"/gpurun_out/r04l/%s.json"%f)); print(f, {k:min(v) for k,v in d["us_per_call"].items()})
PY
