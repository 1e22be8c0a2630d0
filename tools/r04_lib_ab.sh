#!/bin/bash
# A/B of two libuavenv builds on one box, processes alternated: the shipped library against ab_build/$1 (tools/r04_many_ab.py shapes in $2...)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04ab
mkdir -p $O
cd $R
ALT=$1; shift
for rep in 1 2 3; do
  timeout -k 10 200 python tools/r04_many_ab.py "$@" > $O/ab_shipped_$rep.json 2>> $O/err.log
  UAVENV_LIB=$R/ab_build/$ALT timeout -k 10 200 python tools/r04_many_ab.py "$@" > $O/ab_alt_$rep.json 2>> $O/err.log
done
python - <<'PY'
import json,os
R=os.environ.get("GRAFT_REPO_ROOT",".")
for rep in (1,2,3):
    for f in ("ab_shipped_%d"%rep,"ab_alt_%d"%rep):
        d=json.load(open(R+"/gpurun_out/r04ab/%s.json"%f)); print(f, {k:min(v) for k,v in d["us_per_call"].items()})
PY
