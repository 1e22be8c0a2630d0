#!/bin/bash
# usage (GPU box, repo root): bash tools/prof_sq.sh <tag> [bench args] -- SQ counter passes for the step kernel
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$TAG
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/$TAG/sq1 -- python3 $R/bench.py --steps 60 --warmup 20 --no-cpu-baseline "$@" > $R/gpurun_out/$TAG/sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/$TAG/sq2 -- python3 $R/bench.py --steps 60 --warmup 20 --no-cpu-baseline "$@" > $R/gpurun_out/$TAG/sq2.log 2>&1
cd $R
python3 - "$R/gpurun_out/$TAG" <<'PY'
import csv, glob, os, sys
d = sys.argv[1]
for sub in ("sq1", "sq2"):
    fs = glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        print(sub, "no counter file; log tail:"); print(open(os.path.join(d, sub + ".log")).read()[-1500:]); continue
    acc = {}
    for r in csv.DictReader(open(fs[0])):
        if "env_kernel" not in r["Kernel_Name"] or ", 2, " not in r["Kernel_Name"]:
            continue
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print("%-22s n=%d avg=%.0f" % (k, len(v), sum(v) / len(v)))
PY
