#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/prof.sh <tag> [bench args]
# kernel trace + stats, then two PMC passes (FETCH_SIZE, WRITE_SIZE) -- separate runs, as the guide prescribes.
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$TAG
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/trace -- python3 $R/bench.py --steps 1000 --warmup 100 --no-cpu-baseline "$@" > $R/gpurun_out/$TAG/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc_fetch -- python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline "$@" > $R/gpurun_out/$TAG/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc_write -- python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline "$@" > $R/gpurun_out/$TAG/bench_pmc_write.log 2>&1
cd $R
python3 tools/prof_summary.py gpurun_out/$TAG > gpurun_out/$TAG/summary.txt 2>&1 || true
cat gpurun_out/$TAG/summary.txt
