#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04o
mkdir -p $O
cd $R
timeout -k 10 200 python tools/region_overhead.py > $O/region_overhead.json 2> $O/err.log; cat $O/region_overhead.json; tail -2 $O/err.log
