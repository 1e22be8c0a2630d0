#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04hs}
mkdir -p $O
cd $R
UAVAGENT_LIB=$R/ab_build/libuavagent_stamps.so timeout -k 10 300 python tools/head_stamps.py > $O/head_stamps.json 2> $O/err.log
echo rc=$?; cat $O/head_stamps.json; tail -3 $O/err.log
