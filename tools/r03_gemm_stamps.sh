#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03h
mkdir -p $O
cd $R
timeout -k 10 300 python tools/gemm_stamps.py > $O/plain.json 2> $O/plain.err; echo "plain rc=$?"
UAVAGENT_LIB=$R/ab_build/libuavagent_stamps.so timeout -k 10 300 python tools/gemm_stamps.py > $O/stamps.json 2> $O/stamps.err; echo "stamps rc=$?"
cat $O/plain.json $O/stamps.json; tail -3 $O/stamps.err
