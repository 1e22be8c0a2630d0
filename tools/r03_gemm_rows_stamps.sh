#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03m
mkdir -p $O
cd $R
UAVAGENT_LIB=$R/ab_build/libuavagent_stamps.so timeout -k 10 300 python tools/gemm_rows_stamps.py > $O/rows_stamps.json 2> $O/rows_stamps.err; echo "rc=$?"
cat $O/rows_stamps.json; tail -3 $O/rows_stamps.err
