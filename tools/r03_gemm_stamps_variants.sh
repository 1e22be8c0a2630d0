#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03j
mkdir -p $O
cd $R
for d in 0 1 8 2 4; do
UAVGEMM_DBG=$d UAVAGENT_LIB=$R/ab_build/libuavagent_stamps.so timeout -k 10 300 python tools/gemm_stamps.py > $O/stamps_$d.json 2> $O/stamps_$d.err; echo "dbg $d rc=$?"
cat $O/stamps_$d.json
done
