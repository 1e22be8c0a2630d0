#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03i
mkdir -p $O
cd $R
timeout -k 10 400 python tools/bench_gemm_hip.py --tune --rounds 3 > $O/gemm.json 2> $O/gemm.err
echo "gemm rc=$?" | tee -a $O/status.txt
for d in 0 2; do
UAVGEMM_DBG=$d UAVAGENT_LIB=$R/ab_build/libuavagent_stamps.so timeout -k 10 300 python tools/gemm_stamps.py > $O/stamps_$d.json 2> $O/stamps_$d.err; echo "stamps $d rc=$?"
cat $O/stamps_$d.json
done
