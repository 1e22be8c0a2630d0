#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03f
mkdir -p $O
cd $R
timeout -k 10 400 python tools/bench_gemm_hip.py --tune --rounds 3 > $O/gemm.json 2> $O/gemm.err
echo "gemm rc=$?" | tee -a $O/status.txt
tail -5 $O/gemm.err
cat $O/gemm.json
