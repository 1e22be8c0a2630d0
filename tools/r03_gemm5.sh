#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03l
mkdir -p $O
cd $R
timeout -k 10 400 python tools/bench_gemm_hip.py --rounds 3 > $O/gemm_bm128.json 2> $O/gemm.err; echo "rc=$?"
UAVGEMM_BM=64 timeout -k 10 400 python tools/bench_gemm_hip.py --rounds 3 > $O/gemm_bm64.json 2> $O/gemm.err; echo "rc=$?"
