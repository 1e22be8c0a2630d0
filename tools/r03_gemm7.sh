#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for c in 0 1 2; do UAVGEMM_SMALL=$c timeout -k 10 200 python tools/bench_rollout_gemm.py 2>/dev/null; done
