#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/r04_pmc.sh r04u > /dev/null 2>&1
tail -30 gpurun_out/r04u/status.txt
OUT_TAG=r04v bash tools/r04_final_evidence.sh r04v
