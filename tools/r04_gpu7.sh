#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04h
mkdir -p $O
cd $R
timeout -k 10 300 python tools/ab_collect.py default unsplit > $O/ab_collect.json 2> $O/ab_collect.err
rc=$?; echo "ab_collect rc=$rc" | tee -a $O/status.txt; cat $O/ab_collect.json; tail -3 $O/ab_collect.err
sed -i 's#gpurun_out/r04g#gpurun_out/r04h#g; s#O=$R/gpurun_out/r04g#O=$R/gpurun_out/r04h#' tools/r04_gpu6.sh
bash tools/r04_gpu6.sh
