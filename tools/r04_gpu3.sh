#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04c
mkdir -p $O
cd $R
timeout -k 10 300 python tools/ab_collect.py default unsplit stagger > $O/ab_collect.json 2> $O/ab_collect.err
rc=$?; echo "ab_collect rc=$rc" | tee -a $O/status.txt; cat $O/ab_collect.json; tail -3 $O/ab_collect.err
