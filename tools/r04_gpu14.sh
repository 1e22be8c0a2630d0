#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04r
mkdir -p $O
cd $R
timeout -k 10 600 python tools/ab_collect.py default unsplit EAGER > $O/ab_collect.json 2> $O/ab_collect.err
rc=$?; echo "ab_collect rc=$rc" | tee -a $O/status.txt; python -c "
import json
for k,v in json.load(open('$O/ab_collect.json')).items(): print('%-50s'%k, v)"; tail -3 $O/ab_collect.err
