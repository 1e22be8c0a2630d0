#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03n
mkdir -p $O
cd $R
timeout -k 10 400 python tools/profile_a2c.py > $O/a2c_profile.txt 2>&1; echo "rc=$?"
grep -E "Self CUDA time|^void|^Cijk|uavk|Name" $O/a2c_profile.txt | cut -c1-60,150-260 | head -60
