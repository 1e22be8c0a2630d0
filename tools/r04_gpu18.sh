#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04x
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_step_many_gpu.py tests/test_full_size_parity_gpu.py tests/test_launch_variants_gpu.py -q -x > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc" | tee -a $O/status.txt; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/r04_many_ab.py 4096:20 4096:100 8192:20 8192:100 > $O/many_ab.json 2> $O/many_ab.err
python -c "
import json;d=json.load(open('$O/many_ab.json'));print({k:min(v) for k,v in d['us_per_call'].items()})"
