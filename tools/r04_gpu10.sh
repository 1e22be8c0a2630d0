#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04k
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_step_many_gpu.py tests/test_full_size_parity_gpu.py tests/test_learner_kernels_gpu.py -q -x -k "rotation or hand_off or step_many or loss_grad or fused_update or hip_gemms" > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc" | tee -a $O/status.txt; tail -4 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/r04_many_ab.py 4096:20 4096:100 4096:16 5400:100 5900:100 8192:100 9000:100 > $O/many_ab.json 2> $O/many_ab.err
rc=$?; echo "many_ab rc=$rc" | tee -a $O/status.txt; python -c "
import json;d=json.load(open('$O/many_ab.json'));print(d['best_us_per_step'])"
bash tools/r04_pmc.sh r04e
