#!/bin/bash
# round 3: whole GPU suite, default bench line, step_many A/B (nine arrays / packed / rotated)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03d
mkdir -p $O
cd $R
timeout -k 10 800 python -m pytest tests -q -m gpu -x > $O/tests.log 2>&1
rc=$?
echo "tests rc=$rc" | tee -a $O/status.txt
tail -8 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc=$?" | tee -a $O/status.txt
tail -c 3000 $O/bench_default.json
ROUNDS=3 timeout -k 10 200 python tools/r03_many_ab.py > $O/many_ab.json 2> $O/many_ab.err
echo "many_ab rc=$?" | tee -a $O/status.txt
cat $O/many_ab.json
