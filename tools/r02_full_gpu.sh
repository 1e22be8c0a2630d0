#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02d
mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" | tee -a $O/status.txt
tail -8 $O/tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc=$?" | tee -a $O/status.txt
tail -2 $O/smoke.log
timeout -k 10 300 python tools/bench_shim.py > $O/bench_shim.json 2> $O/bench_shim.err
echo "shim rc=$?" | tee -a $O/status.txt
cat $O/bench_shim.json
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench default rc=$?" | tee -a $O/status.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err
echo "bench driver-style rc=$?" | tee -a $O/status.txt
cat $O/status.txt
