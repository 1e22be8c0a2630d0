#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04su}
mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -q -m gpu -x -rs > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc" | tee -a $O/status.txt; tail -5 $O/tests.log
