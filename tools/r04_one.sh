#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04one
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest "$@" -q -x > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -15 $O/tests.log
exit $rc
