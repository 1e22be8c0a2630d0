#!/bin/bash
# CPU sanitizer pass over the oracle (the only native code that runs without a GPU; GPU ASan is not available on this pool):
# builds oracle/uavenv_oracle.c with -fsanitize=address,undefined into a temp copy, runs the golden-vector and Philox tests on it
# under LD_PRELOAD, then restores the normal build.   bash tools/asan_oracle.sh
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cp $R/oracle/libuavenv_oracle.so /tmp/uavenv_oracle_backup.so
trap 'cp /tmp/uavenv_oracle_backup.so $R/oracle/libuavenv_oracle.so' EXIT
gcc -O1 -g -std=c99 -fPIC -ffp-contract=off -fno-fast-math -fsanitize=address,undefined -fno-omit-frame-pointer -shared \
    -o $R/oracle/libuavenv_oracle.so $R/oracle/uavenv_oracle.c -lm
cd $R
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
    python -m pytest tests/test_oracle_golden.py tests/test_philox.py -x -q
