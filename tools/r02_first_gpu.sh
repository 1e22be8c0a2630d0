#!/bin/bash
# round 2, first GPU call: new tests, bench in all launch forms, kernel trace + PMC passes for the 4x20 step kernel and config 5
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02a
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 900 python -m pytest tests/test_step_many_gpu.py tests/test_full_size_parity_gpu.py tests/test_hardening_gpu.py tests/test_bench_launcher.py tests/test_agent_kernel_gpu.py -x -q -m gpu > $O/tests_new.log 2>&1
echo "new tests rc=$?" | tee -a $O/status.txt
tail -5 $O/tests_new.log
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench default rc=$?" | tee -a $O/status.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-a2c > $O/bench_driver_style.json 2> $O/bench_driver_style.err
echo "bench driver-style rc=$?" | tee -a $O/status.txt
timeout -k 10 300 python bench.py --envs 65536 --no-cpu-baseline --no-a2c > $O/bench_65536.json 2> $O/bench_65536.err
echo "bench 65536 rc=$?" | tee -a $O/status.txt
timeout -k 10 300 python bench.py --n-bs 16 --n-ue 200 --envs 8192 --steps 300 --warmup 30 --no-cpu-baseline --no-a2c > $O/bench_config5.json 2> $O/bench_config5.err
echo "bench config5 rc=$?" | tee -a $O/status.txt
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-a2c --no-alt > $O/trace.log 2>&1
echo "trace rc=$?" | tee -a $O/status.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_many -- python3 $R/bench.py --steps 1000 --warmup 100 --launch many --no-cpu-baseline --no-a2c --no-alt > $O/trace_many.log 2>&1
echo "trace many rc=$?" | tee -a $O/status.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 100 --warmup 20 --launch eager --no-cpu-baseline --no-a2c --no-alt > $O/pmc_$c.log 2>&1
  echo "pmc $c rc=$?" | tee -a $O/status.txt
done
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --steps 100 --warmup 20 --launch eager --no-cpu-baseline --no-a2c --no-alt > $O/pmc_sq.log 2>&1
echo "pmc sq rc=$?" | tee -a $O/status.txt
# config 5
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5_trace -- python3 $R/bench.py --n-bs 16 --n-ue 200 --envs 8192 --steps 200 --warmup 20 --launch eager --no-cpu-baseline --no-a2c --no-alt > $O/c5_trace.log 2>&1
echo "c5 trace rc=$?" | tee -a $O/status.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/c5_pmc_$c -- python3 $R/bench.py --n-bs 16 --n-ue 200 --envs 8192 --steps 40 --warmup 10 --launch eager --no-cpu-baseline --no-a2c --no-alt > $O/c5_pmc_$c.log 2>&1
  echo "c5 pmc $c rc=$?" | tee -a $O/status.txt
done
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/c5_pmc_sq -- python3 $R/bench.py --n-bs 16 --n-ue 200 --envs 8192 --steps 40 --warmup 10 --launch eager --no-cpu-baseline --no-a2c --no-alt > $O/c5_pmc_sq.log 2>&1
echo "c5 pmc sq rc=$?" | tee -a $O/status.txt
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/c5_pmc_sq2 -- python3 $R/bench.py --n-bs 16 --n-ue 200 --envs 8192 --steps 40 --warmup 10 --launch eager --no-cpu-baseline --no-a2c --no-alt > $O/c5_pmc_sq2.log 2>&1
echo "c5 pmc sq2 rc=$?" | tee -a $O/status.txt
cd $R
# keep only the small csv summaries (the merge-back limit is 64 MiB)
find $O -name "*_kernel_trace.csv" -size +5M -delete
find $O -name "*.db" -delete
python3 tools/pmc_digest.py $O > $O/digest.txt 2>&1
cat $O/status.txt
