#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02e
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
A="--no-cpu-baseline --no-a2c --no-alt"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_many -- python3 $R/bench.py --steps 2000 --warmup 100 $A > $O/trace_many.log 2>&1
echo "trace many rc=$?" | tee -a $O/status.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_many_$c -- python3 $R/bench.py --steps 400 --warmup 100 $A > $O/pmc_many_$c.log 2>&1
  echo "pmc many $c rc=$?" | tee -a $O/status.txt
done
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_many_sq -- python3 $R/bench.py --steps 400 --warmup 100 $A > $O/pmc_many_sq.log 2>&1
echo "pmc many sq rc=$?" | tee -a $O/status.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_seq -- python3 $R/bench.py --launch seq --steps 2000 --warmup 100 $A > $O/trace_seq.log 2>&1
echo "trace seq rc=$?" | tee -a $O/status.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c5 -- python3 $R/bench.py --launch seq --n-bs 16 --n-ue 200 --envs 8192 --steps 300 --warmup 30 $A > $O/trace_c5.log 2>&1
echo "trace c5 rc=$?" | tee -a $O/status.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_c5_$c -- python3 $R/bench.py --launch seq --n-bs 16 --n-ue 200 --envs 8192 --steps 40 --warmup 10 $A > $O/pmc_c5_$c.log 2>&1
  echo "pmc c5 $c rc=$?" | tee -a $O/status.txt
done
cd $R
find $O -name "*_kernel_trace.csv" -size +5M -delete
python3 tools/pmc_digest.py $O > $O/digest.txt 2>&1
grep -E "packed<4, 2, true, true, true, true>|multipass<16, 2|== " $O/digest.txt | cut -c1-200
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
cat $O/status.txt
