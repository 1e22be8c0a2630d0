#!/bin/bash
# round 3: kernel traces + PMC passes (separate passes: FETCH_SIZE / WRITE_SIZE / SQ) for the four env workloads bench.py's roofline
# block describes; tools/make_traffic_json.py turns the digest into profiles/traffic_current.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03e
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
A="--no-cpu-baseline --no-a2c --no-alt"
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
run() {   # tag, trace steps, pmc steps, bench args...
  local tag=$1 ts=$2 ps=$3; shift 3
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$tag -- python3 $R/bench.py --steps $ts --warmup 100 $A "$@" > $O/trace_$tag.log 2>&1
  echo "trace $tag rc=$?" | tee -a $O/status.txt
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${tag}_$c -- python3 $R/bench.py --steps $ps --warmup 100 $A "$@" > $O/pmc_${tag}_$c.log 2>&1
    echo "pmc $tag $c rc=$?" | tee -a $O/status.txt
  done
  timeout -k 10 300 rocprofv3 --pmc $SQ --output-format csv -d $O/pmc_${tag}_sq -- python3 $R/bench.py --steps $ps --warmup 100 $A "$@" > $O/pmc_${tag}_sq.log 2>&1
  echo "pmc $tag sq rc=$?" | tee -a $O/status.txt
}
run many 2000 400 || exit 1
run seq 2000 400 --launch seq || exit 1
run many65536 600 200 --envs 65536 || exit 1
run c5 300 100 --launch seq --n-bs 16 --n-ue 200 --envs 8192 || exit 1
cd $R
find $O -name "*_kernel_trace.csv" -size +3M -delete
find $O -name "*agent_info.csv" -delete
python3 tools/pmc_digest.py $O > $O/digest.txt 2>&1
grep -E "env_kernel|== " $O/digest.txt | cut -c1-220 | tail -60
cat $O/status.txt
