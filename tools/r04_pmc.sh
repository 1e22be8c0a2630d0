#!/bin/bash
# round 4: kernel traces + PMC passes (separate passes: FETCH_SIZE / WRITE_SIZE / SQ) for every env workload bench.py's roofline block
# describes, each in the DISPATCH FORM the bench runs it in (prewarm_calls: bench.py's measure_env also runs 3 untimed calls of the same
# form on a scratch env, and repeats the timed steps once with uavenv_launch_timing on); writes a manifest (runs.json) with the step counts it used, which
# tools/make_traffic_json.py reads (no hard-coded counts there).   usage: r04_pmc.sh <out tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04e}
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
A="--no-cpu-baseline --no-a2c --no-alt"
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
echo '{"runs": [' > $O/runs.json
first=1
run() {   # tag, envs, n_bs, n_ue, many(1/0), steps per call, schedule, trace steps, pmc steps, [ENV=VAL ...] -- bench args...
  local tag=$1 envs=$2 nbs=$3 nue=$4 many=$5 spc=$6 sched=$7 ts=$8 ps=$9; shift 9
  local envv=()
  while [ "$1" != "--" ]; do envv+=("$1"); shift; done; shift
  [ $first -eq 1 ] || echo ',' >> $O/runs.json; first=0
  echo "{\"tag\": \"$tag\", \"envs\": $envs, \"n_bs\": $nbs, \"n_ue\": $nue, \"many\": $many, \"steps_per_call\": $spc, \"schedule\": \"$sched\", \"warmup\": 100, \"trace_steps\": $ts, \"pmc_steps\": $ps, \"prewarm_calls\": $((many * 3)), \"timed_steps_repeated_for_launch_timing\": $many}" >> $O/runs.json
  for kv in "${envv[@]}"; do export "$kv"; done
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$tag -- python3 $R/bench.py --steps $ts --warmup 100 $A "$@" > $O/trace_$tag.log 2>&1
  echo "trace $tag rc=$?" | tee -a $O/status.txt
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${tag}_$c -- python3 $R/bench.py --steps $ps --warmup 100 $A "$@" > $O/pmc_${tag}_$c.log 2>&1
    echo "pmc $tag $c rc=$?" | tee -a $O/status.txt
  done
  timeout -k 10 300 rocprofv3 --pmc $SQ --output-format csv -d $O/pmc_${tag}_sq -- python3 $R/bench.py --steps $ps --warmup 100 $A "$@" > $O/pmc_${tag}_sq.log 2>&1
  echo "pmc $tag sq rc=$?" | tee -a $O/status.txt
  for kv in "${envv[@]}"; do unset "${kv%%=*}"; done
}
run many100     4096 4 20 1 100 one_launch_rotation 2000 400 -- || exit 1
run many20      4096 4 20 1 20  one_launch_rotation 400 200 -- --chunk 20 || exit 1
run many20plain 4096 4 20 1 20  plain 400 200 UAVENV_ROTATE=0 -- --chunk 20 || exit 1
run seq         4096 4 20 0 1   plain 2000 400 -- --launch seq || exit 1
run seq8192     8192 4 20 0 1   plain 1000 300 -- --launch seq --envs 8192 || exit 1
run many65536   65536 4 20 1 100 plain 600 200 -- --envs 65536 || exit 1
run c5          8192 16 200 0 1 plain 300 100 -- --launch seq --n-bs 16 --n-ue 200 --envs 8192 || exit 1
echo ']}' >> $O/runs.json
cd $R
find $O -name "*_kernel_trace.csv" -size +3M -delete
find $O -name "*agent_info.csv" -delete
python3 tools/pmc_digest.py $O > $O/digest.txt 2>&1
grep -E "env_kernel|== " $O/digest.txt | cut -c1-220 | tail -80
cat $O/status.txt
