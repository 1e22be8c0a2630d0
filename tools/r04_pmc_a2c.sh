#!/bin/bash
# round 4: the learner's kernels over whole A2C iterations on the final build (8192 envs x 50 steps, rollout pipelined over two streams):
# kernel trace + separate PMC passes (FETCH_SIZE / WRITE_SIZE / L2 hit counters), digest by tools/pmc_digest.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04pa}
mkdir -p $O
export TMPDIR=/tmp
export PIPE=1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_a2c -- python3 $R/tools/prof_a2c_run.py > $O/trace_a2c.log 2>&1
echo "trace rc=$?" | tee -a $O/status.txt
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_a2c_$tag -- python3 $R/tools/prof_a2c_run.py > $O/pmc_a2c_$tag.log 2>&1
  echo "pmc $tag rc=$?" | tee -a $O/status.txt
done
cd $R
find $O -name "*_kernel_trace.csv" -size +3M -delete
find $O -name "*agent_info.csv" -delete
python3 tools/pmc_digest.py $O > $O/digest.txt 2>&1
grep -E "rows_sum|sparse_rows|actor_head|gemm_|a2c_loss|env_kernel|== " $O/digest.txt | cut -c1-200 | tail -80
cat $O/status.txt
