#!/bin/bash
# round 4: PMC passes of the two persistent rollout kernels, each ALONE with open gates (tools/gated_probe.py, SKIP_PAIR=1): a --pmc run
# serialises dispatches, so the pair itself cannot be counted
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04pc}
mkdir -p $O
export TMPDIR=/tmp SKIP_PAIR=1
cd /tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$tag -- python3 $R/tools/gated_probe.py > $O/pmc_$tag.log 2>&1
  echo "pmc $tag rc=$?" | tee -a $O/status.txt
done
cd $R
find $O -name "*agent_info.csv" -delete
python3 tools/pmc_digest.py $O > $O/digest.txt 2>&1
grep -E "gated|== " $O/digest.txt | cut -c1-200
