#!/bin/bash
# round 4, final tree (persistent rollout kernels): GPU suite, default + driver-style bench lines, collection A/B, the two kernels alone and as a
# pair, 150 training episodes.  Results are copied to profiles/r04fd_* by hand.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04fd}
mkdir -p $O
cd $R
OUT_TAG=${1:-r04fd} bash tools/r04_final_evidence.sh ${1:-r04fd} || exit 1
timeout -k 10 300 python3 tools/ab_collect.py persistent default unsplit > $O/ab_collect.json 2> $O/ab_collect.err
echo "ab rc=$?" | tee -a $O/status.txt; cat $O/ab_collect.json
timeout -k 10 200 python3 tools/gated_probe.py > $O/gated_probe.json 2> $O/gated_probe.err
echo "probe rc=$?" | tee -a $O/status.txt; cat $O/gated_probe.json
timeout -k 10 400 python tools/train_a2c.py --out $O/run --workers 8192 --episodes 150 > $O/train.log 2> $O/train.err
echo "train rc=$?" | tee -a $O/status.txt; tail -1 $O/train.log | cut -c1-250
rm -f $O/run/*.npz $O/run/*.pt
