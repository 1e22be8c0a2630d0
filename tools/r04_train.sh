#!/bin/bash
# round 4: the reference's training loop on the final build (pipelined rollout, float4 loss kernel): 150 episodes of 8192 workers, 4 UAV x 40 UE
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04tr}
mkdir -p $O
cd $R
timeout -k 10 900 python tools/train_a2c.py --out $O/run --workers 8192 --episodes ${2:-150} > $O/train.log 2> $O/train.err
echo "train rc=$?" | tee -a $O/status.txt
tail -3 $O/train.log; ls $O/run | head
