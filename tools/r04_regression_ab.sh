#!/bin/bash
# same-box A/B of the paths round 4 did NOT mean to change: the round-3 tree (ab_build/r03tree, its own bench.py and libraries) against
# this tree, alternated: 65 536 envs (unpinned multi-step kernel, plain launch), one kernel per step at 4096 / 8192 envs, config 5
# Set-up in the build container (ab_build/ is git-ignored but travels to the GPU box):
#   mkdir -p ab_build/r03tree && git archive bb80d09 drl_uav_cellularnet_amd include bench.py oracle tools profiles/traffic_current.json | tar -x -C ab_build/r03tree
#   (cd ab_build/r03tree && python -c "import sys; sys.path.insert(0, '.'); from drl_uav_cellularnet_amd import build as b; b.build(force=True)" && make -C oracle -s -B)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04s
mkdir -p $O
A="--no-a2c --no-alt --no-cpu-baseline"
run() { # tag dir args...
  local tag=$1 dir=$2; shift 2
  (cd $dir && timeout -k 10 300 python bench.py $A "$@" 2>> $O/err.log | tail -1 > $O/$tag.json)
  python -c "
import json;d=json.loads(open('$O/$tag.json').read());print('$tag', '%.4g'%d['value'], 'us/step gpu', d['roofline']['avg_step_us'])"
}
for rep in 1 2; do
  run r03_65536_$rep $R/ab_build/r03tree --envs 65536 --steps 600
  run r04_65536_$rep $R --envs 65536 --steps 600
  run r03_seq4096_$rep $R/ab_build/r03tree --launch seq
  run r04_seq4096_$rep $R --launch seq
  run r03_seq8192_$rep $R/ab_build/r03tree --launch seq --envs 8192
  run r04_seq8192_$rep $R --launch seq --envs 8192
  run r03_c5_$rep $R/ab_build/r03tree --launch seq --n-bs 16 --n-ue 200 --envs 8192 --steps 300
  run r04_c5_$rep $R --launch seq --n-bs 16 --n-ue 200 --envs 8192 --steps 300
done
