#!/bin/bash
# usage (GPU box, repo root): bash tools/pin_sweep.sh  -- pinned vs unpinned step kernel over batch sizes, interleaved, 2 rounds
set -e
for round in 1 2; do
  for envs in 2048 3072 4096 6144 8192 12288; do
    for pin in 1 0; do
      UAVENV_FORCE_PIN=$pin python3 bench.py --steps 1500 --warmup 100 --no-cpu-baseline --envs $envs 2>/dev/null | tail -1 | \
        python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('round $round  envs %6d  pin=$pin  %.3f us  %.1f M env-steps/s' % ($envs, d['roofline']['avg_launch_us'], d['value']/1e6))"
    done
  done
done
