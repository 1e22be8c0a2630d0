#!/usr/bin/env python3
"""LTEChannel.GetSinrInArea (channel.py:411-433) as the batched HIP kernel: coverage maps per second (secondary measurement).
The reference computes one (G-1)^2 map with (G-1)^2 * B Python-level GetChannelGain calls; main_test.py:85-89 does it every 500 steps."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from drl_uav_cellularnet_amd import BatchedMobiEnv

out = {}
for n_envs, n_bs in ((1, 4), (256, 4), (4096, 4), (256, 16)):
    env = BatchedMobiEnv(n_envs, nBS=n_bs, nUE=20 if n_bs == 4 else 60, grid_n=100)
    for _ in range(3):
        m = env.sinr_area()
    torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        m = env.sinr_area()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    cells = n_envs * 99 * 99
    out["%d envs x %d UAV" % (n_envs, n_bs)] = {"ms_per_call": dt * 1e3, "maps_per_s": n_envs / dt, "cell_uav_pairs_per_s": cells * n_bs / dt}
print(json.dumps({"metric": "GetSinrInArea coverage maps (G = 100, float32 output)", **out}))
