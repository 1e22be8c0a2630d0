#!/usr/bin/env python3
"""Where does an A2C rollout+update spend its time?  (torch profiler, device time by op; secondary tool)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

from drl_uav_cellularnet_amd import BatchedMobiEnv
from drl_uav_cellularnet_amd.agent import A2CRunner

launch = "eager" if "--eager-collect" in sys.argv else "graph"
fused = "--autograd-update" not in sys.argv
env = BatchedMobiEnv(8192, nBS=4, nUE=20, grid_n=100)
runner = A2CRunner(env, rollout=50, collect_launch=launch, fused_update=fused)
print("collect_launch=%s fused_update=%s" % (launch, fused))
for _ in range(2):
    runner.train_rollout()
torch.cuda.synchronize()
for name, fn in (("collect", lambda: runner.collect()), ("update", None)):
    if fn is None:
        batch = runner.collect()
        torch.cuda.synchronize()
        fn = lambda: runner.update(*batch)
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    print("=== %s: top device-time ops" % name)
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=60))
