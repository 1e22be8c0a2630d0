#!/usr/bin/env python3
"""A few whole A2C iterations (collect + update) at BASELINE config 3 for rocprofv3 (tools/r03_pmc_a2c.sh)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drl_uav_cellularnet_amd import BatchedMobiEnv
from drl_uav_cellularnet_amd.agent import A2CRunner
env = BatchedMobiEnv(8192, nBS=4, nUE=20, grid_n=100, groups=[5, 5, 5, 5])
r = A2CRunner(env, rollout=50, collect_launch=os.environ.get("COLLECT", "graph"), pipeline_halves=os.environ.get("PIPE", "0") == "1")
for _ in range(int(os.environ.get("REPS", "4"))):
    r.train_rollout()
torch.cuda.synchronize()
print(r.stats)
