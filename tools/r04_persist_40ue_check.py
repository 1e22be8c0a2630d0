#!/usr/bin/env python3
"""Ad-hoc: A2CRunner with the persistent rollout against the per-step launches at 4 UAV x 40 UE (the training script's shape)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drl_uav_cellularnet_amd import BatchedMobiEnv
from drl_uav_cellularnet_amd.agent import A2CRunner

N, T = int(os.environ.get("N", 512)), int(os.environ.get("T", 6))
env1 = BatchedMobiEnv(N, nBS=4, nUE=40, grid_n=100, max_step=15)
env2 = env1.clone()
fs = os.environ.get("FIRST", "obs")
r1 = A2CRunner(env1, rollout=T, persistent_rollout=True, collect_launch=os.environ.get("LAUNCH", "graph"), first_state=fs)
r2 = A2CRunner(env2, rollout=T, persistent_rollout=False, pipeline_halves=False, collect_launch="eager", first_state=fs)
print("persistent:", r1._persistent)
for it in range(5):
    b1, b2 = r1.collect(), r2.collect()
    for name, x, y in zip(("idx", "act", "rew", "boot"), b1, b2):
        if not torch.equal(x, y):
            d = (x != y)
            print("rollout", it, name, "differs at", int(d.sum()), "of", d.numel(), "first:", torch.nonzero(d)[:5].tolist())
    print(it, "state equal:", np.array_equal(r1.env.get_state(), r2.env.get_state()), "h1a equal:", torch.equal(r1._fwd["h1a"], r2._fwd["h1a"]),
          "h1c equal:", torch.equal(r1._fwd["h1c"], r2._fwd["h1c"]), "logits equal:", torch.equal(r1._logits_pad, r2._logits_pad))
    r1.update(*b1); r2.update(*b2)
    print(it, "params equal:", torch.equal(r1.flat.w, r2.flat.w))
