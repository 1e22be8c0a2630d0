#!/usr/bin/env python3
"""A/B of rollout-collection variants at BASELINE config 3 (8192 envs x 50 steps), interleaved in one process: ms per collect()."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drl_uav_cellularnet_amd import BatchedMobiEnv
from drl_uav_cellularnet_amd.agent import A2CRunner
variants = {"persistent pair of gated kernels": {"persistent_rollout": True}, "default (pipelined halves, heads alternate)": {"pipeline_halves": True, "persistent_rollout": False}, "unsplit (round 3)": {"pipeline_halves": False, "persistent_rollout": False}, "separate_obs_indices": {"fused_obs": False},
            "three_launches": {"fused_head": False}}
if len(sys.argv) > 1:
    variants = {k: v for k, v in variants.items() if any(a in k for a in sys.argv[1:])}
variants["pipelined halves, EAGER launches (no graph)"] = {"pipeline_halves": True, "collect_launch": "eager", "persistent_rollout": False}
variants["persistent pair, EAGER launches (no graph)"] = {"persistent_rollout": True, "collect_launch": "eager"}
variants["unsplit, EAGER launches (no graph)"] = {"pipeline_halves": False, "collect_launch": "eager", "persistent_rollout": False}
for parts in (2, 3, 4):
    for mode in ("alternate", "free", "stagger"):
        if (parts, mode) != (2, "alternate"):
            variants["pipelined %d parts, %s" % (parts, mode)] = {"_mode": mode, "_parts": parts, "pipeline_halves": True, "persistent_rollout": False}
if len(sys.argv) > 1:
    variants = {k: v for k, v in variants.items() if any(a in k for a in sys.argv[1:])}
runners = {}
for k, kw in variants.items():
    kw = dict(kw)
    os.environ["UAVAGENT_PIPE_MODE"] = kw.pop("_mode", "alternate")
    os.environ["UAVAGENT_PIPE_PARTS"] = str(kw.pop("_parts", 2))
    env = BatchedMobiEnv(8192, nBS=4, nUE=20, grid_n=100, groups=[5, 5, 5, 5])
    runners[k] = A2CRunner(env, rollout=50, **kw)
    runners[k].collect(); runners[k].collect()
res = {k: [] for k in variants}
for rnd in range(4):
    for k, r in runners.items():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            r.collect()
        torch.cuda.synchronize(); res[k].append(round((time.perf_counter() - t0) / 5 * 1e3, 3))
print(json.dumps(res))
