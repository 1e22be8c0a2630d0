#!/usr/bin/env python3
"""A/B of A2C update variants at BASELINE config 3 (8192 envs x 50 steps), interleaved in one process: ms per train_rollout() and per
update (collect() timed separately), and whether the variants leave bit-identical parameters behind after the same rollouts."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drl_uav_cellularnet_amd import BatchedMobiEnv
from drl_uav_cellularnet_amd.agent import A2CRunner
variants = {"dw_on_side_stream": {"overlap_dw": True}, "dw_in_line": {}, "dw_in_line_no_early_sort": {"early_sort": False}}
runners = {}
for k, kw in variants.items():
    env = BatchedMobiEnv(8192, nBS=4, nUE=20, grid_n=100, groups=[5, 5, 5, 5])
    runners[k] = A2CRunner(env, rollout=50, **kw)
    for _ in range(3):
        runners[k].train_rollout()
ws = [r.flat.w for r in runners.values()]
res = {"same_parameters_after_3_rollouts": bool(all(torch.equal(ws[0], w) for w in ws[1:])), "train_rollout_ms": {k: [] for k in variants},
       "update_ms": {k: [] for k in variants}, "stats": {k: {kk: vv for kk, vv in r.stats.items() if kk in ("dw_on_side_stream", "hip_gemms", "forward_reused")} for k, r in runners.items()}}
for rnd in range(4):
    for k, r in runners.items():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            r.train_rollout()
        torch.cuda.synchronize(); res["train_rollout_ms"][k].append(round((time.perf_counter() - t0) / 5 * 1e3, 3))
    for k, r in runners.items():
        tu = 0.0
        for _ in range(5):
            bufs = r.collect()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r.update(*bufs)
            torch.cuda.synchronize(); tu += time.perf_counter() - t0
        res["update_ms"][k].append(round(tu / 5 * 1e3, 3))
print(json.dumps(res))
