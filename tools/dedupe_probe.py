#!/usr/bin/env python3
"""How many g rows would the table gradient read if a sample's g row were read once per spatial TILE of table rows instead of once per
index?  (Probe for a tiled form of uavagent_rows_grad_f32; prints pairs per sample for several tile shapes.)"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drl_uav_cellularnet_amd import BatchedMobiEnv
from drl_uav_cellularnet_amd.agent import A2CRunner
env = BatchedMobiEnv(8192, nBS=4, nUE=20, grid_n=100, groups=[5, 5, 5, 5])
r = A2CRunner(env, rollout=50)
for _ in range(int(os.environ.get("WARM", "20"))):
    r.train_rollout()
idx, act, rew, boot = r.collect()
T, N, K = idx.shape
G = 100
idx = idx.reshape(T * N, K)
pl, cell = idx // (G * G), idx % (G * G)
x, y = cell // G, cell % G
out = {"pairs_per_sample": K}
for tx, ty, planes in ((1, 1, False), (2, 2, True), (4, 4, True), (4, 4, False), (5, 5, True), (8, 8, True), (10, 10, True), (10, 10, False), (20, 20, True), (25, 25, True), (50, 50, True)):
    key = (x // tx) * 1000 + (y // ty) + (0 if planes else pl * 1000000)     # planes=True: a tile spans all planes
    ks, _ = key.sort(dim=1)
    uniq = 1 + (ks[:, 1:] != ks[:, :-1]).sum(dim=1)
    n_tiles = (G // tx) * (G // ty) * (1 if planes else 5)
    rows_per_tile = tx * ty * (5 if planes else 1)
    out["tile_%dx%d_%s" % (tx, ty, "allplanes" if planes else "perplane")] = {"reads_per_sample": round(float(uniq.float().mean()), 2), "tiles": n_tiles, "rows_per_tile": rows_per_tile}
print(json.dumps(out, indent=1))
