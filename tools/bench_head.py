#!/usr/bin/env python3
"""uavagent_actor_head_f32 alone: us per call by row count and tile height (UAVAGENT_HEAD_RB = 1: 16-row workgroups, 2: 32-row)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drl_uav_cellularnet_amd import _agent_capi as A
H, NA = 200, 625
g = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda *s: torch.rand(s, device="cuda", generator=g) * 2.0 - 1.0
w2t, b2 = (rnd(H, H) * 0.2).contiguous(), rnd(H)
w3t, b3p = torch.zeros((640, H), device="cuda"), torch.zeros(640, device="cuda")
w3t[:NA], b3p[:NA] = (rnd(H, NA) * 0.3).t(), rnd(NA)
res = {}
for n in (16, 32, 4096, 4128, 8192):
    h1 = (rnd(n, H) * 4 + 2).clamp_(0, 6)
    u = torch.rand(n, device="cuda", generator=g)
    h2, lg, act = torch.empty(n, H, device="cuda"), torch.empty(n, 640, device="cuda"), torch.empty(n, dtype=torch.int64, device="cuda")
    for rb in ("1", "2"):
        os.environ["UAVAGENT_HEAD_RB"] = rb
        ts = []
        for rep in range(5):
            for _ in range(3):
                A.actor_head(h1, w2t, b2, w3t, b3p, u, NA, h2, lg, act)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                A.actor_head(h1, w2t, b2, w3t, b3p, u, NA, h2, lg, act)
            e1.record(); torch.cuda.synchronize()
            ts.append(round(e0.elapsed_time(e1) * 1e3 / 20, 2))
        res["rows=%d,tile=%d" % (n, 16 * int(rb))] = ts
print(json.dumps(res))
