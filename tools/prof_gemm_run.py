#!/usr/bin/env python3
"""A few launches of each hand-written GEMM for rocprofv3 (tools/r03_gemm_prof.sh)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drl_uav_cellularnet_amd import _agent_capi as A
dev = torch.device("cuda", 0)
M, H, NA = 409600, 200, 625
g = torch.Generator(device=dev).manual_seed(5)
rnd = lambda *s: torch.rand(s, device=dev, generator=g) * 2.0 - 1.0
x, y = rnd(M, H), rnd(M, H)
dl = torch.zeros(M, 640, device=dev); dl[:, :NA] = rnd(M, NA)
h = (rnd(M, H) * 4.0 + 2.0).clamp_(0.0, 6.0)
w2, b2 = rnd(H, H) * 0.1, rnd(H)
w3p = torch.zeros(H, 640, device=dev); w3p[:, :NA] = rnd(H, NA) * 0.1
out = torch.empty(M, H, device=dev)
gw2, gw3, gb2, gb3 = torch.empty(H, H, device=dev), torch.empty(H, NA, device=dev), torch.empty(H, device=dev), torch.empty(NA, device=dev)
ws200, ws625 = A.gemm_tn_workspace(M, H, dev), A.gemm_tn_workspace(M, NA, dev)
reps = int(os.environ.get("REPS", "5"))
wscs = A.gemm_rows_workspace(M, dev)
w2t = w2.t().contiguous()
for _ in range(reps):
    A.gemm_rows(x, w2t, out, w_transposed=True, bias=b2, relu6=True)                                   # critic layer 2 forwards
    A.gemm_rows(y, w2, out, w_transposed=True, relu6_mask_h=h, colsum_out=gb2, workspace=wscs)         # dX through a 200-wide layer
    A.gemm_rows(dl, w3p, out, w_transposed=True, relu6_mask_h=h)                                       # dX through the policy head
    A.gemm_tn(x, y, gw2, ws200, dbias_out=gb2)                                                         # dW 200 x 200
    A.gemm_tn(x, dl[:, :NA], gw3, ws625, dbias_out=gb3)                                                # dW 200 x 625
# one rollout step's actor head at 8192 rows (fused kernel, and the three launches it replaces)
N = 8192
xs, us = x[:N].contiguous(), torch.rand(N, device=dev, generator=g)
w3t = w3p.t().contiguous().t().contiguous() if False else None
w3tp = torch.zeros(640, H, device=dev); w3tp[:NA] = (torch.rand(NA, H, device=dev, generator=g) - 0.5) * 0.2
b3p = torch.zeros(640, device=dev)
h2s, lgs, acts = torch.empty(N, H, device=dev), torch.zeros(N, 640, device=dev), torch.empty(N, dtype=torch.int64, device=dev)
for _ in range(reps):
    A.actor_head(xs, w2t, b2, w3tp, b3p, us, NA, h2s, lgs, acts)
    A.gemm_rows(xs, w2t, h2s, w_transposed=True, bias=b2, relu6=True)
    A.gemm_rows(h2s, w3tp, lgs, w_transposed=True, bias=b3p)
    A.sample_actions(lgs[:, :NA], us, out=acts)
torch.cuda.synchronize()
print("ok")
