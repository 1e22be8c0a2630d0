#!/usr/bin/env python3
"""The rollout's two dense layers at 8192 rows through uavagent_gemm_rows_f32 (transposed / padded weights) against torch.addmm."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drl_uav_cellularnet_amd import _agent_capi as A
dev = torch.device("cuda", 0)
N, H, NA = 8192, 200, 625
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *s: torch.rand(s, device=dev, generator=g) - 0.5
x, w2, b2, w3, b3 = rnd(N, H), rnd(H, H), rnd(H), rnd(H, NA), rnd(NA)
w2t = w2.t().contiguous()
w3t = torch.zeros(640, H, device=dev); w3t[:NA] = w3.t()
b3p = torch.zeros(640, device=dev); b3p[:NA] = b3
h2, lg, h2b, lgb = torch.empty(N, H, device=dev), torch.zeros(N, 640, device=dev), torch.empty(N, H, device=dev), torch.empty(N, NA, device=dev)
def timed(fn, reps=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
res = {"cfg": os.environ.get("UAVGEMM_SMALL", "0")}
res["gemm2_hip_us"] = timed(lambda: A.gemm_rows(x, w2t, h2, w_transposed=True, bias=b2, relu6=True))
res["gemm3_hip_us"] = timed(lambda: A.gemm_rows(h2, w3t, lg, w_transposed=True, bias=b3p))
res["gemm2_torch_us"] = timed(lambda: torch.addmm(b2, x, w2, out=h2b).clamp_(0, 6))
res["gemm3_torch_us"] = timed(lambda: torch.addmm(b3, h2b, w3, out=lgb))
u = torch.rand(N, device=dev, generator=g); act = torch.empty(N, dtype=torch.int64, device=dev)
res["head_fused_us"] = timed(lambda: A.actor_head(x, w2t, b2, w3t, b3p, u, NA, h2, lg, act))
res["sample_us"] = timed(lambda: A.sample_actions(lg[:, :NA], u, out=act))
res["err2"] = float((h2 - (x @ w2 + b2).clamp(0, 6)).abs().max()); res["err3"] = float((lg[:, :NA] - (h2 @ w3 + b3)).abs().max())
print(json.dumps(res))
