#!/usr/bin/env python3
"""Time of uavagent_a2c_loss_grad (kernel + its two reduce kernels) at the update's size, M = 409600 x 625 logits in rows of 640.
Tried with it in round 3, not kept: __expf / __logf / __fdividef (-6 %), 8 instead of 4 waves per SIMD (+3 %); tools/rw_probe.py gives the
bandwidth reference (an in-place elementwise pass over the same bytes: 0.35 ms against this kernel's 0.52)."""
import torch, sys, os
sys.path.insert(0, os.getcwd())
from drl_uav_cellularnet_amd import _agent_capi as A
M, NA = 409600, 625
g = torch.Generator(device="cuda").manual_seed(1)
lp = torch.zeros(M, 640, device="cuda"); lp[:, :NA] = torch.randn(M, NA, device="cuda", generator=g)
logits = lp[:, :NA]
v, tgt = torch.randn(M, device="cuda", generator=g), torch.randn(M, device="cuda", generator=g)
act = torch.randint(0, NA, (M,), device="cuda", generator=g)
dv, db, loss = torch.empty(M, device="cuda"), torch.empty(NA, device="cuda"), torch.zeros(3, dtype=torch.float64, device="cuda")
ws = A.loss_grad_workspace(NA, "cuda")
def run(): A.a2c_loss_grad(logits, v, tgt, act, 0.001, dv, db, loss, ws)
for _ in range(3): run()
torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize(); print("loss kernel ms", e0.elapsed_time(e1)/10)
