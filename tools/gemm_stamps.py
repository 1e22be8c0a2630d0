#!/usr/bin/env python3
"""Diagnostics for the dW kernel: (a) time against M (fixed cost vs per-chunk cost), (b) the s_memtime phase sums of the -DUAVGEMM_STAMPS
build (UAVAGENT_LIB=ab_build/libuavagent_stamps.so).  Prints one JSON object."""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drl_uav_cellularnet_amd import _agent_capi as A

def timed(fn, reps):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

dev = torch.device("cuda", 0)
H = 200
g = torch.Generator(device=dev).manual_seed(5)
res = {"lib": A.lib_path(), "dbg": os.environ.get("UAVGEMM_DBG"), "ms_vs_m": {}}
for M in ((409600,) if os.environ.get("UAVGEMM_DBG") else (102400, 204800, 409600, 819200)):
    x = torch.rand(M, H, device=dev, generator=g) - 0.5
    y = torch.rand(M, H, device=dev, generator=g) - 0.5
    gw, gb = torch.empty(H, H, device=dev), torch.empty(H, device=dev)
    ws = A.gemm_tn_workspace(M, H, dev)
    fn = lambda: A.gemm_tn(x, y, gw, ws, dbias_out=gb)
    timed(fn, 3)
    res["ms_vs_m"][M] = round(min(timed(fn, 10) for _ in range(3)), 4)
    if "stamps" in A.lib_path() and M == 409600:
        torch.cuda.synchronize()
        n_wg = 256
        tail = ws[-n_wg * 8 * 4 * 8:].cpu().numpy().view(np.uint64).reshape(n_wg, 8, 4).astype(np.float64)
        res["stamps_409600"] = {"per_wave_mean": {k: float(tail[:, :, i].mean()) for i, k in enumerate(("total", "prologue", "mfma_phases", "boundaries"))},
                                "per_wave_min_total": float(tail[:, :, 0].min()), "per_wave_max_total": float(tail[:, :, 0].max()),
                                "waves_0_3_vs_4_7_mfma": [float(tail[:, :4, 2].mean()), float(tail[:, 4:, 2].mean())],
                                "waves_0_3_vs_4_7_bound": [float(tail[:, :4, 3].mean()), float(tail[:, 4:, 3].mean())], "chunks": 50}
    del x, y, ws
print(json.dumps(res))
