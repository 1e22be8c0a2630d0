#!/usr/bin/env python3
"""s_memtime phase sums of gemm_rows_nt_kernel (-DUAVGEMM_STAMPS build): per wavefront total / prologue / MFMA phases / chunk boundaries /
epilogue, and the start times of the workgroups (how many run at once).  UAVAGENT_LIB=ab_build/libuavagent_stamps.so"""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drl_uav_cellularnet_amd import _agent_capi as A
dev = torch.device("cuda", 0)
M, H = 409600, 200
g = torch.Generator(device=dev).manual_seed(5)
y = torch.rand(M, H, device=dev, generator=g) - 0.5
h = (torch.rand(M, H, device=dev, generator=g) * 8 - 1).clamp_(0, 6)
w = torch.rand(H, H, device=dev, generator=g) - 0.5
out, cs = torch.empty(M, H, device=dev), torch.empty(H, device=dev)
ws = A.gemm_rows_workspace(M, dev)
res = {}
for name, kw in (("plain_colsum", {}), ("mask_colsum", {"relu6_mask_h": h})):
    for _ in range(3):
        A.gemm_rows(y, w, out, w_transposed=True, colsum_out=cs, workspace=ws, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); A.gemm_rows(y, w, out, w_transposed=True, colsum_out=cs, workspace=ws, **kw); e1.record(); torch.cuda.synchronize()
    n_wg = M // 128
    raw = ws.cpu().numpy()
    st = raw[n_wg * 208 * 4: n_wg * 208 * 4 + n_wg * 4 * 6 * 8].view(np.uint64).reshape(n_wg, 4, 6).astype(np.float64)
    t0 = st[:, 0, 5] - st[:, 0, 5].min()
    end = t0 + st[:, 0, 0]
    res[name] = {"ms": e0.elapsed_time(e1), "mean_cycles": {k: float(st[:, :, i].mean()) for i, k in enumerate(("total", "prologue", "mfma", "boundaries", "epilogue"))},
                 "kernel_span_cycles": float(end.max()), "wg_total_sum_over_span_x_cus": float(st[:, 0, 0].sum() / (end.max() * 256)),
                 "wg_start_percentiles": [float(np.percentile(t0, p)) for p in (0, 10, 25, 50, 75, 90, 100)]}
print(json.dumps(res))
