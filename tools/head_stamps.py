#!/usr/bin/env python3
"""s_memtime stamps of actor_head_kernel (-DUAVGEMM_STAMPS build: UAVAGENT_LIB=ab_build/libuavagent_stamps.so): where a workgroup's life goes.
Per wave, cycles from its start: first chunk landed | phase 1 (layer 2) MFMAs + h2 tile written | barrier | phase 2's first chunk landed (h2 store,
two chunk issues before) | phase 2 MFMAs done | logits tile in LDS | logits stores issued | end (action drawn); and the sum of the 20 chunk
boundary waits (counted vmcnt + s_barrier).  Medians over waves; one lone workgroup and a full grid."""
import ctypes as C, json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drl_uav_cellularnet_amd import _agent_capi as A
lib = A.load()
H, NA = 200, 625
g = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda *s: torch.rand(s, device="cuda", generator=g) * 2.0 - 1.0
w2t, b2 = (rnd(H, H) * 0.2).contiguous(), rnd(H)
w3t, b3p = torch.zeros((640, H), device="cuda"), torch.zeros(640, device="cuda")
w3t[:NA], b3p[:NA] = (rnd(H, NA) * 0.3).t(), rnd(NA)
names = ["first_chunk_landed", "phase1_done", "after_barrier", "phase2_first_landed", "phase2_done", "logits_tile", "stores_issued", "end", "boundary_waits_sum"]
res = {}
for n, rb in ((16, "1"), (32, "2"), (4096, "1"), (8192, "2")):
    os.environ["UAVAGENT_HEAD_RB"] = rb
    rows = 16 * int(rb)
    n_wg = (n + rows - 1) // rows
    dbg = torch.zeros(n_wg * 8 * 10, dtype=torch.int64, device="cuda")
    assert lib.uavagent_debug_set_head_stamps(C.c_void_p(dbg.data_ptr())) == 0
    h1 = (rnd(n, H) * 4 + 2).clamp_(0, 6)
    u = torch.rand(n, device="cuda", generator=g)
    h2, lg, act = torch.empty(n, H, device="cuda"), torch.empty(n, 640, device="cuda"), torch.empty(n, dtype=torch.int64, device="cuda")
    for _ in range(5):
        A.actor_head(h1, w2t, b2, w3t, b3p, u, NA, h2, lg, act)
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(n_wg, 8, 10).astype(np.float64)
    t0 = d[:, :, 0]
    span = (t0 + d[:, :, 8]).max() - t0.min()
    res["rows=%d,tile=%d" % (n, rows)] = {"median_cycles": {k: float(np.median(d[:, :, i + 1])) for i, k in enumerate(names)},
                                          "p90_end": float(np.percentile(d[:, :, 8], 90)), "kernel_span_cycles": float(span),
                                          "wg_start_spread_cycles": float(t0.max() - t0.min())}
print(json.dumps(res, indent=1))
