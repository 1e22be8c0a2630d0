#!/usr/bin/env python3
"""The learner's hand-written float32 MFMA GEMMs (csrc/agent_gemm.hip) against torch.mm on the shapes of an A2C update at BASELINE
config 3 (M = 8192 envs x 50 steps): correctness against a float64 product, then interleaved timing rounds in ONE process
(same box, same clocks).  Prints one JSON object.   python tools/bench_gemm_hip.py [--m 409600] [--rounds 5] [--tune]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drl_uav_cellularnet_amd import _agent_capi as A  # noqa: E402


def timed(fn, reps):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps          # ms per call


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=409600)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--tune", action="store_true", help="torch.mm through TunableOp with the shipped picks (what the learner uses)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    if args.tune:
        from drl_uav_cellularnet_amd import agent

        agent.enable_gemm_tuning()
    M, H, NA = args.m, 200, 625
    g = torch.Generator(device=dev).manual_seed(5)
    rnd = lambda *s: torch.rand(s, device=dev, generator=g) * 2.0 - 1.0
    x200, y200 = rnd(M, H), rnd(M, H)
    dl_pad = torch.zeros(M, 640, device=dev)                 # the learner's layout: 625 logits in rows of 640, zero tail
    dl_pad[:, :NA] = rnd(M, NA)
    dl = dl_pad[:, :NA]
    h = rnd(M, H) * 4.0 + 2.0                      # forward activations: some <= 0 and >= 6 after the clamp
    h.clamp_(0.0, 6.0)
    w2, w3, b2 = rnd(H, H) * 0.1, rnd(H, NA) * 0.1, rnd(H)
    w3p = torch.zeros(H, 640, device=dev)
    w3p[:, :NA] = w3
    out200, out200b = torch.empty(M, H, device=dev), torch.empty(M, H, device=dev)
    gw2, gw3, gb2, gb3 = torch.empty(H, H, device=dev), torch.empty(H, NA, device=dev), torch.empty(H, device=dev), torch.empty(NA, device=dev)
    gw2t, gw3t = torch.empty(H, H, device=dev), torch.empty(H, NA, device=dev)
    ws200, ws625 = A.gemm_tn_workspace(M, H, dev), A.gemm_tn_workspace(M, NA, dev)
    wscs = A.gemm_rows_workspace(M, dev)
    res = {"m_rows": M, "tunableop": bool(args.tune), "checks": {}, "ms": {}, "tflops": {}}

    def relerr(got, want64):
        return float((got.double() - want64).abs().max() / want64.abs().max())

    # ---- correctness on a slice small enough for float64 products (the kernels see the full M in the timing legs) ----
    Mc = min(M, 40000 + 37)                       # not a multiple of any tile
    xs, ys, dls, hs, dlps = x200[:Mc], y200[:Mc], dl[:Mc], h[:Mc], dl_pad[:Mc]
    o = torch.empty(Mc, H, device=dev)
    A.gemm_rows(xs, w2, o, bias=b2, relu6=True)
    res["checks"]["fwd_bias_relu6"] = relerr(o, (xs.double() @ w2.double() + b2.double()).clamp(0.0, 6.0))
    A.gemm_rows(xs, w2, o)
    res["checks"]["fwd_plain"] = relerr(o, xs.double() @ w2.double())
    A.gemm_rows(ys, w2, o, w_transposed=True)
    res["checks"]["dx200"] = relerr(o, ys.double() @ w2.double().t())
    A.gemm_rows(ys, w2, o, w_transposed=True, relu6_mask_h=hs)
    res["checks"]["dx200_mask"] = relerr(o, (ys.double() @ w2.double().t()) * ((hs > 0) & (hs < 6)).double())
    A.gemm_rows(dlps, w3p, o, w_transposed=True)
    res["checks"]["dx625"] = relerr(o, dls.double() @ w3.double().t())
    A.gemm_rows(dls.contiguous(), w3, o, w_transposed=True)
    res["checks"]["dx625_unaligned_kernel"] = relerr(o, dls.double() @ w3.double().t())
    wsc, wsc2 = A.gemm_tn_workspace(Mc, NA, dev), A.gemm_tn_workspace(Mc, H, dev)
    A.gemm_tn(xs, ys, gw2, wsc2, dbias_out=gb2)
    res["checks"]["dw200"] = relerr(gw2, xs.double().t() @ ys.double())
    res["checks"]["dbias200"] = relerr(gb2, ys.double().sum(dim=0))
    A.gemm_tn(xs, dls, gw3, wsc, dbias_out=gb3)
    res["checks"]["dw625"] = relerr(gw3, xs.double().t() @ dls.double())
    res["checks"]["dbias625"] = relerr(gb3, dls.double().sum(dim=0))
    first = gw3.clone()
    A.gemm_tn(xs, dls, gw3, wsc, dbias_out=gb3)
    res["checks"]["dw625_bit_reproducible"] = bool(torch.equal(first, gw3))
    # torch.mm's own error on the same data, for scale
    res["checks"]["torch_dw625"] = relerr(torch.mm(xs.t(), dls), xs.double().t() @ dls.double())
    res["checks"]["torch_dx625"] = relerr(torch.mm(dls, w3.t()), dls.double() @ w3.double().t())

    legs = {
        "fwd200_hip": (lambda: A.gemm_rows(x200, w2, out200, bias=b2, relu6=True), 2.0 * M * H * H),
        "fwd200_torch": (lambda: torch.addmm(b2, x200, w2, out=out200b).clamp_(0.0, 6.0), 2.0 * M * H * H),
        "dx200_hip": (lambda: A.gemm_rows(y200, w2, out200, w_transposed=True), 2.0 * M * H * H),
        "dx200_mask_hip": (lambda: A.gemm_rows(y200, w2, out200, w_transposed=True, relu6_mask_h=h), 2.0 * M * H * H),
        "dx200_mask_colsum_hip": (lambda: A.gemm_rows(y200, w2, out200, w_transposed=True, relu6_mask_h=h, colsum_out=gb2, workspace=wscs), 2.0 * M * H * H),
        "dx200_torch": (lambda: torch.mm(y200, w2.t(), out=out200b), 2.0 * M * H * H),
        "dx625_hip": (lambda: A.gemm_rows(dl_pad, w3p, out200, w_transposed=True), 2.0 * M * H * NA),
        "dx625_mask_hip": (lambda: A.gemm_rows(dl_pad, w3p, out200, w_transposed=True, relu6_mask_h=h), 2.0 * M * H * NA),
        "dx625_torch": (lambda: torch.mm(dl, w3.t(), out=out200b), 2.0 * M * H * NA),
        "dw200_hip": (lambda: A.gemm_tn(x200, y200, gw2, ws200, dbias_out=gb2), 2.0 * M * H * H),
        "dw200_torch": (lambda: torch.mm(x200.t(), y200, out=gw2t), 2.0 * M * H * H),
        "dw625_hip": (lambda: A.gemm_tn(x200, dl, gw3, ws625, dbias_out=gb3), 2.0 * M * H * NA),
        "dw625_torch": (lambda: torch.mm(x200.t(), dl, out=gw3t), 2.0 * M * H * NA),
    }
    for name, (fn, _) in legs.items():
        timed(fn, 2)
    for _ in range(args.rounds):
        for name, (fn, _) in legs.items():
            res["ms"].setdefault(name, []).append(round(timed(fn, args.reps), 4))
    for name, (_, flop) in legs.items():
        res["tflops"][name] = round(flop / (min(res["ms"][name]) * 1e-3) / 1e12, 1)
    res["sum_ms_update_gemms"] = {
        "hip": round(min(res["ms"]["fwd200_hip"]) + 2 * min(res["ms"]["dx200_hip"]) + min(res["ms"]["dx625_hip"]) + 2 * min(res["ms"]["dw200_hip"]) + min(res["ms"]["dw625_hip"]), 3),
        "torch": round(min(res["ms"]["fwd200_torch"]) + 2 * min(res["ms"]["dx200_torch"]) + min(res["ms"]["dx625_torch"]) + 2 * min(res["ms"]["dw200_torch"]) + min(res["ms"]["dw625_torch"]), 3)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
