#!/usr/bin/env python3
"""fp32 GEMM shapes of the A2C update (M = 8192 envs x 50 steps), timed through torch (rocBLAS vs hipBLASLt): which of them
are far from the 157 TFLOP/s dense f32-MFMA peak?  Secondary tool; output kept under profiles/."""
import sys

import torch

M, H, NA = 409600, 200, 625
dev = "cuda"


def t_ms(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    x = torch.randn(M, H, device=dev)
    d = torch.randn(M, NA, device=dev)
    w2 = torch.randn(H, H, device=dev)
    w3 = torch.randn(H, NA, device=dev)
    b2, b3 = torch.randn(H, device=dev), torch.randn(NA, device=dev)
    o_h, o_a = torch.empty(M, H, device=dev), torch.empty(M, NA, device=dev)
    g_w2, g_w3 = torch.empty(H, H, device=dev), torch.empty(H, NA, device=dev)
    shapes = [
        ("fwd  [M,200]x[200,200]+b", lambda: torch.addmm(b2, x, w2, out=o_h), 2 * M * H * H),
        ("fwd  [M,200]x[200,625]+b", lambda: torch.addmm(b3, x, w3, out=o_a), 2 * M * H * NA),
        ("dX   [M,625]x[625,200]", lambda: torch.mm(d, w3.t(), out=o_h), 2 * M * H * NA),
        ("dX   [M,200]x[200,200]", lambda: torch.mm(x, w2.t(), out=o_h), 2 * M * H * H),
        ("dW   [200,M]x[M,625]", lambda: torch.mm(x.t(), d, out=g_w3), 2 * M * H * NA),
        ("dW   [200,M]x[M,200]", lambda: torch.mm(x.t(), x, out=g_w2), 2 * M * H * H),
    ]
    for lib in ("default", "hipblaslt", "tunableop"):
        if lib == "tunableop":
            try:
                import torch.cuda.tunable as tun

                tun.enable(True)
                tun.tuning_enable(True)
                tun.set_max_tuning_duration(2000)
                tun.set_filename("/tmp/tunableop_results.csv")
            except Exception as ex:
                print("tunableop not available:", ex)
                break
        if lib == "hipblaslt":
            try:
                torch.backends.cuda.preferred_blas_library("hipblaslt")
            except Exception as ex:
                print("hipblaslt not selectable:", ex)
                break
        print("== BLAS:", lib)
        for name, fn, flop in shapes:
            ms = t_ms(fn)
            print("  %-28s %8.3f ms  %6.1f TFLOP/s" % (name, ms, flop / ms / 1e9))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
