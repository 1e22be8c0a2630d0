#!/usr/bin/env python3
"""Bandwidth reference at the loss kernel's size: torch elementwise passes over a [409600, 640] float32 tensor (in place, copy, read only)."""
import torch, time
x = torch.rand(409600, 640, device="cuda")
y = torch.empty_like(x)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
print("inplace mul (r+w 2.1 GB): %.3f ms" % t(lambda: x.mul_(1.0001)))
print("copy (r 1.05 + w 1.05 GB): %.3f ms" % t(lambda: y.copy_(x)))
print("read-only sum (1.05 GB): %.3f ms" % t(lambda: x.sum()))
