#!/usr/bin/env python3
"""How fast can the host issue step() calls?  Tiny batch (GPU work negligible) -> launch-to-launch time is the host
path: Python wrapper + ctypes + hipLaunchKernel.  Also times the raw C-ABI call without the Python wrapper."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from drl_uav_cellularnet_amd import BatchedMobiEnv, _capi

for n in (64, 4096):
    env = BatchedMobiEnv(n)
    a = torch.randint(0, 625, (n,), device=env.device)
    K = 20000
    for _ in range(200):
        env.step(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        env.step(a)
    t_issue = time.perf_counter() - t0          # host time to ISSUE K steps (no sync)
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    lib, h, out, stream = env._lib, env._h, env._out_ref, env._stream()
    ap = a.data_ptr()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        lib.uavenv_step(h, ap, None, out, stream)
    t_raw_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_raw_all = time.perf_counter() - t0
    print("envs %5d | env.step(): issue %.2f us/step, issue+drain %.2f us/step | raw ctypes uavenv_step: issue %.2f, "
          "issue+drain %.2f us/step" % (n, t_issue / K * 1e6, t_all / K * 1e6, t_raw_issue / K * 1e6, t_raw_all / K * 1e6))

# --- does the stream matter?  torch's default stream is the legacy NULL stream (handle 0) --------------------
print("default stream handle:", torch.cuda.current_stream().cuda_stream)
side = torch.cuda.Stream()
for label, ctx in (("null stream", torch.cuda.stream(torch.cuda.default_stream())), ("created stream", torch.cuda.stream(side))):
    with ctx:
        env = BatchedMobiEnv(64)
        a = torch.randint(0, 625, (64,), device=env.device)
        lib, h, out, ap = env._lib, env._h, env._out_ref, a.data_ptr()
        stream = env._stream()
        for _ in range(500):
            lib.uavenv_step(h, ap, None, out, stream)
        torch.cuda.synchronize()
        K = 20000
        t0 = time.perf_counter()
        for _ in range(K):
            lib.uavenv_step(h, ap, None, out, stream)
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print("envs    64 | %-14s (handle %s) raw uavenv_step: issue %.2f us/step, issue+drain %.2f us/step" % (
            label, stream, t_issue / K * 1e6, t_all / K * 1e6))
# --- cost of an empty torch op for scale -------------------------------------------------------------------------
x = torch.zeros(64, device="cuda")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20000):
    x.add_(1.0)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
print("torch x.add_(1) on 64 floats: issue %.2f us/op, issue+drain %.2f us/op" % (t_issue / 20000 * 1e6, (time.perf_counter() - t0) / 20000 * 1e6))
