#!/usr/bin/env python3
"""Phase timeline of the packed step kernel from in-kernel s_memtime stamps (diagnostic build -DUAVENV_STAMPS).
   UAVENV_LIB=ab_build/libuavenv_stamps.so python tools/stamps.py [envs]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from drl_uav_cellularnet_amd import BatchedMobiEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = BatchedMobiEnv(n)
waves = (n + 2) // 3
waves_alloc = (waves + 3) // 4 * 4
buf = torch.zeros((waves_alloc, 8), dtype=torch.int64, device=env.device)
env._lib.uavenv_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_void_p]
assert env._lib.uavenv_debug_set_stamp_buffer(env._h, buf.data_ptr()) == 0
a = torch.randint(0, 625, (n,), device=env.device)
for _ in range(300):
    env.step(a)
torch.cuda.synchronize()
t = buf.cpu().numpy()[:waves].astype(np.int64)
names = ["start -> kernarg+lane setup", "load phase issue + ALL loads returned", "UAV move", "mobility", "channel update",
         "store phase issue", "stores acknowledged"]
d = np.diff(t, axis=1)
print("envs %d, %d wavefronts; s_memtime ticks, median [p10, p90] over wavefronts (last launch)" % (n, waves))
for k, name in enumerate(names):
    print("  %-40s %7.0f  [%6.0f, %6.0f]" % (name, np.median(d[:, k]), np.percentile(d[:, k], 10), np.percentile(d[:, k], 90)))
tot = t[:, 7] - t[:, 0]
print("  %-40s %7.0f  [%6.0f, %6.0f]" % ("wave lifetime (start -> stores acked)", np.median(tot), np.percentile(tot, 10), np.percentile(tot, 90)))
print("  first wave start -> last wave end: %d ticks;  wave starts spread over %d ticks" % (t[:, 7].max() - t[:, 0].min(), t[:, 0].max() - t[:, 0].min()))
