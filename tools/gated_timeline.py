#!/usr/bin/env python3
"""Timeline of ONE pair of blocks through the two persistent rollout kernels (diagnostic builds with -DUAVENV_GATE_STAMPS / -DUAVAGENT_GATE_STAMPS,
loaded through UAVENV_LIB / UAVAGENT_LIB): per step and block the s_memrealtime of  h1 seen -> actions published (policy kernel)  and
actions seen -> env step done -> encoded rows published (env kernel).  Prints average phase lengths in us over the steady-state steps."""
import ctypes as C, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drl_uav_cellularnet_amd import BatchedMobiEnv, _capi
from drl_uav_cellularnet_amd import _agent_capi as A

N, T = int(os.environ.get("N", 8192)), int(os.environ.get("T", 50))
env = BatchedMobiEnv(N, nBS=4, nUE=20, grid_n=100, groups=[5, 5, 5, 5])
dev = env.device
g = torch.Generator(device="cuda").manual_seed(1)
rows, hid, K = 5 * 100 * 100, 200, 24
rnd = lambda *s: torch.rand(s, device=dev, generator=g) * 2.0 - 1.0
wa, ba, wc, bc = rnd(rows, hid) * 0.1, rnd(hid), rnd(rows, hid) * 0.1, rnd(hid)
w2t, b2 = (rnd(hid, hid) * 0.2).contiguous(), rnd(hid)
w3t, b3p = torch.zeros((640, hid), device=dev), torch.zeros(640, device=dev)
w3t[:625], b3p[:625] = rnd(625, hid) * 0.3, rnd(625)
u = torch.rand((T, N), device=dev, generator=g)
h1a, h1c = torch.rand((T, N, hid), device=dev, generator=g) * 6, torch.empty((T, N, hid), device=dev)
h2, lg = torch.empty((T, N, hid), device=dev), torch.empty((T, N, 640), device=dev)
act = torch.zeros((T, N), dtype=torch.int64, device=dev)
idx = torch.empty((T + 1, N, K), dtype=torch.int64, device=dev)
rew = torch.empty((T, N), device=dev)
nb = (N + 15) // 16
g_obs, g_act = torch.zeros(nb, dtype=torch.int32, device=dev), torch.zeros(nb, dtype=torch.int32, device=dev)
claim = torch.zeros(2, dtype=torch.int32, device=dev)
stamps = torch.zeros((T, 2, 8), dtype=torch.int64, device=dev)
A.gate_prepare()
assert _capi.load().uavenv_debug_set_gate_stamps(C.c_void_p(stamps.data_ptr())) == 0
assert A.load().uavagent_debug_set_gate_stamps(C.c_void_p(stamps.data_ptr())) == 0
side = torch.cuda.Stream(device=dev, priority=-1)


def pair():
    g_obs.fill_(1); g_act.zero_(); claim.zero_()
    main = torch.cuda.current_stream(dev)
    side.wait_stream(main)
    A.actor_head_gated(h1a, w2t, b2, w3t, b3p, u, 625, h2, lg, act, g_obs, g_act, claim[1:2])
    with torch.cuda.stream(side):
        env.rollout_gated(act, g_act, g_obs, claim[0:1], wa, ba, h1a, wc, bc, h1c, idx_out=idx, reward_out=rew)
    main.wait_stream(side)


for _ in range(3):
    pair(); torch.cuda.synchronize()
assert A.device_error() == 0 and env.device_error() == 0
s = stamps.cpu().double() / 100.0          # us
lo, hi = 5, T - 2                          # steady state
seg = lambda a, b: float((b - a)[lo:hi].mean())
out = {"us_per_step_of_the_pair": round(float((s[hi, 0, 0] - s[lo, 0, 0]) / (hi - lo)), 2)}
for half in (0, 1):
    h = s[:, half]
    out["block %d" % half] = {"head: h1 seen -> actions published": round(seg(h[:, 0], h[:, 1]), 2),
                              "hand-off: actions published -> seen by the env kernel": round(seg(h[:, 1], h[:, 2]), 2),
                              "env step (actions seen -> outputs in L2)": round(seg(h[:, 2], h[:, 3]), 2),
                              "encoder (-> rows published)": round(seg(h[:, 3], h[:, 4]), 2),
                              "hand-off: rows published(t) -> h1 seen(t+1)": round(float((h[lo + 1:hi + 1, 0] - h[lo:hi, 4]).mean()), 2)}
print(json.dumps(out, indent=1))
print("first steps of block 0 (us from the first stamp): [h1 seen, actions out, actions seen, env done, rows out]")
t0 = s[0, 0, 0]
for t in range(6, 10):
    print(t, [[round(float(s[t, hf, k] - t0), 1) for k in range(5)] for hf in (0, 1)])
