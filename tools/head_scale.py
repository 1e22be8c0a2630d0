import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from drl_uav_cellularnet_amd import _agent_capi as A
dev = torch.device("cuda", 0); H, NA = 200, 625
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *s: torch.rand(s, device=dev, generator=g) - 0.5
w2t, b2 = rnd(H, H), rnd(H)
w3t = torch.zeros(640, H, device=dev); w3t[:NA] = rnd(NA, H)
b3p = torch.zeros(640, device=dev); b3p[:NA] = rnd(NA)
def timed(fn, reps=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps * 1e3, 2)
res = {}
for N in (32, 1024, 2048, 4096, 8192, 16384):
    x, u = rnd(N, H), torch.rand(N, device=dev, generator=g)
    h2, lg, act = torch.empty(N, H, device=dev), torch.zeros(N, 640, device=dev), torch.empty(N, dtype=torch.int64, device=dev)
    res[N] = {"fused": timed(lambda: A.actor_head(x, w2t, b2, w3t, b3p, u, NA, h2, lg, act)),
              "g2": timed(lambda: A.gemm_rows(x, w2t, h2, w_transposed=True, bias=b2, relu6=True)),
              "g3": timed(lambda: A.gemm_rows(h2, w3t, lg, w_transposed=True, bias=b3p)),
              "smp": timed(lambda: A.sample_actions(lg[:, :NA], u, out=act))}
print(json.dumps(res))
