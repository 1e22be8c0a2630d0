#!/usr/bin/env python3
"""Where the wall clock of a 20-step timed region goes: one uavenv_step_many(20) call at 4096 envs between two synchronises, with and
without HIP events inside, device-wide vs stream synchronise.  us per region, best / median of 30 (each preceded by busy scratch work)."""
import ctypes as C, json, os, statistics, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drl_uav_cellularnet_amd import BatchedMobiEnv
dev = torch.device("cuda", 0)
mk = lambda: BatchedMobiEnv(4096, nBS=4, nUE=20, grid_n=100, groups=[5, 5, 5, 5], device=dev, seed=0x5EED)
env, scratch = mk(), mk()
g = torch.Generator().manual_seed(1)
tape = torch.randint(0, 625, (20, 4096), generator=g, dtype=torch.int64).to(dev)
out = env.step_many(tape)
st = env.out_struct_for(out)
stream = C.c_void_p(env._stream())
call = lambda: env._lib.uavenv_step_many(env._h, tape.data_ptr(), 20, C.byref(st), stream)
hip = C.CDLL("libamdhip64.so")
res = {}
def busy():
    for t in range(64):
        scratch.step(tape[t % 20])
def region(events, sync):
    busy()
    sync()
    e0 = e1 = None
    if events:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    if events: e0.record()
    call()
    if events: e1.record()
    sync()
    return (time.perf_counter() - t0) * 1e6
dsync = torch.cuda.synchronize
ssync = lambda: hip.hipStreamSynchronize(stream)
for name, ev, sy in (("no_events_device_sync", False, dsync), ("events_device_sync", True, dsync), ("no_events_stream_sync", False, ssync), ("events_stream_sync", True, ssync)):
    for _ in range(5): region(ev, sy)
    v = [region(ev, sy) for _ in range(30)]
    res[name] = {"best": round(min(v), 1), "median": round(statistics.median(v), 1)}
env.launch_timing(True)
def region_t():
    busy(); dsync()
    env.launch_timing(True)
    t0 = time.perf_counter()
    call()
    dsync()
    w = (time.perf_counter() - t0) * 1e6
    return w, env.launch_times_us()[0]
for _ in range(5): region_t()
v = [region_t() for _ in range(30)]
res["dispatch_events_device_sync"] = {"best": round(min(a for a, _ in v), 1), "median": round(statistics.median([a for a, _ in v]), 1),
                                      "kernel_us_median": round(statistics.median([b for _, b in v]), 1)}
env.launch_timing(False)
print(json.dumps(res))
