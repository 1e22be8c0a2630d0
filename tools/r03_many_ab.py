#!/usr/bin/env python3
"""Round-3 measurements around uavenv_step_many at BASELINE configs[1] (4096 envs, 4 UAV x 20 UE), interleaved on one box:
  (a) nine-array outputs (uavenv_step_many) vs packed records (uavenv_step_many_packed), T = 100 steps per launch;
  (b) the bound of "chunk rotation" (VERDICT r2 next #3): 1366 wavefronts on 1024 SIMDs leave a third of the SIMDs with two waves.
      A rotation would run the same work as 4 launches of <= 1024 wavefronts x ~34 steps.  Its best case is measured here
      WITHOUT building it: 4 launches of a 3072-env batch (exactly 1024 wavefronts) x 34 steps, against 1 launch of 4096 x 100.
Prints one JSON object."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drl_uav_cellularnet_amd import BatchedMobiEnv  # noqa: E402


def timed(fn, reps):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps          # us per call


def main():
    dev = torch.device("cuda", 0)
    rounds = int(os.environ.get("ROUNDS", "5"))
    res = {"rounds": rounds, "legs": {}}

    def mk(n):
        return BatchedMobiEnv(n, nBS=4, nUE=20, grid_n=100, groups=[5, 5, 5, 5], device=dev, seed=0x5EED)

    g = torch.Generator().manual_seed(1)
    os.environ["UAVENV_ROTATE"] = "0"                 # read once per handle, in uavenv_create
    envs = {n: mk(n) for n in (4096, 3072)}
    os.environ.pop("UAVENV_ROTATE")
    er = mk(4096)                                      # automatic: the rotation schedule (4 launches of 1024 wavefronts)
    tapes = {(n, T): torch.randint(0, 625, (T, n), generator=g, dtype=torch.int64).to(dev) for n, T in ((4096, 100), (3072, 34), (3072, 100), (4096, 34))}
    e = envs[4096]
    out9 = e.step_many(tapes[(4096, 100)])
    outp = e.step_many_packed(tapes[(4096, 100)])
    e3 = envs[3072]
    o3 = e3.step_many(tapes[(3072, 34)])
    o3p = e3.step_many_packed(tapes[(3072, 34)])
    o3l = e3.step_many(tapes[(3072, 100)])
    o4s = e.step_many(tapes[(4096, 34)])
    outr = er.step_many(tapes[(4096, 100)])
    outrp = er.step_many_packed(tapes[(4096, 100)])
    legs = {
        "many_9arrays_4096x100": lambda: e.step_many(tapes[(4096, 100)], out=out9, refresh_out=False),
        "many_packed_4096x100": lambda: e.step_many_packed(tapes[(4096, 100)], out=outp),
        "ROTATED_many_9arrays_4096x100": lambda: er.step_many(tapes[(4096, 100)], out=outr, refresh_out=False),
        "ROTATED_many_packed_4096x100": lambda: er.step_many_packed(tapes[(4096, 100)], out=outrp),
        "rotation_bound_4x(3072x34)_9arrays": lambda: [e3.step_many(tapes[(3072, 34)], out=o3, refresh_out=False) for _ in range(4)],
        "rotation_bound_4x(3072x34)_packed": lambda: [e3.step_many_packed(tapes[(3072, 34)], out=o3p) for _ in range(4)],
        "many_9arrays_3072x100": lambda: e3.step_many(tapes[(3072, 100)], out=o3l, refresh_out=False),
        "many_9arrays_3x(4096x34)": lambda: [e.step_many(tapes[(4096, 34)], out=o4s, refresh_out=False) for _ in range(3)],
    }
    for name, fn in legs.items():       # warm
        timed(fn, 3)
    for r in range(rounds):
        for name, fn in legs.items():
            res["legs"].setdefault(name, []).append(round(timed(fn, 10), 2))
    res["note"] = ("us per call; rotation_bound legs do 4 x 34 = 136 steps of 3072 envs = the work of 1366 x 100 wave-steps spread over 1024 "
                   "wavefronts per launch (what a perfect rotation of the 4096-env batch would cost); compare with many_*_4096x100")
    res["us_per_step_4096"] = {k: round(min(v) / 100.0, 3) for k, v in res["legs"].items() if "4096x100" in k or "rotation" in k}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
