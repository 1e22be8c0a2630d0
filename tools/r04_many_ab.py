#!/usr/bin/env python3
"""Round-4 A/B of uavenv_step_many's launch forms, interleaved in one process on one box:
   plain (one launch, wavefront w = env-wavefront w) and one-launch rotation (S persistent wavefronts with hand-offs, UAVENV_ROTATE=1)
   -- for each (n_envs, steps per call) pair asked for.  (profiles/r04a_* also has round 3's several-launch form, removed since.)
   us per call from HIP events around `reps` back-to-back calls, and a single-call figure (one call between two synchronises: what a
   20-step timed region sees).  Prints one JSON object.   usage: r04_many_ab.py [n_envs:T ...]"""
import json
import os
import sys
import time

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


def single(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6


def main():
    dev = torch.device("cuda", 0)
    rounds = int(os.environ.get("ROUNDS", "5"))
    pairs = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]] or [(4096, 20), (4096, 100), (8192, 20), (8192, 100)]
    res = {"rounds": rounds, "us_per_call": {}, "us_single_call_wall": {}, "launches": {}}
    g = torch.Generator().manual_seed(1)
    import ctypes as C

    legs = {}
    for n, T in pairs:
        tape = torch.randint(0, 625, (T, n), generator=g, dtype=torch.int64).to(dev)
        for form, val in (("plain", "0"), ("rot_one_launch", "1")):
            os.environ["UAVENV_ROTATE"] = val                                  # read once per handle, in uavenv_create
            e = BatchedMobiEnv(n, nBS=4, nUE=20, grid_n=100, groups=[5, 5, 5, 5], device=dev, seed=0x5EED)
            os.environ.pop("UAVENV_ROTATE")
            nl = C.c_int(-1)
            e._lib.uavenv_debug_rotation_info(e._h, T, C.byref(nl), None)
            key = "%dx%d_%s" % (n, T, form)
            res["launches"][key] = nl.value
            if form != "plain" and nl.value == 0:
                continue                                                       # no schedule of this form for this shape
            out = e.step_many(tape)
            legs[key] = (lambda e=e, tape=tape, out=out: e.step_many(tape, out=out, refresh_out=False))
    for name, fn in legs.items():       # warm
        timed(fn, 3)
    for r in range(rounds):
        for name, fn in legs.items():
            res["us_per_call"].setdefault(name, []).append(round(timed(fn, 10), 2))
            res["us_single_call_wall"].setdefault(name, []).append(round(single(fn), 1))
    res["best_us_per_step"] = {k: round(min(v) / int(k.split("_")[0].split("x")[1]), 3) for k, v in res["us_per_call"].items()}
    res["best_env_steps_per_s"] = {k: round(int(k.split("x")[0]) / (v * 1e-6), 0) for k, v in res["best_us_per_step"].items()}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
