#!/usr/bin/env python3
"""Steps/s of the N = 1 drop-in (drl_uav_cellularnet_amd.MobiEnvironment.step: one env, NumPy in / NumPy out, full (nBS+1, G, G)
float64 state returned per step) -- the like-for-like figure next to the reference's own MobiEnvironment.step, which measured
~670 steps/s (4 UAV x 20 UE) and ~400-470 steps/s (4 x 40) per process in the build container.
    python tools/bench_shim.py [--steps 2000]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000)
    a = ap.parse_args()
    import numpy as np

    from drl_uav_cellularnet_amd import MobiEnvironment

    out = {}
    for n_ue in (20, 40):
        env = MobiEnvironment(4, n_ue, 100)
        env.reset()
        rs = np.random.RandomState(0)
        acts = rs.randint(0, 625, a.steps + 50)
        for t in range(50):
            env.step(acts[t])
        t0 = time.perf_counter()
        for t in range(a.steps):
            s, r, d, info = env.step(acts[50 + t])
            if d:
                env.reset()
        el = time.perf_counter() - t0
        out["4x%d" % n_ue] = {"steps_per_s": a.steps / el, "us_per_step": el / a.steps * 1e6, "state_shape": list(s.shape)}
    print(json.dumps({"metric": "MobiEnvironment.step (N = 1 drop-in shim) steps/s", "grid": 100, **out}))


if __name__ == "__main__":
    main()
