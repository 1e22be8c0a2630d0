#!/usr/bin/env python3
"""BASELINE config 1: the evaluation loop of the reference's main_test.py (:46-113) on the HIP path.

  python tools/run_eval.py --out test/run1 [--trace ue_trace_10k.npy] [--actor Global_A_PARA.npz] [--steps 2000]

* The reference's trace file ue_trace_10k.npy is not in the mount (.MISSING_LARGE_BLOBS); README.md:32 says it was
  produced by saving the group model's integer UE cells.  --make-trace does exactly that with this repo's env
  ((T, 40, 2) int16, T = 10001 by default).
* Policy: greedy argmax of the actor (main_test.py:68,73), weights from save_actor_npz (a fresh N(0,0.1) net if none).
* Saves the arrays main_test.py saves: reward, decomposed_reward, sinr, time, outage_fraction, ue_location,
  bs_location, action, and sinr_area at steps 0, 500, ... (GetSinrInArea, :85-89)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def make_trace(n_rows=10001, n_ue=40, grid=100, seed=0x7ACE):
    """Integer UE cells of the group model, one row per mobility tick (README.md:32; main_test.py:114)."""
    from drl_uav_cellularnet_amd import BatchedMobiEnv

    env = BatchedMobiEnv(1, nBS=4, nUE=n_ue, grid_n=grid, seed=seed)         # ctor = 201 ticks, like mobile_env.py:76-98
    rows = np.empty((n_rows, n_ue, 2), np.int16)
    rows[0] = env.out["ue_xy"][0].cpu().numpy()
    stay = torch.full((1,), 5 ** 4 - 1, dtype=torch.int64, device=env.device)
    for t in range(1, n_rows):
        env.step(stay)                                                    # one next(self.mm) per step
        rows[t] = env.out["ue_xy"][0].cpu().numpy()
    return rows


def run_test(trace, out_dir, actor_npz=None, max_step=2000, n_bs=4, n_ue=40, grid=100, seed=0x5EED, area_every=500):
    from drl_uav_cellularnet_amd import MobiEnvironment
    from drl_uav_cellularnet_amd.agent import ACNet, load_actor_npz, obs_to_indices

    os.makedirs(out_dir, exist_ok=True)
    test_env = MobiEnvironment(n_bs, n_ue, grid, "read_trace", trace, seed=seed)       # main_test.py:51
    net = ACNet(test_env.observation_space_dim, test_env.action_space_dim)
    if actor_npz:
        load_actor_npz(net, actor_npz)                                                   # main_test.py:11-26
    net = net.to(test_env._env.device)
    test_env.reset()                                                                     # :54
    buf = {k: [] for k in ("reward", "decomposed_reward", "sinr", "time", "outage_fraction", "ue_location",
                           "bs_location", "action", "sinr_area")}
    step = 0
    while step <= max_step and step < len(trace):                                        # :69 (2001 calls)
        t0 = time.time()
        with torch.no_grad():
            idx = obs_to_indices(test_env._env.observation(), grid, n_bs)                # the state, as its non-zero cells
            action = int(torch.argmax(net.actor_only(idx), dim=1)[0])                    # :68,73 greedy
        buf["time"].append(time.time() - t0)
        _, r, done, info = test_env.step_test(np.array([action]), False)                 # :75
        buf["reward"].append(r)
        buf["sinr"].append(test_env.channel.current_BS_sinr.copy())
        buf["decomposed_reward"].append(info.r_dissect)
        buf["outage_fraction"].append(info.outage_fraction)
        buf["ue_location"].append(np.array(info.ue_loc))
        buf["bs_location"].append(np.array(info.bs_loc))
        buf["action"].append(info.bs_actions)
        if step % area_every == 0 or step == max_step:
            buf["sinr_area"].append(test_env.channel.GetSinrInArea(info.bs_loc))         # :85-89
        step += 1
    for k, v in buf.items():
        np.save(os.path.join(out_dir, k), np.array(v))                                   # :106-113
    return {k: np.array(v) for k, v in buf.items()}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="test/eval")
    ap.add_argument("--trace", default=None)
    ap.add_argument("--make-trace", default=None, help="write a synthesised trace to this .npy and exit")
    ap.add_argument("--trace-rows", type=int, default=10001)
    ap.add_argument("--actor", default=None)
    ap.add_argument("--steps", type=int, default=2000)
    a = ap.parse_args()
    if a.make_trace:
        np.save(a.make_trace, make_trace(a.trace_rows))
        print("wrote", a.make_trace)
        sys.exit(0)
    tr = np.load(a.trace, allow_pickle=False) if a.trace else make_trace(a.steps + 2)
    t0 = time.time()
    res = run_test(tr, a.out, a.actor, a.steps)
    print("eval: %d step_test calls in %.1f s, mean reward %.4f, mean outage fraction %.4f -> %s" % (
        len(res["reward"]), time.time() - t0, float(res["reward"].mean()), float(res["outage_fraction"].mean()), a.out))
