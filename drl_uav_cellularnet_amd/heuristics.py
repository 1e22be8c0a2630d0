"""The reference's hand-written "SINR-gradient" controller (gradient.py:14-37) on top of the drop-in MobiEnvironment:
a policy-free, deterministic-given-the-seed end-to-end driver (SURVEY.md section 8 f, row 3).  Host-side logic only;
every env operation (deepcopy, step_test) goes through the HIP path."""
import warnings
from copy import deepcopy

import numpy as np


def side_means(current_bs_sinr, ue_loc, bs_xy):
    """dir_grad of gradient.py:27-31 for one UAV: mean serving SINR of the UEs with x > bx, x <= bx, y > by, y <= by
    (NaN for an empty side, exactly like np.mean of an empty selection)."""
    out = np.full(4, np.nan)
    sel = (ue_loc[:, 0] > bs_xy[0], ue_loc[:, 0] <= bs_xy[0], ue_loc[:, 1] > bs_xy[1], ue_loc[:, 1] <= bs_xy[1])
    for k, m in enumerate(sel):
        if m.any():
            out[k] = np.mean(current_bs_sinr[m])
    return out


def choose_act_gradient(actual_env, n_act=5):
    """Choose_Act_Gradient (gradient.py:14-37): look one step ahead on a deep copy with every UAV staying (action
    n_act**nBS - 1 = "44..4"), then move each UAV towards the side whose UEs have the lowest mean serving SINR
    (digit 0: +x, 1: -x, 2: +y, 3: -y; ue_mobility.py:221-235).  The caller's env is not modified."""
    virtual_env = deepcopy(actual_env)                                   # :15
    stay = n_act ** actual_env.nBS - 1                                   # 624 for 4 UAVs (:17)
    virtual_env.step_test(stay, False)
    sinr, bs_loc, ue_loc = virtual_env.channel.current_BS_sinr, virtual_env.bsLoc, virtual_env.ueLoc   # :20-22
    action = 0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        for i_bs in range(len(bs_loc)):                                  # :26-32
            digit = int(np.nanargmin(side_means(sinr, ue_loc, bs_loc[i_bs])))
            action = action * n_act + digit                              # most significant digit -> UAV 0 (:34)
    return action


def run_gradient_policy(env, n_steps, reset_every=2000):
    """The loop of gradient.py:56-86 without its file output: returns (rewards, actions)."""
    env.reset()
    rewards, actions = [], []
    for step in range(n_steps):
        a = choose_act_gradient(env)
        _, r, done, _ = env.step_test(a, False)
        rewards.append(r)
        actions.append(a)
        if (step + 1) % reset_every == 0:
            env.reset()
    return np.array(rewards), np.array(actions)
