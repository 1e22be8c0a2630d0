"""MobiEnvironment: single-env, NumPy-in / NumPy-out drop-in for the reference class of the same name
(/root/reference/mobile_env.py:35-268), computed by the HIP path (libuavenv.so) with N = 1.

Same constructor signature, same ``reset()`` / ``step(action)`` / ``step_test(action)`` return shapes and
types, same attributes the reference's drivers read (main.py:37,173,190,198; a2c_single_thread.py:37,143-158;
main_test.py:51-89; gradient.py:15-22): ``action_space_dim``, ``observation_space_dim``, ``state``, ``bsLoc``,
``ueLoc``, ``step_n``, ``channel.current_BS``, ``channel.current_BS_sinr``; survives ``copy.deepcopy``.

Differences, all deliberate and documented in DESIGN.md:
 * randomness comes from per-env Philox streams (``seed=``), not the process-global NumPy stream;
 * ``nUE`` walkers move in 4 equal groups; the reference always walks 40 (mobile_env.py:76) and silently ignores
   the tail when nUE < 40 -- identical for the nUE = 40 every reference script uses;
 * mobility models no reference driver selects are not provided: ``in_coverage`` raises NotImplementedError,
   ``random_waypoint`` is not defined in the reference either (mobile_env.py:73 -> NameError there);
 * rendering (``render`` / ``plot_sinr_map``) is outside the hot path.
"""
import copy
import sys
from collections import namedtuple

import numpy as np

# module constants of the reference (mobile_env.py:18-32)
MAXSTEP = 2000
UE_STEP = 1
N_ACT = 5
MAX_UE_PER_GRID = 1
H_BS = 10
MIN_BS_DIST = 2
R_BS = 50
BS_STEP = 2

info_tup = namedtuple("info_tup", ["r_dissect", "step_n", "ue_loc", "bs_loc", "outage_fraction", "bs_actions"])


class _ChannelView:
    """The attributes of LTEChannel that callers read (main_test.py:79; gradient.py:20)."""

    def __init__(self, owner):
        self._o = owner
        self.nUE, self.nBS = owner.nUE, owner.nBS
        self.current_BS = np.zeros(owner.nUE, dtype=np.int64)
        self.current_BS_sinr = np.zeros(owner.nUE, dtype=np.float64)

    def GetCurrentAssociationMap(self, ueLoc=None):
        """(nBS, G, G) float64 count map of the serving UAV of every UE (channel.py:387-409)."""
        return np.array(self._o._dense()[1:])

    def GetSinrInArea(self, bsLoc=None):
        """(G, G) float64 dB map of the nearest-UAV DL SINR with fresh shadowing (channel.py:411-433) for the UAV cells
        ``bsLoc`` (default: the env's current cells, which is what every caller passes -- main_test.py:89: info.bs_loc)."""
        import torch

        cells = None
        if bsLoc is not None:                                   # any (nBS, >= 2) array, as channel.py:411 accepts (z is ignored, Q1)
            cells = np.asarray(bsLoc)[:, :2].astype(np.int32).reshape(1, -1, 2)
            if cells.shape[1] != self._o.nBS:
                raise ValueError("bsLoc must have nBS rows")     # the reference indexes interfDL[bs_id] of ITS nBS (channel.py:425)
        return self._o._env.sinr_area(dtype=torch.float64, bs_xy=cells)[0].cpu().numpy()


class MobiEnvironment:
    def __init__(self, nBS, nUE, grid_n=200, mobility_model="group", test_mobi_file_name="", device=None,
                 seed=0x5EED, _env=None, _draw_hook=None):
        self.nBS, self.nUE = int(nBS), int(nUE)
        self.bs_h = H_BS
        self.grid_n = int(grid_n)
        self.boundaries = [1, self.grid_n, 1, self.grid_n]          # mobile_env.py:44-45
        self.mobility_model = mobility_model
        self._draw_hook = _draw_hook   # tests: callable(kind) -> dict of injected draws for that call
        if mobility_model not in ("group", "read_trace"):
            if mobility_model == "in_coverage":
                raise NotImplementedError("mobility_model 'in_coverage' is not selected by any reference driver "
                                          "and is not provided (SURVEY.md section 2 row 5)")
            sys.exit("mobility model not defined")                    # mobile_env.py:90-91
        from .batched_env import BatchedMobiEnv

        if mobility_model == "read_trace":
            if isinstance(test_mobi_file_name, np.ndarray):             # extension: an in-memory trace (also used by deepcopy)
                self.ueLoc_trace = np.asarray(test_mobi_file_name)
            else:
                assert test_mobi_file_name                              # mobile_env.py:84
                self.ueLoc_trace = np.load(test_mobi_file_name, allow_pickle=False)
            if self.ueLoc_trace.ndim != 3 or self.ueLoc_trace.shape[1] < self.nUE or self.ueLoc_trace.shape[2] < 2:
                raise ValueError("trace must be (T, >=nUE, 2|3) integer cells (mobile_env.py:85-88)")
        self._env = _env if _env is not None else BatchedMobiEnv(
            1, nBS=self.nBS, nUE=self.nUE, grid_n=self.grid_n, device=device, seed=seed, f64_outputs=True,
            construct=False)
        cfg = self._env.cfg
        self.initBsLoc = np.array([[cfg.bs_init_xy[b][0], cfg.bs_init_xy[b][1], self.bs_h] for b in range(self.nBS)],
                                  dtype=np.int64)                     # mobile_env.py:58
        self.bsLoc = copy.deepcopy(self.initBsLoc)
        self.ueLoc = np.zeros((self.nUE, 2), dtype=np.int64)
        self.channel = _ChannelView(self)
        self.action_space_dim = N_ACT ** self.nBS                      # mobile_env.py:104
        self.observation_space_dim = self.grid_n * self.grid_n * (self.nBS + 1) * MAX_UE_PER_GRID  # :105
        self.state = np.zeros((self.nBS + 1, self.grid_n, self.grid_n))   # mobile_env.py:107
        self.step_n = 0
        if _env is None:
            self._construct()

    # ---- helpers ---------------------------------------------------------------------------------------------
    def _draws(self, kind):
        return self._draw_hook(kind) if self._draw_hook is not None else {}

    def _construct(self):
        """mobile_env.py:76-98: generator + 200 warm-up ticks + one more tick + LTEChannel.__init__."""
        env = self._env
        if self.mobility_model == "group":
            if self._draw_hook is None:
                env.init()
                env.warmup(env.WARMUP_TICKS)
            else:
                env.init(**self._draw_hook("init"))
                for _ in range(env.WARMUP_TICKS):
                    d = self._draw_hook("warmup")
                    env.warmup(1, theta_u=d.get("theta_u"), group_u=d.get("group_u"))
            env.reset(**self._draws("ctor"))
        else:
            env.init()
            env.reset_trace(self._trace_row(0), fading=self._draws("ctor").get("fading"))
        self._pull(refresh_state=False)
        self.association = self.channel.GetCurrentAssociationMap(self.ueLoc)   # mobile_env.py:99
        self.state = np.zeros((self.nBS + 1, self.grid_n, self.grid_n))         # stays zero until reset/step (:107)
        self.step_n = 0

    def _trace_row(self, i):
        return np.ascontiguousarray(self.ueLoc_trace[i][:self.nUE, :2].astype(np.int16))[None]

    def _dense(self):
        """env.state from the compact cells, on the host: plane 0 counts the UAV cells (GetGridMap, ue_mobility.py:173-188),
        plane 1 + b the UEs served by UAV b (GetCurrentAssociationMap, channel.py:387-409).  The device-side dense tensor
        (uavenv_obs_dense) holds the same numbers (tests/test_shim_dropin.py); building the nBS + nUE cells here avoids a
        400 KB device-to-host copy per step."""
        G = self.grid_n
        st = np.zeros((self.nBS + 1, G, G))
        np.add.at(st[0], (self.bsLoc[:, 0], self.bsLoc[:, 1]), 1.0)
        ok = (self.ueLoc >= 0).all(axis=1) & (self.ueLoc < G).all(axis=1)     # a walker on x == G has no cell
        np.add.at(st, (1 + self.channel.current_BS[ok], self.ueLoc[ok, 0], self.ueLoc[ok, 1]), 1.0)
        return st

    def _pull(self, refresh_state=True):
        h = self._env.out_host()                                        # ONE device-to-host copy + synchronisation
        self._h = h
        self.ueLoc = h["ue_xy"][0].astype(np.int64)
        xy = h["bs_xy"][0].astype(np.int64)
        self.bsLoc = np.concatenate([xy, np.full((self.nBS, 1), self.bs_h, dtype=np.int64)], axis=1)
        self.channel.current_BS = h["serving"][0].astype(np.int64)
        self.channel.current_BS_sinr = h["cur_sinr_f64"][0].copy()
        self.step_n = int(h["step_n"][0])
        if refresh_state:
            self.state = self._dense()                                  # mobile_env.py:139-140 / 169-170

    @staticmethod
    def _action(action):
        return int(np.asarray(action).ravel()[0])                     # int, NumPy int or shape-(1,) array (main_test.py:73-75)

    def _action_dev(self, a):
        """The action as a device tensor through a pinned staging word (no per-step tensor construction)."""
        import torch

        if getattr(self, "_act_host", None) is None:
            self._act_host = torch.zeros(1, dtype=torch.int64, pin_memory=True)
            self._act_dev = torch.zeros(1, dtype=torch.int64, device=self._env.device)
        self._act_host[0] = a
        self._act_dev.copy_(self._act_host, non_blocking=True)
        return self._act_dev

    def _digits(self, a):
        d = np.zeros(self.nBS)
        for i in range(self.nBS - 1, -1, -1):                          # Decimal_to_Base_N, ue_mobility.py:310-336
            d[i] = a % N_ACT
            a //= N_ACT
        return d

    # ---- the reference's public methods --------------------------------------------------------------------------
    def SetBsH(self, h):
        self.bs_h = h                                                  # mobile_env.py:111-112 (height never enters d)

    def reset(self):
        """mobile_env.py:115-148.  Returns a fresh (nBS+1, G, G) float64 array."""
        if self.mobility_model == "group":
            self._env.reset(**self._draws("reset"))
        else:
            self._env.reset_trace(self._trace_row(0), fading=self._draws("reset").get("fading"))
        self._pull()
        self.association = np.array(self.state[1:])
        return np.array(self.state)

    def _finish_step(self):
        import torch

        self._pull()
        o = self._h
        mean_sinr = float(o["mean_sinr_f64"][0])
        n_out = int(o["n_out"][0])
        r_dissect = [mean_sinr / 20, -1.0 * n_out / self.nUE]          # mobile_env.py:163-167
        reward = float(o["reward_f64"][0])                             # max(sum(r_dissect), -1)  (:189)
        done = bool(o["done"][0])
        self.association_map = np.array(self.state[1:])
        return r_dissect, reward, done, n_out

    def step(self, action, ifrender=False):
        """mobile_env.py:150-194 -> (state, reward, done, [r_dissect, step_n])."""
        import torch

        a = self._action(action)
        if self.mobility_model == "group":
            self._env.step(self._action_dev(a), **self._draws("step"))
        else:  # the reference's step() would call next(self.mm) on an empty list here; replay the trace instead
            self._env.step_trace(self._action_dev(a), self._trace_row(self.step_n), fading=self._draws("step").get("fading"))
        r_dissect, reward, done, _ = self._finish_step()
        return np.array(self.state), reward, done, [r_dissect, self.step_n]

    def step_test(self, action, ifrender=False):
        """mobile_env.py:196-233 -> (state, reward, done, info_tup)."""
        import torch

        a = self._action(action)
        at = self._action_dev(a)
        if self.mobility_model == "read_trace":
            self._env.step_trace(at, self._trace_row(self.step_n), fading=self._draws("step").get("fading"))  # :202-203
        else:
            self._env.step(at, **self._draws("step"))
        r_dissect, reward, done, n_out = self._finish_step()
        info = info_tup(r_dissect, self.step_n, self.ueLoc, self.bsLoc, (1.0 * n_out) / self.nUE, self._digits(a))
        return np.array(self.state), reward, done, info

    def render(self):
        raise NotImplementedError("matplotlib rendering (mobile_env.py:236-245) is outside the hot path")

    def plot_sinr_map(self):
        raise NotImplementedError("matplotlib rendering (mobile_env.py:247-268) is outside the hot path")

    # ---- copy.deepcopy(env) (gradient.py:15) ----------------------------------------------------------------------
    def __deepcopy__(self, memo):
        twin = MobiEnvironment.__new__(MobiEnvironment)
        MobiEnvironment.__init__(twin, self.nBS, self.nUE, self.grid_n, self.mobility_model,
                                 self.ueLoc_trace if self.mobility_model == "read_trace" else "",
                                 _env=self._env.clone())
        twin.bs_h = self.bs_h
        twin._draw_hook = self._draw_hook
        twin.bsLoc, twin.ueLoc = self.bsLoc.copy(), self.ueLoc.copy()
        twin.state, twin.step_n = self.state.copy(), self.step_n
        twin.channel.current_BS = self.channel.current_BS.copy()
        twin.channel.current_BS_sinr = self.channel.current_BS_sinr.copy()
        for name in ("association", "association_map"):
            if hasattr(self, name):
                setattr(twin, name, getattr(self, name).copy())
        return twin
