"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  It wraps ``oracle/libuavenv_oracle.so`` (built from
``uavenv_oracle.c`` by ``oracle/Makefile``), the scalar float64 restatement of
/root/reference ``mobile_env.py`` / ``channel.py`` / ``ue_mobility.py``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libuavenv_oracle.so")
MAX_GROUPS, MAX_BS = 16, 32


class UavoConfig(C.Structure):
    _fields_ = [
        ("n_bs", C.c_int32), ("n_ue", C.c_int32), ("n_groups", C.c_int32), ("grid", C.c_int32),
        ("group_size", C.c_int32 * MAX_GROUPS), ("bs_init_xy", (C.c_int32 * 2) * MAX_BS),
        ("max_step", C.c_int32), ("bs_step", C.c_int32), ("min_bs_dist", C.c_int32), ("n_act", C.c_int32),
        ("agg_init", C.c_int32), ("deagg_len", C.c_int32), ("agg_len", C.c_int32), ("_pad", C.c_int32),
        ("grid_width", C.c_double), ("p_bs_dbm", C.c_double), ("noise_dbm", C.c_double),
        ("pl_a", C.c_double), ("pl_b", C.c_double), ("pl_dis", C.c_double),
        ("antenna_gain", C.c_double), ("eq_loss", C.c_double),
        ("shadow_mean", C.c_double), ("shadow_sd", C.c_double),
        ("ho_thresh_db", C.c_double), ("out_thresh", C.c_double),
        ("ue_velocity", C.c_double), ("grp_v_min", C.c_double), ("grp_v_max", C.c_double),
        ("aggregation", C.c_double),
    ]


_P = C.c_void_p


class UavoState(C.Structure):
    _fields_ = [("n_envs", C.c_int64), ("seed", C.c_uint64), ("env_id_base", C.c_uint32), ("_pad", C.c_uint32)] + [
        (n, _P) for n in ("ue_x", "ue_y", "ue_hu", "g_x", "g_y", "g_fl", "g_v", "g_cos", "g_sin", "agg", "deagg",
                          "tick", "bs_xy", "serving", "fifo", "fifo_depth", "out_bits", "step_n", "ue_xy")]


class UavoInitInject(C.Structure):
    _fields_ = [(n, _P) for n in ("u_x", "u_y", "u_th", "u_g")]


class UavoInject(C.Structure):
    _fields_ = [(n, _P) for n in ("theta_u", "group_u", "fading")]


class UavoOut(C.Structure):
    _fields_ = [(n, _P) for n in ("reward", "done", "mean_sinr", "n_out", "ue_xy", "bs_xy", "serving", "cur_sinr",
                                  "step_n", "cur_sinr_f64", "mean_sinr_f64", "reward_f64")]


_lib = None


def build(force=False):
    if force or not os.path.isfile(_SO) or os.path.getmtime(_SO) < os.path.getmtime(
            os.path.join(_HERE, "uavenv_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.uavo_np_pairwise_sum.restype = C.c_double
        _lib.uavo_np_pairwise_sum.argtypes = [_P, C.c_int64]
    return _lib


def make_config(n_bs, n_ue, grid, groups=None, bs_init=None, **over):
    cfg = UavoConfig()
    lib().uavo_default_config(C.byref(cfg), int(n_bs), int(n_ue), int(grid))
    if groups is not None:
        groups = [int(g) for g in groups]
        assert sum(groups) == n_ue and len(groups) <= MAX_GROUPS
        cfg.n_groups = len(groups)
        for i in range(MAX_GROUPS):
            cfg.group_size[i] = groups[i] if i < len(groups) else 0
    if bs_init is not None:
        bs_init = np.asarray(bs_init).reshape(n_bs, 2)
        for b in range(n_bs):
            cfg.bs_init_xy[b][0] = int(bs_init[b, 0])
            cfg.bs_init_xy[b][1] = int(bs_init[b, 1])
    elif n_bs != 4:
        raise ValueError("bs_init required for n_bs != 4 (mobile_env.py:49-50 only lays out 4 UAVs)")
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_P)


class OracleEnv:
    """N independent envs stepped by the scalar C oracle.  State lives in numpy arrays (``self.s``)."""

    STATE_FIELDS = ("ue_x", "ue_y", "ue_hu", "g_x", "g_y", "g_fl", "g_v", "g_cos", "g_sin", "agg", "deagg", "tick",
                    "bs_xy", "serving", "fifo", "fifo_depth", "out_bits", "step_n", "ue_xy")

    def __init__(self, cfg, n_envs, seed=0x5EED, env_id_base=0):
        self.cfg, self.N = cfg, int(n_envs)
        N, U, B, Gr = self.N, cfg.n_ue, cfg.n_bs, cfg.n_groups
        self.U, self.B, self.Gr, self.W64 = U, B, Gr, (U + 63) // 64
        f8, i4 = np.float64, np.int32
        self.s = {
            "ue_x": np.zeros((N, U), f8), "ue_y": np.zeros((N, U), f8), "ue_hu": np.zeros((N, U), f8),
            "g_x": np.zeros((N, Gr), f8), "g_y": np.zeros((N, Gr), f8), "g_fl": np.zeros((N, Gr), f8),
            "g_v": np.zeros((N, Gr), f8), "g_cos": np.zeros((N, Gr), f8), "g_sin": np.zeros((N, Gr), f8),
            "agg": np.zeros(N, i4), "deagg": np.zeros(N, i4), "tick": np.zeros(N, np.uint32),
            "bs_xy": np.zeros((N, B, 2), i4), "serving": np.zeros((N, U), np.int8),
            "fifo": np.zeros((N, 3, U), np.int8), "fifo_depth": np.zeros(N, i4),
            "out_bits": np.zeros((N, self.W64), np.uint64), "step_n": np.zeros(N, i4),
            "ue_xy": np.zeros((N, U, 2), np.int16),
        }
        self.o = {
            "reward": np.zeros(N, np.float32), "done": np.zeros(N, np.uint8), "mean_sinr": np.zeros(N, np.float32),
            "n_out": np.zeros(N, i4), "ue_xy": np.zeros((N, U, 2), np.int16), "bs_xy": np.zeros((N, B, 2), i4),
            "serving": np.zeros((N, U), np.int8), "cur_sinr": np.zeros((N, U), np.float32),
            "step_n": np.zeros(N, i4), "cur_sinr_f64": np.zeros((N, U), f8), "mean_sinr_f64": np.zeros(N, f8),
            "reward_f64": np.zeros(N, f8),
        }
        self.st = UavoState()
        self.st.n_envs, self.st.seed, self.st.env_id_base = N, int(seed), int(env_id_base)
        for k in self.STATE_FIELDS:
            setattr(self.st, k, _ptr(self.s[k]))
        self.out = UavoOut()
        for k, v in self.o.items():
            setattr(self.out, k, _ptr(v))
        self._keep = None

    # -- helpers --------------------------------------------------------------------------
    def _inject(self, theta_u=None, group_u=None, fading=None):
        if theta_u is None and group_u is None and fading is None:
            return None
        arrs = []
        inj = UavoInject()
        for name, a, shape in (("theta_u", theta_u, (self.N, self.U)), ("group_u", group_u, (self.N, self.Gr, 3)),
                               ("fading", fading, (self.N, self.U, self.B))):
            if a is not None:
                a = np.ascontiguousarray(np.asarray(a, np.float64).reshape(shape))
                arrs.append(a)
                setattr(inj, name, _ptr(a))
        self._keep = arrs
        return C.byref(inj)

    def init(self, u_x=None, u_y=None, u_th=None, u_g=None):
        inj = None
        if u_x is not None:
            a = [np.ascontiguousarray(np.asarray(v, np.float64).reshape(s)) for v, s in (
                (u_x, (self.N, self.U)), (u_y, (self.N, self.U)), (u_th, (self.N, self.U)),
                (u_g, (self.N, 5, self.Gr)))]
            ii = UavoInitInject()
            ii.u_x, ii.u_y, ii.u_th, ii.u_g = (_ptr(v) for v in a)
            self._keep = a
            inj = C.byref(ii)
        rc = lib().uavo_init(C.byref(self.cfg), C.byref(self.st), inj)
        if rc:
            raise ValueError("oracle: bad config")

    def warmup(self, **inj):
        lib().uavo_warmup(C.byref(self.cfg), C.byref(self.st), self._inject(**inj))

    def construct(self, warmup_ticks=200):
        """Philox-mode equivalent of MobiEnvironment.__init__ (mobile_env.py:76-98)."""
        self.init()
        for _ in range(warmup_ticks):
            self.warmup()
        return self.reset()

    def reset(self, mask=None, **inj):
        m = None if mask is None else np.ascontiguousarray(np.asarray(mask, np.uint8))
        lib().uavo_reset(C.byref(self.cfg), C.byref(self.st), _ptr(m), self._inject(**inj), C.byref(self.out))
        return self.o

    def step(self, actions, **inj):
        a = np.ascontiguousarray(np.asarray(actions, np.int64).reshape(self.N))
        lib().uavo_step(C.byref(self.cfg), C.byref(self.st), _ptr(a), self._inject(**inj), C.byref(self.out))
        return self.o

    def reset_trace(self, ue_xy, mask=None, **inj):
        m = None if mask is None else np.ascontiguousarray(np.asarray(mask, np.uint8))
        x = np.ascontiguousarray(np.asarray(ue_xy, np.int16).reshape(self.N, self.U, 2))
        lib().uavo_reset_trace(C.byref(self.cfg), C.byref(self.st), _ptr(m), _ptr(x), self._inject(**inj),
                               C.byref(self.out))
        return self.o

    def step_trace(self, actions, ue_xy, **inj):
        a = np.ascontiguousarray(np.asarray(actions, np.int64).reshape(self.N))
        x = np.ascontiguousarray(np.asarray(ue_xy, np.int16).reshape(self.N, self.U, 2))
        lib().uavo_step_trace(C.byref(self.cfg), C.byref(self.st), _ptr(a), _ptr(x), self._inject(**inj),
                              C.byref(self.out))
        return self.o

    def sinr_area(self, fading=None):
        """GetSinrInArea (channel.py:411-433) -> float64 [N, G, G]."""
        G, W = self.cfg.grid, self.cfg.grid - 1
        out = np.zeros((self.N, G, G), np.float64)
        f = None if fading is None else np.ascontiguousarray(np.asarray(fading, np.float64).reshape(self.N, W * W, self.B))
        lib().uavo_sinr_area(C.byref(self.cfg), C.byref(self.st), _ptr(f), _ptr(out))
        return out

    def obs_dense(self):
        G = self.cfg.grid
        obs = np.zeros((self.N, self.B + 1, G, G), np.float32)
        lib().uavo_obs_dense(C.byref(self.cfg), C.byref(self.st), _ptr(obs))
        return obs


def philox4x32_10(ctr, key):
    c = (C.c_uint32 * 4)(*[int(v) for v in ctr])
    k = (C.c_uint32 * 2)(*[int(v) for v in key])
    o = (C.c_uint32 * 4)()
    lib().uavo_philox4x32_10(c, k, o)
    return [int(v) for v in o]


def np_pairwise_sum(a):
    a = np.ascontiguousarray(a, np.float64)
    return float(lib().uavo_np_pairwise_sum(_ptr(a), a.size))
