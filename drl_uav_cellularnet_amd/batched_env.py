"""BatchedMobiEnv: N independent UAV-cellular environments stepped by one HIP kernel launch.

Host-side mirror of the reference's ``MobiEnvironment`` (mobile_env.py:35-194) for a batch:
same constructor meaning (nBS, nUE, grid_n), same ``reset()`` / ``step(action)`` /
``step_test(action)`` semantics per env, tensors instead of scalars.  PyTorch is plumbing only
(device memory + streams); all computation happens in libuavenv.so (csrc/uavenv_kernels.h).

    env = BatchedMobiEnv(4096, nBS=4, nUE=20, grid_n=100)       # ctor = init + 200 warm-up ticks + reset
    obs = env.reset()                                          # compact obs: dict of device tensors
    obs, reward, done, info = env.step(actions)                # actions: int64 [N] in [0, 5**nBS)
    dense = env.dense_obs()                                    # (N, nBS+1, G, G) float32, reference layout
"""
import ctypes as C

import numpy as np
import torch

from . import _capi
from ._capi import UavEnvError  # noqa: F401  (re-export)

_OUT_SPECS = {
    # name: (dtype, shape-builder)
    "reward": (torch.float32, lambda N, U, B: (N,)),
    "done": (torch.uint8, lambda N, U, B: (N,)),
    "mean_sinr": (torch.float32, lambda N, U, B: (N,)),
    "n_out": (torch.int32, lambda N, U, B: (N,)),
    "ue_xy": (torch.int16, lambda N, U, B: (N, U, 2)),
    "bs_xy": (torch.int32, lambda N, U, B: (N, B, 2)),
    "serving": (torch.int8, lambda N, U, B: (N, U)),
    "cur_sinr": (torch.float32, lambda N, U, B: (N, U)),
    "step_n": (torch.int32, lambda N, U, B: (N,)),
    "cur_sinr_f64": (torch.float64, lambda N, U, B: (N, U)),
    "mean_sinr_f64": (torch.float64, lambda N, U, B: (N,)),
    "reward_f64": (torch.float64, lambda N, U, B: (N,)),
}

_NP_DTYPES = {torch.float32: np.float32, torch.float64: np.float64, torch.uint8: np.uint8, torch.int8: np.int8,
              torch.int16: np.int16, torch.int32: np.int32}

# Record formats of the state blob (include/uavenv.h, csrc/state_layout.h), little-endian, no implicit padding.
_REC_DTYPES = {
    "ue_pos": np.dtype([("x", "<f8"), ("y", "<f8")]),
    "ue_aux": np.dtype([("hu", "<f8"), ("ix", "<i2"), ("iy", "<i2"), ("serving", "i1"), ("r0", "i1"), ("r1", "i1"), ("r2", "i1")]),
    "grp": np.dtype([("x", "<f8"), ("y", "<f8"), ("fl", "<f8"), ("v", "<f8"), ("c", "<f8"), ("s", "<f8")]),
    "env": np.dtype([("tick", "<u4"), ("agg", "<i4"), ("deagg", "<i4"), ("fifo_depth", "<i4"), ("step_n", "<i4"), ("pad", "<i4", (3,))]),
}
assert [_REC_DTYPES[k].itemsize for k in ("ue_pos", "ue_aux", "grp", "env")] == [16, 16, 48, 32]


class BatchedMobiEnv:
    N_ACT = 5  # mobile_env.py:21
    WARMUP_TICKS = 200  # mobile_env.py:77-79

    def __init__(self, n_envs, nBS=4, nUE=20, grid_n=100, groups=None, bs_init=None, device=None, seed=0x5EED,
                 env_id_base=0, f64_outputs=False, construct=True, _cfg=None, **config_overrides):
        if not torch.cuda.is_available():
            raise UavEnvError("BatchedMobiEnv needs a ROCm GPU (gfx950); there is no CPU fallback")
        self._lib = _capi.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.type != "cuda":
            raise UavEnvError("device must be a cuda/HIP device")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.n_envs, self.nBS, self.nUE, self.grid_n = int(n_envs), int(nBS), int(nUE), int(grid_n)
        self.seed, self.env_id_base = int(seed), int(env_id_base)
        self.cfg = _cfg if _cfg is not None else _capi.make_config(nBS, nUE, grid_n, groups=groups, bs_init=bs_init,
                                                                   **config_overrides)
        self.n_groups = int(self.cfg.n_groups)
        self.action_space_dim = self.N_ACT ** self.nBS                      # mobile_env.py:104
        self.observation_space_dim = self.grid_n * self.grid_n * (self.nBS + 1)  # mobile_env.py:105
        self._h = C.c_void_p()
        _capi.check(self._lib.uavenv_create(C.byref(self.cfg), self.n_envs, self.device.index, self.seed,
                                            self.env_id_base, C.byref(self._h)))
        N, U, B = self.n_envs, self.nUE, self.nBS
        # All output tensors are views of ONE device arena (each on a 256-byte boundary): out_host() then brings every output
        # to the host with a single device-to-host copy (the N = 1 shim reads them all after each step).
        self.out = {}
        self._out_struct = _capi.UavEnvOut()
        specs, off = [], 0
        for name, (dt, shp) in _OUT_SPECS.items():
            if name.endswith("_f64") and not f64_outputs:
                continue
            shape = shp(N, U, B)
            nbytes = int(np.prod(shape)) * torch.empty((), dtype=dt).element_size()
            specs.append((name, dt, shape, off, nbytes))
            off = (off + nbytes + 255) // 256 * 256
        self._arena = torch.zeros(max(off, 256), dtype=torch.uint8, device=self.device)
        self._arena_specs = specs
        self._host_arena = None
        for name, dt, shape, o, nbytes in specs:
            t = self._arena[o:o + nbytes].view(dt).view(shape)
            self.out[name] = t
            setattr(self._out_struct, name + "_dev", t.data_ptr())
        self._out_ref = C.byref(self._out_struct)
        self._lay = _capi.UavEnvStateLayout()
        _capi.check(self._lib.uavenv_state_layout(self._h, C.byref(self._lay)))
        self._keep = None
        self._many_structs = {}          # step_many: id(out dict) -> (dict, byref(UavEnvOut), T, struct)
        self._constructed = False
        if construct:
            self.construct()

    # ---- lifetime -------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.uavenv_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    # ---- injected randomness (parity tests) --------------------------------------------------------
    def _dev64(self, a, shape):
        if not isinstance(a, torch.Tensor):
            a = np.ascontiguousarray(a, dtype=np.float64)   # (a read-only broadcast view would make torch.as_tensor warn)
        t = torch.as_tensor(a, dtype=torch.float64).reshape(shape)
        return t.to(self.device).contiguous()

    def _inject(self, theta_u=None, group_u=None, fading=None):
        if theta_u is None and group_u is None and fading is None:
            return None
        N, U, B, Gr = self.n_envs, self.nUE, self.nBS, self.n_groups
        inj = _capi.UavEnvInject()
        keep = []
        for name, a, shape in (("theta_u_dev", theta_u, (N, U)), ("group_u_dev", group_u, (N, Gr, 3)),
                               ("fading_dev", fading, (N, U, B))):
            if a is not None:
                t = self._dev64(a, shape)
                keep.append(t)
                setattr(inj, name, t.data_ptr())
        self._keep = keep  # stream-ordered use: keep alive until the next call
        return C.byref(inj)

    # ---- constructor pieces (mobile_env.py:76-98) ----------------------------------------------------
    def init(self, u_x=None, u_y=None, u_th=None, u_g=None):
        """reference_point_group state construction (ue_mobility.py:433-451)."""
        inj = None
        if u_x is not None:
            N, U, Gr = self.n_envs, self.nUE, self.n_groups
            ts = [self._dev64(u_x, (N, U)), self._dev64(u_y, (N, U)), self._dev64(u_th, (N, U)),
                  self._dev64(u_g, (N, 5, Gr))]
            ii = _capi.UavEnvInitInject()
            ii.u_x_dev, ii.u_y_dev, ii.u_th_dev, ii.u_g_dev = (t.data_ptr() for t in ts)
            self._keep = ts
            inj = C.byref(ii)
        _capi.check(self._lib.uavenv_init(self._h, inj, self._stream()))

    def warmup(self, n_ticks=1, theta_u=None, group_u=None):
        """n_ticks x next(self.mm) with no channel update (mobile_env.py:77-79)."""
        _capi.check(self._lib.uavenv_warmup(self._h, int(n_ticks), self._inject(theta_u, group_u), self._stream()))

    def construct(self):
        """Everything MobiEnvironment.__init__ does: initial draws, 200 warm-up ticks, then the 201st tick +
        LTEChannel.__init__, which is exactly reset() with the UAVs already on their start cells."""
        self.init()
        self.warmup(self.WARMUP_TICKS)
        self._constructed = True
        return self.reset()

    # ---- gym-style API -----------------------------------------------------------------------------
    def reset(self, mask=None, theta_u=None, group_u=None, fading=None):
        """MobiEnvironment.reset (mobile_env.py:115-148) for every env, or those with mask[e] != 0."""
        mptr = None
        if mask is not None:
            mask = torch.as_tensor(mask).to(device=self.device, dtype=torch.uint8).contiguous()
            if mask.numel() != self.n_envs:
                raise ValueError("mask must have n_envs elements")
            self._mask_keep = mask
            mptr = mask.data_ptr()
        _capi.check(self._lib.uavenv_reset(self._h, mptr, self._inject(theta_u, group_u, fading), self._out_ref,
                                           self._stream()))
        return self.observation()

    def reset_trace(self, ue_xy, mask=None, fading=None):
        """reset() / constructor tail with mobility_model == 'read_trace' (mobile_env.py:85-89,128-131):
        UE cells = ``ue_xy`` [N, U, 2] (row 0 of the trace), no mobility tick."""
        x = torch.as_tensor(ue_xy).to(device=self.device, dtype=torch.int16).contiguous()
        if x.numel() != self.n_envs * self.nUE * 2:
            raise ValueError("ue_xy must be [N, U, 2]")
        self._trace_keep = x
        mptr = None
        if mask is not None:
            mask = torch.as_tensor(mask).to(device=self.device, dtype=torch.uint8).contiguous()
            if mask.numel() != self.n_envs:
                raise ValueError("mask must have n_envs elements")
            self._mask_keep = mask
            mptr = mask.data_ptr()
        _capi.check(self._lib.uavenv_reset_trace(self._h, mptr, x.data_ptr(), self._inject(None, None, fading),
                                                 self._out_ref, self._stream()))
        return self.observation()

    def step(self, actions, theta_u=None, group_u=None, fading=None, reward_out=None):
        """MobiEnvironment.step (mobile_env.py:150-194): returns (obs, reward, done, info).

        ``actions``: int64 tensor [N] on this device, each in [0, 5**nBS) (base-5 digits, most significant
        digit -> UAV 0, ue_mobility.py:310-336).  Outputs are persistent device tensors, overwritten by the
        next call (copy what must survive).  No auto-reset, as in the reference."""
        a = self._actions(actions)
        inj = None if (theta_u is None and group_u is None and fading is None) else self._inject(theta_u, group_u,
                                                                                                 fading)
        out_ref = self._out_ref
        if reward_out is not None:       # this step's reward straight into the caller's rollout buffer (float32 [N], contiguous)
            if reward_out.dtype != torch.float32 or reward_out.numel() != self.n_envs or not reward_out.is_contiguous():
                raise ValueError("reward_out must be a contiguous float32 [n_envs] tensor")
            st = _capi.UavEnvOut.from_buffer_copy(self._out_struct)
            st.reward_dev = reward_out.data_ptr()
            out_ref = C.byref(st)
        rc = self._lib.uavenv_step(self._h, a.data_ptr(), inj, out_ref, self._stream())
        if rc:
            _capi.check(rc)
        o = self.out
        if reward_out is not None:
            return self.observation(), reward_out, o["done"], {"mean_sinr": o["mean_sinr"], "n_out": o["n_out"],
                                                               "step_n": o["step_n"], "cur_sinr": o["cur_sinr"]}
        return self.observation(), o["reward"], o["done"], {"mean_sinr": o["mean_sinr"], "n_out": o["n_out"],
                                                             "step_n": o["step_n"], "cur_sinr": o["cur_sinr"]}

    @property
    def envs_per_wavefront(self):
        """Env instances one wavefront of the step kernel hosts."""
        packed = self.nUE <= 64 and self.nUE >= self.nBS and self.nUE >= int(self.cfg.n_groups)
        return min(64 // self.nUE, 8) if packed else 1

    def step_range(self, actions, first_env, n_envs, reward_out=None):
        """step() for envs [first_env, first_env + n_envs) only (uavenv_step_range).  ``actions`` / ``reward_out`` are the WHOLE batch's
        [N] tensors; only the range's entries are read / written, and only the range's slices of ``self.out`` change.  Ranges that do
        not overlap may be stepped concurrently on different streams.  No return value: read ``self.out`` / observation()."""
        a = self._actions(actions)
        out_ref = self._out_ref
        if reward_out is not None:
            if reward_out.dtype != torch.float32 or reward_out.numel() != self.n_envs or not reward_out.is_contiguous():
                raise ValueError("reward_out must be a contiguous float32 [n_envs] tensor")
            st = _capi.UavEnvOut.from_buffer_copy(self._out_struct)
            st.reward_dev = reward_out.data_ptr()
            out_ref = C.byref(st)
        rc = self._lib.uavenv_step_range(self._h, a.data_ptr(), int(first_env), int(n_envs), None, out_ref, self._stream())
        if rc:
            _capi.check(rc)

    GATE_ROWS = 16          # UAVENV_GATE_ROWS: envs per gate word of rollout_gated

    def rollout_gated(self, actions, gate_actions, gate_obs, claim, table_a, bias_a, out_a, table_c=None, bias_c=None, out_c=None, idx_out=None,
                      reward_out=None, relu6=True):
        """uavenv_rollout_gated: T = len(actions) steps in ONE persistent launch that takes its actions from a policy kernel running
        beside it on another stream (include/uavenv.h has the protocol).  ``actions`` int64 [T, N] (written by the policy while this
        launch runs), ``gate_actions`` / ``gate_obs`` int32 [ceil(N / 16)] step counters, ``claim`` int32 [1] (zero before the launch), tables float32 [n_rows, hidden], outputs
        float32 [T, N, hidden] (slot t + 1 written after step t), ``idx_out`` int64 [T + 1, N, B + U], ``reward_out`` float32 [T, N].
        Asynchronous; a partner that never arrives ends in device_error() != 0, not in a hang."""
        T, N = int(actions.shape[0]), self.n_envs
        nb = (N + self.GATE_ROWS - 1) // self.GATE_ROWS

        def chk(t, dtype, shape, what):
            if not (isinstance(t, torch.Tensor) and t.dtype == dtype and t.device == self.device and t.is_contiguous() and tuple(t.shape) == tuple(shape)):
                raise ValueError("%s must be a contiguous %s %s tensor on the env's device" % (what, dtype, list(shape)))
        chk(actions, torch.int64, (T, N), "actions")
        chk(gate_actions, torch.int32, (nb,), "gate_actions")
        chk(gate_obs, torch.int32, (nb,), "gate_obs")
        chk(claim, torch.int32, (1,), "claim")
        rows, hid = int(table_a.shape[0]), int(table_a.shape[1])
        chk(table_a, torch.float32, (rows, hid), "table_a")
        chk(out_a, torch.float32, (T, N, hid), "out_a")
        if bias_a is not None:
            chk(bias_a, torch.float32, (hid,), "bias_a")
        if table_c is not None:
            chk(table_c, torch.float32, (rows, hid), "table_c")
            chk(out_c, torch.float32, (T, N, hid), "out_c")
            if bias_c is not None:
                chk(bias_c, torch.float32, (hid,), "bias_c")
        if idx_out is not None:
            chk(idx_out, torch.int64, (T + 1, N, self.nBS + self.nUE), "idx_out")
        if reward_out is not None:
            chk(reward_out, torch.float32, (T, N), "reward_out")
        r = _capi.UavEnvGatedRollout()
        r.n_steps = T
        r.actions_dev, r.gate_actions_dev, r.gate_obs_dev, r.claim_dev = actions.data_ptr(), gate_actions.data_ptr(), gate_obs.data_ptr(), claim.data_ptr()
        r.reward_dev = reward_out.data_ptr() if reward_out is not None else None
        r.enc_table_a_dev, r.enc_out_a_dev = table_a.data_ptr(), out_a.data_ptr()
        r.enc_bias_a_dev = bias_a.data_ptr() if bias_a is not None else None
        r.enc_table_c_dev = table_c.data_ptr() if table_c is not None else None
        r.enc_bias_c_dev = bias_c.data_ptr() if (table_c is not None and bias_c is not None) else None
        r.enc_out_c_dev = out_c.data_ptr() if table_c is not None else None
        r.idx_out_dev = idx_out.data_ptr() if idx_out is not None else None
        r.enc_rows, r.enc_hidden, r.enc_relu6 = rows, hid, 1 if relu6 else 0
        rc = self._lib.uavenv_rollout_gated(self._h, C.byref(r), self._out_ref, self._stream())
        if rc:
            _capi.check(rc)

    def capture_steps(self, actions):
        """A hipGraph of len(actions) step() launches (one kernel node per step, step t reading ``actions[t]``): ``g.replay()``
        then costs one graph launch instead of T host calls -- the per-step host cost (ctypes call + hipLaunchKernel, ~8 us
        from Python, tools/host_floor.py) is what bounds a 4096-env batch, not the kernel.  Outputs go to ``self.out`` as
        with step(), so after a replay they hold the last step's results.  ``actions`` int64 [T, N] on this device; the graph
        reads that tensor at replay time (refill it in place to feed new actions).  Capturing executes nothing."""
        a = actions
        if not (isinstance(a, torch.Tensor) and a.dtype == torch.int64 and a.device == self.device and a.is_contiguous()
                and a.dim() == 2 and a.shape[1] == self.n_envs):
            raise ValueError("actions must be a contiguous int64 [T, n_envs] tensor on the env's device")
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):   # (other threads, e.g. RCCL's watchdog, may poll events meanwhile)
            stream = self._stream()                      # the capturing stream
            for t in range(int(a.shape[0])):
                _capi.check(self._lib.uavenv_step(self._h, a[t].data_ptr(), None, self._out_ref, stream))
        g._uavenv_keep = (a, self)                       # the graph holds raw pointers into both
        return g

    def step_seq(self, actions):
        """len(actions) step() launches issued by ONE C call (uavenv_step_seq): one kernel per step like step(), without the
        per-step Python -> ctypes round trip.  ``actions`` int64 [T, N] on this device; ``self.out`` holds the last step's
        results afterwards, as after T step() calls."""
        a = actions
        if not (isinstance(a, torch.Tensor) and a.dtype == torch.int64 and a.device == self.device and a.is_contiguous()
                and a.dim() == 2 and a.shape[1] == self.n_envs):
            raise ValueError("actions must be a contiguous int64 [T, n_envs] tensor on the env's device")
        _capi.check(self._lib.uavenv_step_seq(self._h, a.data_ptr(), int(a.shape[0]), self._out_ref, self._stream()))
        o = self.out
        return self.observation(), o["reward"], o["done"], {"mean_sinr": o["mean_sinr"], "n_out": o["n_out"],
                                                             "step_n": o["step_n"], "cur_sinr": o["cur_sinr"]}

    def out_struct_for(self, out):
        """A UavEnvOut whose members point at the tensors of ``out`` (a dict with this env's output names): for callers that bind
        uavenv_step_many / uavenv_step_seq themselves.  The caller keeps both alive."""
        st = _capi.UavEnvOut()
        for k in self.out:
            setattr(st, k + "_dev", out[k].data_ptr())
        return st

    def device_error(self):
        """Sticky device-side error word of the handle (uavenv_device_error): 0, or the code a kernel left when it gave up waiting for
        a hand-off of a one-launch rotation schedule.  Meaningful after the launch's stream has been synchronised; while it is
        non-zero every call on this env raises UavEnvError (UAVENV_E_DEVICE) until set_state() installs a whole state again."""
        code = C.c_uint32(0)
        _capi.check(self._lib.uavenv_device_error(self._h, C.byref(code)))
        return int(code.value)

    def launch_timing(self, enable=True):
        """Attach start / stop events to the following multi-step dispatches themselves (uavenv_launch_timing); see launch_times_us()."""
        _capi.check(self._lib.uavenv_launch_timing(self._h, 1 if enable else 0))

    def launch_times_us(self):
        """Durations (us) of the step_many launches since launch_timing(True), in issue order (waits for them)."""
        buf = (C.c_double * 256)()
        n = C.c_int(0)
        _capi.check(self._lib.uavenv_launch_times_us(self._h, buf, 256, C.byref(n)))
        return [buf[i] for i in range(min(n.value, 256))]

    def step_many(self, actions, out=None, refresh_out=True):
        """T consecutive step() calls in ONE launch (uavenv_step_many) for actions that do not depend on the observations in
        between: ``actions`` int64 [T, N] on this device.  Returns a dict of [T, ...] tensors (block t = what step t returned;
        same names and dtypes as ``self.out``); ``self.out`` is then refreshed with the last step's block (``refresh_out``), so
        observation() and the step()/reset() API continue from there.  Bit-identical to T step() calls.  ``out``: a dict previously returned for
        the same T, to be overwritten instead of allocating."""
        a = actions
        if not (isinstance(a, torch.Tensor) and a.dtype == torch.int64 and a.device == self.device and a.is_contiguous()):
            a = torch.as_tensor(actions).to(device=self.device, dtype=torch.int64).contiguous()
            self._act_keep = a
        if a.dim() != 2 or a.shape[1] != self.n_envs:
            raise ValueError("actions must be [T, n_envs]")
        T = int(a.shape[0])
        if out is None:
            out = {k: torch.empty((T,) + tuple(v.shape), dtype=v.dtype, device=self.device) for k, v in self.out.items()}
        cached = self._many_structs.get(id(out))
        if cached is None or cached[0] is not out or cached[2] != T:       # validate and build the pointer struct once per dict
            if set(out) != set(self.out) or any(out[k].shape != (T,) + tuple(v.shape) or not out[k].is_contiguous()
                                                for k, v in self.out.items()):
                raise ValueError("out must be a dict returned by step_many for the same number of steps")
            st = _capi.UavEnvOut()
            for k, v in out.items():
                setattr(st, k + "_dev", v.data_ptr())
            if len(self._many_structs) > 16:
                self._many_structs.clear()
            cached = self._many_structs[id(out)] = (out, C.byref(st), T, st)
        rc = self._lib.uavenv_step_many(self._h, a.data_ptr(), T, cached[1], self._stream())
        if rc:
            _capi.check(rc)
        if T > 0 and refresh_out:                        # (refresh_out=False: self.out goes stale until the next step()/reset())
            for k, v in self.out.items():
                v.copy_(out[k][T - 1])
        return out

    def prepare_step_many(self, n_steps):
        """Build whatever a step_many call of ``n_steps`` steps needs ahead of time (the launch schedule of the
        4096-env batch, uavenv_step_many_prepare): the first call with a new n_steps would otherwise build it, synchronously."""
        _capi.check(self._lib.uavenv_step_many_prepare(self._h, int(n_steps)))

    def step_trace(self, actions, ue_xy, fading=None):
        """MobiEnvironment.step_test with mobility_model == 'read_trace' (mobile_env.py:196-233)."""
        a = self._actions(actions)
        x = torch.as_tensor(ue_xy).to(device=self.device, dtype=torch.int16).contiguous()
        if x.numel() != self.n_envs * self.nUE * 2:
            raise ValueError("ue_xy must be [N, U, 2]")
        self._trace_keep = x
        _capi.check(self._lib.uavenv_step_trace(self._h, a.data_ptr(), x.data_ptr(), self._inject(None, None, fading),
                                                self._out_ref, self._stream()))
        o = self.out
        return self.observation(), o["reward"], o["done"], {"mean_sinr": o["mean_sinr"], "n_out": o["n_out"],
                                                             "step_n": o["step_n"], "cur_sinr": o["cur_sinr"]}

    def _actions(self, actions):
        a = actions
        if not (isinstance(a, torch.Tensor) and a.dtype == torch.int64 and a.device == self.device
                and a.is_contiguous()):
            a = torch.as_tensor(actions).to(device=self.device, dtype=torch.int64).contiguous()
            self._act_keep = a
        if a.numel() != self.n_envs:
            raise ValueError("actions must have n_envs elements")
        return a

    def out_host(self):
        """Every output of the last reset / step as NumPy views of one pinned host buffer: one D2H copy, one synchronisation.
        The views are overwritten by the next call."""
        if self._host_arena is None:
            self._host_arena = torch.empty(self._arena.shape, dtype=torch.uint8, pin_memory=True)
            self._host_np = self._host_arena.numpy()
            self._host_views = {name: self._host_np[o:o + nbytes].view(_NP_DTYPES[dt]).reshape(shape)
                                for name, dt, shape, o, nbytes in self._arena_specs}
        self._host_arena.copy_(self._arena, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        return self._host_views

    def observation(self):
        """Compact observation: the ~(U+B) non-zero cells of the reference's state tensor."""
        o = self.out
        return {"ue_xy": o["ue_xy"], "bs_xy": o["bs_xy"], "serving": o["serving"]}

    def dense_obs(self, out=None):
        """env.state for every env: float32 [N, nBS+1, G, G] (plane 0 = UAV cells, plane 1+b = UEs served by b)."""
        N, B, G = self.n_envs, self.nBS, self.grid_n
        if out is None:
            out = torch.empty((N, B + 1, G, G), dtype=torch.float32, device=self.device)
        elif out.dtype != torch.float32 or out.numel() != N * (B + 1) * G * G or not out.is_contiguous():
            raise ValueError("out must be contiguous float32 [N, nBS+1, G, G]")
        _capi.check(self._lib.uavenv_obs_dense(self._h, out.data_ptr(), self._stream()))
        return out

    def dense_obs_update(self, buf):
        """Bring ``buf`` (the tensor the previous dense_obs / dense_obs_update call of this env wrote) up to date IN PLACE:
        only the <= nUE + nBS cells per env that changed are touched, instead of rewriting N*(nBS+1)*G*G floats."""
        N, B, G = self.n_envs, self.nBS, self.grid_n
        if buf.dtype != torch.float32 or buf.numel() != N * (B + 1) * G * G or not buf.is_contiguous():
            raise ValueError("buf must be the contiguous float32 [N, nBS+1, G, G] tensor dense_obs returned")
        _capi.check(self._lib.uavenv_obs_dense_update(self._h, buf.data_ptr(), self._stream()))
        return buf

    def sinr_area(self, fading=None, dtype=torch.float32, bs_xy=None):
        """LTEChannel.GetSinrInArea (channel.py:411-433) for every env: [N, G, G] dB, nearest-UAV SINR per cell with
        fresh shadowing (row / column 0 are 0, as in the reference).  ``fading``: injected draws [N, (G-1)^2, B].
        ``bs_xy``: any UAV cells [N, B, 2] (the reference's ``bsLoc`` argument); default = the env's current cells."""
        N, B, G = self.n_envs, self.nBS, self.grid_n
        out = torch.empty((N, G, G), dtype=dtype, device=self.device)
        bptr = None
        if bs_xy is not None:
            if not isinstance(bs_xy, torch.Tensor):
                bs_xy = np.array(bs_xy, dtype=np.int32)            # (a copy: a read-only broadcast view would make torch.as_tensor warn)
            b = torch.as_tensor(bs_xy).to(device=self.device, dtype=torch.int32).contiguous()
            if b.numel() != N * B * 2:
                raise ValueError("bs_xy must be [n_envs, nBS, 2]")
            self._bs_keep = b
            bptr = b.data_ptr()
        fptr = None
        if fading is not None:
            f = self._dev64(fading, (N, (G - 1) * (G - 1), B))
            self._keep = [f]
            fptr = f.data_ptr()
        o32 = out.data_ptr() if dtype == torch.float32 else None
        o64 = out.data_ptr() if dtype == torch.float64 else None
        if o32 is None and o64 is None:
            raise ValueError("dtype must be torch.float32 or torch.float64")
        _capi.check(self._lib.uavenv_sinr_area_at(self._h, bptr, fptr, o32, o64, self._stream()))
        return out

    # ---- state blob: copy.deepcopy(env) (gradient.py:15) / checkpoint --------------------------------
    def get_state(self):
        """Whole persistent state as one host uint8 array (synchronises)."""
        buf = np.empty(self._lay.total_bytes, np.uint8)
        _capi.check(self._lib.uavenv_get_state(self._h, buf.ctypes.data, 0, self._stream()))
        return buf

    def set_state(self, blob):
        blob = np.ascontiguousarray(blob, np.uint8)
        if blob.size != self._lay.total_bytes:
            raise ValueError("state blob has the wrong size")
        _capi.check(self._lib.uavenv_set_state(self._h, blob.ctypes.data, 0, self._stream()))

    def copy_state_to(self, dev_buf):
        """Device-to-device snapshot of the state blob into a uint8 tensor of _lay.total_bytes (stream-ordered, no sync)."""
        _capi.check(self._lib.uavenv_get_state(self._h, dev_buf.data_ptr(), 1, self._stream()))

    def copy_state_from(self, dev_buf):
        _capi.check(self._lib.uavenv_set_state(self._h, dev_buf.data_ptr(), 1, self._stream()))

    def state_fields(self, blob=None):
        """The state blob decoded into named arrays (record formats: UavEnvStateLayout in include/uavenv.h).  Field views
        alias the blob except ``fifo`` and ``ue_xy``, which are assembled from record members."""
        blob = self.get_state() if blob is None else blob
        N, U, B, Gr = self.n_envs, self.nUE, self.nBS, self.n_groups
        W64 = (U + 63) // 64
        def rec(name, shape):
            dt, off = _REC_DTYPES[name], getattr(self._lay, name)
            return blob[off:off + int(np.prod(shape)) * dt.itemsize].view(dt).reshape(shape)

        def plain(name, dtype, shape):
            dt, off = np.dtype(dtype), getattr(self._lay, name)
            return blob[off:off + int(np.prod(shape)) * dt.itemsize].view(dt).reshape(shape)

        pos, aux, grp, env = rec("ue_pos", (N, U)), rec("ue_aux", (N, U)), rec("grp", (N, Gr)), rec("env", (N,))
        views = {"ue_x": pos["x"], "ue_y": pos["y"], "ue_hu": aux["hu"],
                 "g_x": grp["x"], "g_y": grp["y"], "g_fl": grp["fl"], "g_v": grp["v"], "g_cos": grp["c"], "g_sin": grp["s"],
                 "agg": env["agg"], "deagg": env["deagg"], "tick": env["tick"], "fifo_depth": env["fifo_depth"], "step_n": env["step_n"],
                 "bs_xy": plain("bs_xy", np.int32, (N, B, 2)), "serving": aux["serving"],
                 "fifo": np.stack([aux["r0"], aux["r1"], aux["r2"]], axis=1),          # [N, 3, U], oldest row first (a copy)
                 "out_bits": plain("out_bits", np.uint64, (N, W64)),
                 "ue_xy": np.stack([aux["ix"], aux["iy"]], axis=-1)}                   # [N, U, 2] (a copy)
        return views

    def clone(self):
        """Independent copy (same config, seed and state), as copy.deepcopy(env) gives in the reference."""
        other = BatchedMobiEnv.__new__(BatchedMobiEnv)
        BatchedMobiEnv.__init__(other, self.n_envs, self.nBS, self.nUE, self.grid_n, device=self.device,
                                seed=self.seed, env_id_base=self.env_id_base,
                                f64_outputs="cur_sinr_f64" in self.out, construct=False, _cfg=self.cfg)
        tmp = torch.empty(self._lay.total_bytes, dtype=torch.uint8, device=self.device)
        _capi.check(self._lib.uavenv_get_state(self._h, tmp.data_ptr(), 1, self._stream()))
        _capi.check(other._lib.uavenv_set_state(other._h, tmp.data_ptr(), 1, self._stream()))
        for k, v in self.out.items():
            other.out[k].copy_(v)
        other._constructed = self._constructed
        torch.cuda.current_stream(self.device).synchronize()
        return other
