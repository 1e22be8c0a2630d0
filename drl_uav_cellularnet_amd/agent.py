"""MLP actor-critic + synchronous A2C for the batched env, in PyTorch-ROCm (SURVEY.md section 8 f, row 1).

Restated from the reference's TF1 code, which cannot run here (no tensorflow): **parity unpinned** -- the tests
compare against NumPy restatements of the same formulas, not against TF outputs.
  network   main.py:143-156   two separate trunks  N_S -> 200 relu6 -> 200 relu6 -> {N_A softmax | 1}
            weights N(0, 0.1) (main.py:145), biases 0 (tf.layers.dense default)
  loss      main.py:64-74     td = v_target - v;  c_loss = mean(td^2);
                              a_loss = mean(-(log(p[a] + 1e-5) * stopgrad(td) + beta * H)),  H = -sum p log(p + 1e-5)
  optimiser main.py:300-301   two tf.train.RMSPropOptimizer(lr=1e-4): decay 0.9, momentum 0, epsilon 1e-10 INSIDE
                              the sqrt, mean-square accumulator initialised to ONES (TF1 semantics)
  rollout   a2c_single_thread.py:107-133,153-186   T-step rollouts, n-step returns with gamma = 0.9 bootstrapped
                              from v(s_T) (0 when done), one update per rollout over all workers' samples

MI355X-side design: the observation is a count map with at most U + B non-zero cells (mobile_env.py:169-170), so
the 50000 x 200 first layer is a gather-sum of weight rows over the COMPACT observation the env kernel already
writes (ue_xy, serving, bs_xy) -- no dense (N, B+1, G, G) tensor is ever materialised on the training path.
Gradients are synchronised with one flat all-reduce per update (RCCL over xGMI when launched one process per GPU).
"""

import torch
import torch.nn.functional as F

GAMMA = 0.9           # main.py:20
ENTROPY_BETA = 0.001  # main.py:21
LR_A = 1e-4           # main.py:22
LR_C = 1e-4           # main.py:23
HIDDEN = 200          # main.py:147-148


def obs_to_indices(obs, grid_n, n_bs):
    """Compact observation -> flat indices of the non-zero cells of the reference's raveled state
    (np.ravel of (nBS+1, G, G), main.py:190,202):  plane 0 = UAV cells, plane 1+b = UEs served by UAV b."""
    G = int(grid_n)
    ue = obs["ue_xy"].long()
    bs = obs["bs_xy"].long()
    srv = obs["serving"].long()
    ue_idx = (1 + srv) * (G * G) + ue[..., 0] * G + ue[..., 1]
    bs_idx = bs[..., 0] * G + bs[..., 1]
    # A walker exactly on x == G or y == G (measure zero; the reference raises IndexError there) has no cell in
    # the G x G planes: its entry becomes -1 = "no row", as obs_cell() does for the dense tensor, instead of aliasing the next
    # row / plane.  Both first-layer implementations skip negative indices.
    ue_ok = ((ue >= 0) & (ue < G)).all(dim=-1)
    ue_idx = torch.where(ue_ok, ue_idx, torch.full_like(ue_idx, -1))
    return torch.cat([bs_idx, ue_idx], dim=-1)          # [N, B + U]; duplicates add, like the count map


def first_layer_reference(idx, w_a, b_a, w_c=None, b_c=None):
    """The sparse first layer in plain PyTorch: (h_a, h_c), h = sum_k W[idx[:, k]] + b.  Reference implementation of the HIP
    kernel (tests/test_agent_kernel_gpu.py) and the path CPU tensors take."""
    keep = (idx >= 0).to(w_a.dtype)                     # -1 = "no row" (obs_to_indices): weight 0
    safe = idx.clamp(min=0)
    ha = F.embedding_bag(safe, w_a, per_sample_weights=keep, mode="sum") + b_a
    hc = None if w_c is None else F.embedding_bag(safe, w_c, per_sample_weights=keep, mode="sum") + b_c
    return ha, hc


def sparse_first_layer(idx, w_a, b_a, w_c=None, b_c=None):
    """x @ W + b of main.py:147-148,153 for the raveled state with its nBS + nUE non-zero entries given as row indices.
    CUDA tensors: one launch of libuavagent.so for both tables (they share idx), an error if the library is missing."""
    if idx.is_cuda:
        from ._agent_capi import sparse_first_layer_cuda

        return sparse_first_layer_cuda(idx, w_a, b_a, w_c, b_c)
    return first_layer_reference(idx, w_a, b_a, w_c, b_c)


class ACNet(torch.nn.Module):
    """Actor and critic trunks of main.py:143-156.  The first layers are stored as [N_S, 200] tables so that the
    sparse path is an embedding-bag sum; ``forward_dense`` is the textbook matmul on the raveled dense state."""

    def __init__(self, n_state, n_action, hidden=HIDDEN, seed=6):
        super().__init__()
        g = torch.Generator().manual_seed(int(seed))    # TENSOR_SEED = 6 (main.py:26); not TF's stream
        def w(*shape):
            return torch.nn.Parameter(torch.randn(*shape, generator=g) * 0.1)   # random_normal_initializer(0, .1)
        def b(n):
            return torch.nn.Parameter(torch.zeros(n))
        self.n_state, self.n_action = int(n_state), int(n_action)
        self.a_w1, self.a_b1 = w(n_state, hidden), b(hidden)
        self.a_w2, self.a_b2 = w(hidden, hidden), b(hidden)
        self.a_w3, self.a_b3 = w(hidden, n_action), b(n_action)
        self.c_w1, self.c_b1 = w(n_state, hidden), b(hidden)
        self.c_w2, self.c_b2 = w(hidden, hidden), b(hidden)
        self.c_w3, self.c_b3 = w(hidden, 1), b(1)

    def actor_params(self):
        return [self.a_w1, self.a_b1, self.a_w2, self.a_b2, self.a_w3, self.a_b3]

    def critic_params(self):
        return [self.c_w1, self.c_b1, self.c_w2, self.c_b2, self.c_w3, self.c_b3]

    def _heads(self, ha, hc):
        ha = F.relu6(F.relu6(ha) @ self.a_w2 + self.a_b2)
        hc = F.relu6(F.relu6(hc) @ self.c_w2 + self.c_b2)
        a_prob = torch.softmax(ha @ self.a_w3 + self.a_b3, dim=-1)
        v = hc @ self.c_w3 + self.c_b3
        return a_prob, v

    def forward(self, idx):
        """idx: int64 [M, K] flat indices of the non-zero cells (obs_to_indices)."""
        ha, hc = sparse_first_layer(idx, self.a_w1, self.a_b1, self.c_w1, self.c_b1)
        return self._heads(ha, hc)

    def actor_only(self, idx):
        ha, _ = sparse_first_layer(idx, self.a_w1, self.a_b1)
        ha = F.relu6(F.relu6(ha) @ self.a_w2 + self.a_b2)
        return torch.softmax(ha @ self.a_w3 + self.a_b3, dim=-1)

    def critic_only(self, idx):
        hc, _ = sparse_first_layer(idx, self.c_w1, self.c_b1)
        hc = F.relu6(F.relu6(hc) @ self.c_w2 + self.c_b2)
        return hc @ self.c_w3 + self.c_b3

    def forward_dense(self, s):
        """s: float [M, N_S], the raveled (nBS+1, G, G) state exactly as the reference feeds it (main.py:190)."""
        return self._heads(s @ self.a_w1 + self.a_b1, s @ self.c_w1 + self.c_b1)


def a2c_losses(a_prob, v, actions, v_target, beta=ENTROPY_BETA):
    """(a_loss, c_loss) of main.py:64-74."""
    td = v_target - v                                                     # :64
    c_loss = (td ** 2).mean()                                             # :66
    log_prob = torch.log(a_prob.gather(1, actions.view(-1, 1)) + 1e-5)    # :69
    exp_v = log_prob * td.detach()                                        # :70
    entropy = -(a_prob * torch.log(a_prob + 1e-5)).sum(dim=1, keepdim=True)  # :71-72
    a_loss = (-(beta * entropy + exp_v)).mean()                           # :73-74
    return a_loss, c_loss


class TFRMSProp:
    """tf.train.RMSPropOptimizer(learning_rate) as TF1 implements it (main.py:300-301):
        ms <- decay * ms + (1 - decay) * g^2         (ms initialised to ONES)
        var <- var - lr * g / sqrt(ms + epsilon)     (epsilon = 1e-10 inside the sqrt, momentum = 0)"""

    def __init__(self, params, lr, decay=0.9, eps=1e-10):
        self.params = list(params)
        self.lr, self.decay, self.eps = float(lr), float(decay), float(eps)
        self.ms = [torch.ones_like(p) for p in self.params]

    @torch.no_grad()
    def step(self):
        grads = [p.grad for p in self.params]
        torch._foreach_mul_(self.ms, self.decay)
        torch._foreach_addcmul_(self.ms, grads, grads, value=1.0 - self.decay)
        denom = torch._foreach_add(self.ms, self.eps)
        torch._foreach_sqrt_(denom)
        torch._foreach_addcdiv_(self.params, grads, denom, value=-self.lr)

    def zero_grad(self):
        for p in self.params:
            if p.grad is not None:
                p.grad.zero_()          # keeps FlatParams' gradient views in place

    def state_dict(self):
        return {"ms": [m.clone() for m in self.ms], "lr": self.lr, "decay": self.decay, "eps": self.eps}

    def load_state_dict(self, sd):
        for m, s in zip(self.ms, sd["ms"]):
            m.copy_(s)


def nstep_returns(rewards, bootstrap, gamma=GAMMA):
    """a2c_single_thread.py:176-183: value_estimate = r + gamma * value_estimate, backwards over the rollout.
    rewards [T, N], bootstrap [N] (v(s_T), or 0 where the episode ended) -> targets [T, N]."""
    T = rewards.shape[0]
    out = torch.empty_like(rewards)
    run = bootstrap
    for t in range(T - 1, -1, -1):
        run = rewards[t] + gamma * run
        out[t] = run
    return out


def allreduce_mean_grads(params):
    """One flat all-reduce of every gradient (sum / world size).  With one process per GPU and backend 'nccl' this is
    RCCL over xGMI; a no-op without an initialised process group.  Returns the number of float32 elements reduced."""
    import torch.distributed as dist

    grads = [p.grad for p in params if p.grad is not None]
    n = sum(g.numel() for g in grads)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return n
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(dist.get_world_size())
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()
    return n


def sample_actions(prob, generator=None, uniforms=None):
    """One action per row of ``prob`` [N, n_action], the way the reference draws it (main.py:167-168):
    np.random.choice(range(n), p=p) is cdf = p.cumsum(); cdf /= cdf[-1]; searchsorted(cdf, uniform, side='right').
    Same three steps on the device (one uniform per env from ``generator``); a third of the time of torch.multinomial at
    [8192, 625] (tools/profile_a2c.py).  An action of probability 0 is never drawn; the result is always < n_action."""
    cdf = prob.cumsum(dim=1)
    if uniforms is None:
        uniforms = torch.rand((prob.shape[0], 1), device=prob.device, dtype=prob.dtype, generator=generator)
    u = uniforms.reshape(-1, 1).to(prob.dtype) * cdf[:, -1:]
    return torch.searchsorted(cdf, u, right=True).squeeze(1).clamp_(max=prob.shape[1] - 1)


_TUNABLE_DONE = False
_TUNABLE_DIR = None


def freeze_gemm_tuning():
    """Stop TunableOp from TUNING further shapes (the picks already made or loaded stay in use): called by A2CRunner.train_rollout
    after its first rollout + update, so that no later GEMM of the host application -- or of a graph capture -- triggers a timing
    run, and two processes that warmed up on the same shapes keep the same picks.  Returns True when tuning was on and is now off."""
    if _TUNABLE_DONE:
        import torch.cuda.tunable as tun

        tun.tuning_enable(False)
        return True
    return False


def gemm_tuning_is_active():
    """True while TunableOp may still start timing runs for new shapes (enable_gemm_tuning() called, freeze_gemm_tuning() not yet)."""
    if not _TUNABLE_DONE:
        return False
    import torch.cuda.tunable as tun

    return bool(tun.tuning_is_enabled())


def _remove_tunable_dir():
    import shutil

    if _TUNABLE_DIR:
        try:                                  # TunableOp rewrites its results file when the process ends, after this handler:
            import os                         # point it away from the directory that is about to disappear

            import torch.cuda.tunable as tun

            tun.set_filename(os.devnull)
        except Exception:
            pass
        shutil.rmtree(_TUNABLE_DIR, ignore_errors=True)


def enable_gemm_tuning(max_ms=400):
    """Let PyTorch's TunableOp pick the fastest rocBLAS / hipBLASLt solution per GEMM shape.  The defaults run the learner's
    K = 200 shapes and its 409 600-deep weight-gradient reductions at 34-73 TFLOP/s; the tuned picks reach 45-108
    (profiles/r02f_gemm_tunableop.txt: the update's seven large GEMMs 7.3 -> 5.2 ms).  Picks for the BASELINE config 3 shapes ship
    in data/tunableop_gfx950.csv (valid for this image's ROCm / hipBLASLt build: TunableOp checks the versions recorded in the file
    and ignores it otherwise); any other shape is tuned at first use for at most ``max_ms``, outside graph capture (the rollout graph
    is captured after an eager warm-up pass).  Once per process; a no-op without a GPU or without torch.cuda.tunable.
    OPT-IN (A2CRunner(tune_gemms=True); bench.py and tools/train_a2c.py ask for it): it switches TunableOp on for the whole process.
    Bit-identical checkpoint resume holds for the shipped shapes (their picks come from the file); a shape tuned at first use may get
    another pick in another process."""
    global _TUNABLE_DONE, _TUNABLE_DIR
    if _TUNABLE_DONE or not torch.cuda.is_available():
        return _TUNABLE_DONE
    try:
        import atexit
        import os
        import shutil
        import tempfile

        import torch.cuda.tunable as tun

        src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "tunableop_gfx950.csv")
        _TUNABLE_DIR = tempfile.mkdtemp(prefix="uavagent_tunable_")
        atexit.register(_remove_tunable_dir)
        dst = os.path.join(_TUNABLE_DIR, "tunableop_results.csv")   # TunableOp rewrites its file at exit
        if os.path.isfile(src):
            shutil.copyfile(src, dst)
        tun.set_filename(dst)
        tun.set_max_tuning_duration(int(max_ms))
        tun.enable(True)
        tun.tuning_enable(True)
        _TUNABLE_DONE = True
    except Exception:                                        # an optional speed-up: never a reason to fail
        _TUNABLE_DONE = False
    return _TUNABLE_DONE


PARAM_ORDER = ("a_w1", "a_b1", "a_w2", "a_b2", "a_w3", "a_b3", "c_w1", "c_b1", "c_w2", "c_b2", "c_w3", "c_b3")
N_ACTOR_PARAMS = 6


class FlatParams:
    """All parameters of an ACNet as views of ONE flat float32 buffer, their gradients as views of a second one and the RMSProp
    mean squares in a third (every view starts on a 256-byte boundary; the padding stays zero).  One buffer = one all-reduce
    with no gather / scatter copies, one fused optimiser launch per trunk, one memset to clear the gradients.  Autograd
    accumulates into the existing .grad views in place, so the reference (autograd) path fills the same buffer."""

    ALIGN = 64   # floats

    def __init__(self, net):
        params = [getattr(net, k) for k in PARAM_ORDER]
        dev = params[0].device
        offs, off = [], 0
        for p in params:
            offs.append(off)
            off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.n_flat = off
        self.n_real = sum(p.numel() for p in params)
        self.actor_end = offs[N_ACTOR_PARAMS]               # [0, actor_end) actor trunk, [actor_end, n_flat) critic trunk
        self.w = torch.zeros(off, dtype=torch.float32, device=dev)
        self.g = torch.zeros(off, dtype=torch.float32, device=dev)
        self.ms = torch.ones(off, dtype=torch.float32, device=dev)    # TF1: accumulator initialised to ones (main.py:300-301)
        self.gv = {}
        with torch.no_grad():
            for k, p, o in zip(PARAM_ORDER, params, offs):
                view = self.w[o:o + p.numel()].view_as(p)
                view.copy_(p)
                p.data = view
                p.grad = self.g[o:o + p.numel()].view_as(p)
                self.gv[k] = p.grad

    def zero_grad(self):
        self.g.zero_()


class A2CRunner:
    """Synchronous A2C over a BatchedMobiEnv: every env instance plays the role of one of the reference's workers
    (a2c_single_thread.py:113-118), all stepped by one kernel launch per time step.

    On a GPU the rollout loop is ONE hipGraph (per step three kernels: first layer fed from the env's compact observation + the actor's
    head (layer 2, policy head, action draw) + env step) and the update is a hand-derived backward pass on libuavagent's kernels
    (float32 MFMA GEMMs with fused epilogues, loss gradient, table gradient, RMSProp).  Every fusion has a switch
    (``hip_gemms``, ``fused_head``, ``fused_obs``, ``overlap_dw``, ``overlap_allreduce``) whose other setting is the form it replaced; the
    tests compare the two bit for bit where the arithmetic is the same and to tolerance where it is not.  ``update_reference`` is the same
    update through autograd."""

    def __init__(self, env, net=None, rollout=50, gamma=GAMMA, beta=ENTROPY_BETA, lr_a=LR_A, lr_c=LR_C, seed=6,
                 update_chunk=65536, first_state="obs", collect_launch="graph", fused_update=True, tune_gemms=False, hip_gemms=True,
                 overlap_allreduce=True, fused_head=True, overlap_dw=False, fused_obs=True, early_sort=True, pipeline_halves=True, force_exchange=False,
                 persistent_rollout="auto"):
        self.env = env
        self.dev = env.device
        self.gemm_tuning = enable_gemm_tuning() if (tune_gemms and self.dev.type == "cuda") else False
        self.G, self.B = env.grid_n, env.nBS
        self.net = (net if net is not None else ACNet(env.observation_space_dim, env.action_space_dim, seed=seed)).to(self.dev)
        self.flat = FlatParams(self.net)
        self.lr_a, self.lr_c = float(lr_a), float(lr_c)
        self.opt_a = TFRMSProp(self.net.actor_params(), lr_a)       # reference path only; shares FlatParams' accumulators
        self.opt_c = TFRMSProp(self.net.critic_params(), lr_c)
        for opt, keys in ((self.opt_a, PARAM_ORDER[:N_ACTOR_PARAMS]), (self.opt_c, PARAM_ORDER[N_ACTOR_PARAMS:])):
            opt.ms = [self._ms_view(k) for k in keys]
        self.T, self.gamma, self.beta = int(rollout), float(gamma), float(beta)
        self.update_chunk = int(update_chunk)
        self.gen = torch.Generator(device=self.dev).manual_seed(int(seed) + 1000 * int(env.env_id_base + 1))
        if first_state not in ("obs", "zeros"):
            raise ValueError("first_state must be 'obs' or 'zeros'")
        if collect_launch not in ("graph", "eager"):
            raise ValueError("collect_launch must be 'graph' or 'eager'")
        self.collect_launch = collect_launch
        self.fused_update = bool(fused_update)
        # hip_gemms: the update's dense layers through libuavagent's float32 MFMA kernels (csrc/agent_gemm.hip: relu6 masks and bias
        # gradients fused) instead of torch.mm + separate relu6-backward passes.  They are written for the reference's layer widths.
        self.hip_gemms = bool(hip_gemms) and HIDDEN == 200 and self.net.n_action <= 640
        # more than one rank: the critic trunk's gradient (40 MB, first half of the exchange) is all-reduced on a side stream while the
        # actor trunk's backward pass still runs (update_fused, hip_gemms path)
        self.overlap_allreduce = bool(overlap_allreduce)
        # force_exchange (tests): run the gradient exchange -- the collective calls, the side stream, the bucket order -- even when the
        # process group has ONE rank, so that a one-GPU box executes the call sequence an 8-GPU run starts with (backend nccl = RCCL)
        self.force_exchange = bool(force_exchange)
        # fused_head: layer 2, the policy head and the action draw of a rollout step as ONE kernel (uavagent_actor_head_f32)
        self.fused_head = bool(fused_head) and self.hip_gemms and 576 < self.net.n_action <= 640
        # overlap_dw (one rank, hip_gemms): the three dW GEMMs (MFMA bound, 1.6 ms at config 3) run on a side stream beside the
        # first-layer table gradient (sort + indexed row sums: memory bound, 2.4 ms) instead of between the dX GEMMs.  Off by default:
        # side by side each runs that much slower (rocprofv3: the gather 2.2 -> 3.2 ms, a dW GEMM 0.33 -> up to 2.1 ms); the update takes
        # the same 6.9 ms either way (tools/ab_update.py)
        self.overlap_dw = bool(overlap_dw)
        # early_sort (hip_gemms): the table gradient's sort of the (row, sample) pairs starts on a side stream before the update's forward pass
        self.early_sort = bool(early_sort)
        # fused_obs: steps 1 .. T-1 of a rollout build their index list inside the first layer's kernel (uavagent_first_layer_from_obs_f32)
        # instead of a separate obs_indices launch after every env step
        self.fused_obs = bool(fused_obs) and env.nBS + env.nUE <= 64
        self._side = None
        # pipeline_halves (GPU, fused_head + fused_obs): the workers of a2c_single_thread.py:113-118 are independent of each other for a
        # whole rollout, so the batch is cut in two halves that ping-pong on two streams inside the rollout (and inside its captured
        # graph): while the actor's head of one half runs (MFMA / LDS-DMA bound; 16-row workgroups, so that half a batch still gives
        # every CU one), the other half steps its envs (uavenv_step_range) and gathers its first layer (fabric bound, no MFMA).  The
        # two heads never overlap each other (events), which is what keeps the halves out of lockstep.  Same arithmetic per row with the
        # same uniforms: bit-identical to the unsplit rollout.  Measured at 8192 envs x 50 steps, interleaved in one process
        # (profiles/r04i_ab_collect_pipeline_forms.json): 4.40-4.46 ms per rollout against 5.04-5.17 unsplit; three or four parts, or
        # no order between the heads, are slower (every cross-stream edge of the graph costs ~9 us, every node of a two-stream graph ~5).
        # True = when each half still fills the chip (>= 4096 envs), "force" = whenever the batch can be cut (tests).
        self._halves = None
        self._pipe_streams = []
        import os as _os

        # UAVAGENT_PIPE_MODE (A/B runs, tools/ab_collect.py): "alternate" = the parts' heads run one after the other (events), everything
        # else free; "free" = no cross-stream order at all; "stagger" = free, but part p starts p gathers late.  UAVAGENT_PIPE_PARTS: 2..8.
        self.pipeline_mode = _os.environ.get("UAVAGENT_PIPE_MODE", "alternate")
        n_parts = int(_os.environ.get("UAVAGENT_PIPE_PARTS", "2"))
        if pipeline_halves and self.dev.type == "cuda" and self.fused_head and self.fused_obs and 2 <= n_parts <= 8:
            unit = 16                                                 # the head's 16-row tiles stay whole (a part of a batch runs on 16-row
            per = (env.n_envs // n_parts + unit - 1) // unit * unit   #  workgroups: 8192 envs = 2 x 256 tiles = one workgroup per CU and half)
            cuts = [min(i * per, env.n_envs) for i in range(n_parts)] + [env.n_envs]
            if all(cuts[i] < cuts[i + 1] for i in range(n_parts)) and (pipeline_halves == "force" or env.n_envs >= 4096):
                self._halves = tuple((cuts[i], cuts[i + 1]) for i in range(n_parts))
        # persistent_rollout (GPU, fused_head + fused_obs, the reference's 4 UAVs and <= 64 UEs): the whole rollout as TWO persistent kernel
        # launches on two streams -- uavagent_actor_head_gated_f32 (the policy) and uavenv_rollout_gated (env step + next observation's first
        # layer) -- that hand 16-env blocks to each other through step counters in device memory (include/uavenv.h has the protocol): no kernel
        # boundary, graph node or cross-stream edge per step any more; each CU hosts one workgroup of either kernel and ping-pongs between the
        # two blocks of its pair.  Same arithmetic per env and step: bit-identical to the other forms.  Every device-side wait is bounded; a
        # timeout makes collect() restore the state the rollout started from and collect it again with the per-step launches.  "auto" = from 4096 envs on; True = whenever the shapes allow (tests); the first
        # collect() proves on a CLONE of the env state that the two kernels do run side by side, and falls back to pipeline_halves if not.
        self._persistent = False
        if _os.environ.get("UAVAGENT_PERSISTENT") in ("0", "1"):          # A/B runs (tools, bench): overrides the argument
            persistent_rollout = _os.environ["UAVAGENT_PERSISTENT"] == "1"
        self._persist_stream = None
        self._persistent_proven = False
        self._persist_same_stream = _os.environ.get("UAVAGENT_PERSIST_SAME_STREAM", "0") == "1"
        self._gate_spin_us = int(_os.environ.get("UAVAGENT_GATE_SPIN_US", "0"))     # 0 = the library's 2 s
        if (persistent_rollout and self.dev.type == "cuda" and self.fused_head and self.fused_obs and env.nBS == 4 and env.nUE <= 64
                and env.n_envs % 4 == 0 and "cur_sinr_f64" not in env.out and (persistent_rollout is True or env.n_envs >= 4096)):
            self._persistent = True
        N, T, K = env.n_envs, self.T, env.nBS + env.nUE
        # rollout buffers (persistent: the captured graph holds their addresses).  idx_buf[t] = observation BEFORE step t,
        # idx_buf[T] = the state the rollout ended in (bootstrap value; copied to slot 0 when the next rollout starts).
        self.idx_buf = torch.empty((T + 1, N, K), dtype=torch.int64, device=self.dev)
        self.act_buf = torch.empty((T, N), dtype=torch.int64, device=self.dev)
        self.rew_buf = torch.empty((T, N), dtype=torch.float32, device=self.dev)
        self.u_buf = torch.empty((T, N), dtype=torch.float32, device=self.dev)
        # Forward activations of the rollout, kept for the update: between collect() and update() the weights do not change, so
        # relu6(first layer) of both trunks, the actor's second layer and its logits for sample (t, n) ARE the update's forward
        # pass for that sample -- the update only adds the critic's second layer and value head.  (2 GB at 8192 envs x 50 steps.)
        self._fwd = None
        if self.dev.type == "cuda":
            H, NA = HIDDEN, self.net.n_action
            f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=self.dev)
            # logits live in rows of LDL = n_action rounded up to 16 floats (625 -> 640) with a ZERO tail: rows start 16-byte aligned,
            # so the update's GEMMs stage them with float4 loads and may read the tail (it multiplies zero weights)
            self._ldl = (NA + 15) // 16 * 16
            self._logits_pad = torch.zeros((T, N, self._ldl), dtype=torch.float32, device=self.dev)
            self._fwd = {"h1a": f(T, N, H), "h1c": f(T, N, H), "h2a": f(T, N, H), "logits": self._logits_pad[:, :, :NA]}
            # the actor's dense layers handed to uavagent_gemm_rows_f32 in its fast form: W^T (k-contiguous rows), the policy head
            # padded to LDL rows / bias entries of zeros (so the tail of every logits row comes out zero); refreshed from the
            # parameters at the start of every collect()
            self._wt = {"a_w2t": f(H, H), "c_w2t": f(H, H), "a_w3t": torch.zeros((self._ldl, H), dtype=torch.float32, device=self.dev),
                        "a_b3p": torch.zeros(self._ldl, dtype=torch.float32, device=self.dev)} if self.hip_gemms else None
        self._fwd_valid = False
        if self._persistent:
            nb = (N + 15) // 16
            self._gate_obs = torch.zeros(nb, dtype=torch.int32, device=self.dev)
            self._gate_act = torch.zeros(nb, dtype=torch.int32, device=self.dev)
            self._gate_claim = torch.zeros(2, dtype=torch.int32, device=self.dev)
            self._persist_state = torch.empty(env._lay.total_bytes, dtype=torch.uint8, device=self.dev)
            from . import _agent_capi as _A

            _A.gate_prepare()
        # first_state: "obs" = the observation the constructor's channel update produced; "zeros" = what the reference's first
        # work() call sees, the all-zero env.state of a never-reset env (a2c_single_thread.py:143,155): no non-zero cell, i.e.
        # every index is -1 = "no row" and the first layer returns its bias.
        self.idx_buf[T] = obs_to_indices(env.observation(), self.G, self.B)
        if first_state == "zeros":
            self.idx_buf[T].fill_(-1)
        self.ep_r = torch.zeros(N, device=self.dev)
        self.running_r = None                                          # GLOBAL_RUNNING_R EMA, :169-172
        self.last_episode_return = None
        self.stats = {}
        self._graph = None
        self._upd = None
        self._tuning_frozen = False

    def _ms_view(self, key):
        p = getattr(self.net, key)
        off = (p.data_ptr() - self.flat.w.data_ptr()) // 4
        return self.flat.ms[off:off + p.numel()].view_as(p)

    @property
    def idx(self):
        """Indices of the current observation [N, B + U] (what the next action will be chosen from)."""
        return self.idx_buf[self.T]

    # ---- rollout ---------------------------------------------------------------------------------------------------
    def _indices_into(self, out):
        if self.dev.type == "cuda":
            from . import _agent_capi as A

            A.obs_indices(self.env.observation(), self.G, self.B, out=out)
        else:
            out.copy_(obs_to_indices(self.env.observation(), self.G, self.B))

    def _rollout_steps(self):
        """The T-step loop: choose_action (main.py:165-169) -> env.step -> next observation.  No host synchronisation, no
        allocation visible to the caller: capturable."""
        if self._persistent:
            return self._rollout_steps_persistent()
        if self._halves is not None:
            return self._rollout_steps_pipelined()
        env, T, net = self.env, self.T, self.net
        self.idx_buf[0].copy_(self.idx_buf[T])
        cuda = self.dev.type == "cuda"
        if cuda:
            from . import _agent_capi as A
        fw = self._fwd
        wt = self._wt if cuda else None
        fused_obs = cuda and self.fused_obs
        for t in range(T):
            if cuda and fused_obs and t > 0:
                # the index list of step t is built from the observation step t - 1 left behind, inside the gather (and stored)
                A.first_layer_from_obs(env.observation(), self.G, net.a_w1, net.a_b1, net.c_w1, net.c_b1, fw["h1a"][t], fw["h1c"][t],
                                       idx_out=self.idx_buf[t])
            elif cuda:
                # both trunks' first layers in one gather (tried twice, round 2 and round 3: the critic's half on a second stream beside
                # the actor's GEMMs is SLOWER, 6.3-6.4 against 5.5 ms per rollout: two 800-byte gathers cost more than one of 1600)
                A.sparse_rows_sum(self.idx_buf[t], net.a_w1, net.a_b1, net.c_w1, net.c_b1, relu6=True, out_a=fw["h1a"][t],
                                  out_c=fw["h1c"][t])
            if cuda:
                if wt is not None and self.fused_head:      # layer 2 + policy head + action draw: one launch, 32 rows per workgroup
                    A.actor_head(fw["h1a"][t], wt["a_w2t"], net.a_b2, wt["a_w3t"], wt["a_b3p"], self.u_buf[t], net.n_action,
                                 fw["h2a"][t], self._logits_pad[t], self.act_buf[t])
                elif wt is not None:     # float32 MFMA kernels, bias / relu6 fused, 64-row workgroups (8192 rows fill the chip)
                    A.gemm_rows(fw["h1a"][t], wt["a_w2t"], fw["h2a"][t], w_transposed=True, bias=net.a_b2, relu6=True)
                    A.gemm_rows(fw["h2a"][t], wt["a_w3t"], self._logits_pad[t], w_transposed=True, bias=wt["a_b3p"])
                    A.sample_actions(fw["logits"][t], self.u_buf[t], out=self.act_buf[t])
                else:
                    torch.addmm(net.a_b2, fw["h1a"][t], net.a_w2, out=fw["h2a"][t]).clamp_(0.0, 6.0)
                    torch.addmm(net.a_b3, fw["h2a"][t], net.a_w3, out=fw["logits"][t])
                    A.sample_actions(fw["logits"][t], self.u_buf[t], out=self.act_buf[t])
            else:
                prob = net.actor_only(self.idx_buf[t])
                self.act_buf[t] = sample_actions(prob, uniforms=self.u_buf[t])
            env.step(self.act_buf[t], reward_out=self.rew_buf[t])
            if not fused_obs or t == T - 1:
                self._indices_into(self.idx_buf[t + 1])

    def _rollout_steps_persistent(self):
        """The T-step loop as two persistent launches (see persistent_rollout in __init__): the first layer of the state the rollout starts
        from, the gates, then the policy kernel on the calling stream and the env kernel on a side stream.  Capturable."""
        from . import _agent_capi as A

        env, T, net, fw, wt = self.env, self.T, self.net, self._fwd, self._wt
        env.copy_state_to(self._persist_state)       # one 11 MB device copy per rollout: what collect() falls back on if a kernel gives up
        self.idx_buf[0].copy_(self.idx_buf[T])
        A.sparse_rows_sum(self.idx_buf[0], net.a_w1, net.a_b1, net.c_w1, net.c_b1, relu6=True, out_a=fw["h1a"][0], out_c=fw["h1c"][0])
        self._gate_obs.fill_(1)
        self._gate_act.zero_()
        self._gate_claim.zero_()
        main = torch.cuda.current_stream(self.dev)
        if self._persist_stream is None:
            # a stream of its own, HIGH priority: the runtime keeps separate hardware queues per priority level, so this stream cannot land on
            # the queue of the (normal-priority) stream the policy kernel is launched on -- two kernels that wait for each other must never sit
            # behind one another in ONE queue (that ends in the bounded waits' error code, and in the fallback of _prove_persistent)
            self._persist_stream = torch.cuda.Stream(device=self.dev, priority=-1)
        side = self._persist_stream
        if self._persist_same_stream:                # test hook (UAVAGENT_PERSIST_SAME_STREAM=1): both kernels in ONE queue, i.e. never side by side
            side = main
        fork = torch.cuda.Event()
        fork.record(main)
        side.wait_event(fork)
        A.actor_head_gated(fw["h1a"], wt["a_w2t"], net.a_b2, wt["a_w3t"], wt["a_b3p"], self.u_buf, net.n_action, fw["h2a"], self._logits_pad, self.act_buf,
                           self._gate_obs, self._gate_act, self._gate_claim[1:2], spin_us=self._gate_spin_us)
        with torch.cuda.stream(side):
            env.rollout_gated(self.act_buf, self._gate_act, self._gate_obs, self._gate_claim[0:1], net.a_w1, net.a_b1, fw["h1a"], net.c_w1, net.c_b1, fw["h1c"],
                              idx_out=self.idx_buf, reward_out=self.rew_buf)
        join = torch.cuda.Event()
        join.record(side)
        main.wait_event(join)

    def _prove_persistent(self):
        """Eager launches: one trial rollout on a CLONE of the env state shows whether the two persistent kernels run side by side on the
        streams this runner uses (the stream -> hardware-queue mapping is fixed when a stream is created); if not, the per-step launches take
        over for good.  (The captured form has the same trial in _capture.)"""
        env = self.env
        state = torch.empty(env._lay.total_bytes, dtype=torch.uint8, device=self.dev)
        env.copy_state_to(state)
        keep = {k: v.clone() for k, v in env.out.items()}
        keep_idx = self.idx_buf[self.T].clone()
        refused = None
        try:
            self._rollout_steps()
        except Exception as ex:                        # an entry point refused the shapes before launching anything
            refused = ex
        torch.cuda.synchronize(self.dev)
        failed = refused is not None or self._persistent_failed()
        if failed:
            import warnings
            from . import _agent_capi as A

            warnings.warn("A2CRunner: the persistent rollout kernels did not run side by side (eager trial: %s); using the per-step launches "
                          "instead" % ("refused: %s" % refused if refused is not None else "a gate wait timed out"))
            self._persistent = False
            A.device_error_clear()
        env.copy_state_from(state)                     # (also clears the env handle's device-error word)
        for k, v in keep.items():
            env.out[k].copy_(v)
        self.idx_buf[self.T].copy_(keep_idx)
        self._persistent_proven = True

    def _persistent_failed(self):
        """True when a gated launch gave up on the device (both libraries keep a sticky word in host-mapped memory)."""
        from . import _agent_capi as A

        return A.device_error() != 0 or self.env.device_error() != 0

    def _rollout_steps_pipelined(self):
        """The T-step loop with the batch cut in parts, one stream each (see pipeline_halves in __init__).  Per part and step: first
        layer from the observation the part's previous step left (gather), actor head, env step of the part's envs
        (uavenv_step_range).  Cross-stream order in mode "alternate": head(part p, step t) after head(part p - 1, t), head(part 0, t + 1)
        after head(last part, t).  Capturable: the extra streams fork from and join the calling stream through events."""
        from . import _agent_capi as A

        env, T, net, fw, wt = self.env, self.T, self.net, self._fwd, self._wt
        self.idx_buf[0].copy_(self.idx_buf[T])
        main = torch.cuda.current_stream(self.dev)
        P = len(self._halves)
        while len(self._pipe_streams) < P - 1:
            # (normal priority.  A high-priority stream would have a hardware queue of its own -- with a dozen streams alive in one process the
            #  side stream can end up in the calling stream's queue, where the halves run one after the other: 8.8-10.3 instead of 4.4 ms in a
            #  five-runner A/B process, profiles/r04fd_ab_collect_five_runners_in_one_process.json -- but the captured graph then runs the rollout in
            #  12.9 ms: profiles/r04gu_ab_bench_a2c_pipelined_on_a_high_priority_stream.txt)
            self._pipe_streams.append(torch.cuda.Stream(device=self.dev))
        streams = [main] + self._pipe_streams[:P - 1]
        fork = torch.cuda.Event()
        fork.record(main)
        for st in streams[1:]:
            st.wait_event(fork)
        obs = env.observation()
        alternate, stagger = self.pipeline_mode == "alternate", self.pipeline_mode == "stagger"
        last_head = None                                              # event behind the head issued last (mode "alternate")
        first_gather = [None] * P
        for t in range(T):
            for h, ((lo, hi), st) in enumerate(zip(self._halves, streams)):
                with torch.cuda.stream(st):
                    if t == 0 and h > 0 and stagger:                 # part h starts when part h - 1 has done its first gather, then runs free
                        st.wait_event(first_gather[h - 1])
                    if t == 0:
                        A.sparse_rows_sum(self.idx_buf[0][lo:hi], net.a_w1, net.a_b1, net.c_w1, net.c_b1, relu6=True, out_a=fw["h1a"][0][lo:hi],
                                          out_c=fw["h1c"][0][lo:hi])
                        if stagger:
                            first_gather[h] = torch.cuda.Event()
                            first_gather[h].record(st)
                    else:
                        A.first_layer_from_obs({k: v[lo:hi] for k, v in obs.items()}, self.G, net.a_w1, net.a_b1, net.c_w1, net.c_b1,
                                               fw["h1a"][t][lo:hi], fw["h1c"][t][lo:hi], idx_out=self.idx_buf[t][lo:hi])
                    if alternate and last_head is not None:
                        st.wait_event(last_head)                     # the heads run one after the other; everything else overlaps them
                    A.actor_head(fw["h1a"][t][lo:hi], wt["a_w2t"], net.a_b2, wt["a_w3t"], wt["a_b3p"], self.u_buf[t][lo:hi], net.n_action,
                                 fw["h2a"][t][lo:hi], self._logits_pad[t][lo:hi], self.act_buf[t][lo:hi])
                    if alternate:
                        last_head = torch.cuda.Event()
                        last_head.record(st)
                    env.step_range(self.act_buf[t], lo, hi - lo, reward_out=self.rew_buf[t])
                    if t == T - 1:
                        A.obs_indices({k: v[lo:hi] for k, v in obs.items()}, self.G, self.B, out=self.idx_buf[T][lo:hi])
        for st in streams[1:]:
            join = torch.cuda.Event()
            join.record(st)
            main.wait_event(join)

    @torch.no_grad()
    def collect(self):
        """One rollout: returns (idx [T,N,K], actions [T,N], rewards [T,N], bootstrap [N]) -- views of persistent buffers, valid
        until the next collect()."""
        env, T = self.env, self.T
        self.u_buf.copy_(torch.rand(self.u_buf.shape, device=self.dev, dtype=torch.float32, generator=self.gen))
        self._refresh_transposed()
        if self.collect_launch == "graph" and self.dev.type == "cuda" and self._graph is None:
            try:
                self._capture()
            except Exception as ex:      # a failed capture executes nothing (the warm-up pass has been rolled back): run eagerly
                import warnings

                warnings.warn("A2CRunner: hipGraph capture of the rollout failed (%s: %s); collecting eagerly" % (type(ex).__name__, ex))
                self.collect_launch = "eager"
                self._graph = None
        persistent_ran = self._persistent
        if self.collect_launch == "graph" and self._graph is not None:
            self._graph.replay()
        else:
            if self._persistent and not self._persistent_proven:
                self._prove_persistent()
                persistent_ran = self._persistent
            self._rollout_steps()
        done = env.out["done"].bool()
        any_done = bool(done.any())                                                  # (host sync: the rollout's kernels have finished)
        if persistent_ran and self._persistent_failed():
            # A persistent rollout kernel gave up waiting for its partner in the MIDDLE of a run (they had been proven to run side by side: so
            # something else kept one of them off the chip for longer than the spin budget).  Nothing is lost: the rollout started from the
            # state snapshot _rollout_steps_persistent took, the uniforms are still in u_buf -- restore, switch to the per-step launches for
            # good, and collect this rollout again: same results as if the persistent kernels had finished.
            import warnings
            from . import _agent_capi as _A2

            warnings.warn("A2CRunner.collect: a persistent rollout kernel gave up waiting for its partner (device error words: policy 0x%08x, env "
                          "0x%08x); the rollout is collected again with the per-step launches, which this runner uses from now on"
                          % (_A2.device_error(), env.device_error()))
            _A2.device_error_clear()
            env.copy_state_from(self._persist_state)                                 # (also clears the env handle's device-error word)
            self.idx_buf[T].copy_(self.idx_buf[0])
            self._persistent = False
            self._graph = None                                                       # (the next collect() captures the per-step form)
            self._rollout_steps()
            done = env.out["done"].bool()
            any_done = bool(done.any())
        self._fwd_valid = self._fwd is not None
        self.ep_r += self.rew_buf.sum(dim=0)
        boot = self.net.critic_only(self.idx_buf[T]).squeeze(1)                      # :173-176
        boot = torch.where(done, torch.zeros_like(boot), boot)                       # value_estimate = 0 when done
        if any_done:                                                                 # :167-172 reset_worker
            m = float(self.ep_r[done].mean())
            self.last_episode_return = m                                             # mean return of the episodes that just ended
            self.running_r = m if self.running_r is None else 0.99 * self.running_r + 0.01 * m
            self.ep_r[done] = 0.0
            env.reset(mask=done)
            self._indices_into(self.idx_buf[T])
        return self.idx_buf[:T], self.act_buf, self.rew_buf, boot

    def _refresh_transposed(self):
        """The transposed / padded copies of the actor's dense layers the rollout's GEMM kernel reads (three small copies; the captured
        graph reads the same buffers)."""
        wt = getattr(self, "_wt", None)
        if wt is not None:
            net = self.net
            wt["a_w2t"].copy_(net.a_w2.t())
            wt["c_w2t"].copy_(net.c_w2.t())
            wt["a_w3t"][:net.n_action].copy_(net.a_w3.t())
            wt["a_b3p"][:net.n_action].copy_(net.a_b3)

    def _capture(self):
        """hipGraph of the rollout loop.  A warm-up pass on a side stream first (rocBLAS handles / workspaces), on a CLONE of the
        env state so that capturing changes nothing the caller can observe."""
        env = self.env
        state = torch.empty(env._lay.total_bytes, dtype=torch.uint8, device=self.dev)
        env.copy_state_to(state)
        keep = {k: v.clone() for k, v in env.out.items()}
        keep_idx = self.idx_buf[self.T].clone()
        def restore():
            env.copy_state_from(state)                 # (also clears the env handle's device-error word)
            for k, v in keep.items():
                env.out[k].copy_(v)
            self.idx_buf[self.T].copy_(keep_idx)

        def give_up_persistent(where):
            import warnings
            from . import _agent_capi as A

            warnings.warn("A2CRunner: the persistent rollout kernels did not run side by side (%s: a gate wait timed out); using the "
                          "per-step launches instead" % where)
            self._persistent = False
            A.device_error_clear()
            restore()

        s = torch.cuda.Stream(device=self.dev)
        for attempt in range(2):
            s.wait_stream(torch.cuda.current_stream(self.dev))
            refused = None
            with torch.cuda.stream(s):
                try:
                    self._rollout_steps()
                except Exception as ex:                    # (persistent form: an entry point refused the shapes before launching anything)
                    if not self._persistent:
                        raise
                    refused = ex
            torch.cuda.current_stream(self.dev).wait_stream(s)
            torch.cuda.synchronize(self.dev)
            if self._persistent and (refused is not None or self._persistent_failed()):
                give_up_persistent("eager warm-up pass" + (", refused: %s" % refused if refused is not None else ""))
                continue
            restore()
            # capture_error_mode="thread_local": with torch.distributed initialised, RCCL's watchdog thread polls events while this
            # thread captures; in the default "global" mode that invalidates the capture.
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self._rollout_steps()
            if self._persistent:
                # a graph's parallel branches need not run at the same time; the two persistent kernels must: one trial replay on the clone
                g.replay()
                torch.cuda.synchronize(self.dev)
                if self._persistent_failed():
                    give_up_persistent("trial replay of the captured graph")
                    continue
                restore()
                self._persistent_proven = True
            self._graph = g
            return

    # ---- update ----------------------------------------------------------------------------------------------------
    def update(self, idx_buf, act_buf, rew_buf, boot):
        """One gradient step on all T*N samples (a2c_single_thread.py:120-133)."""
        if self.fused_update and self.dev.type == "cuda":
            return self.update_fused(idx_buf, act_buf, rew_buf, boot)
        return self.update_reference(idx_buf, act_buf, rew_buf, boot)

    def _allreduce(self):
        """Mean of the flat gradient over ranks: ONE all-reduce (RCCL over xGMI with backend nccl) straight on the buffer the
        backward pass wrote; the 1 / world_size is folded into the optimiser step.  -> (elements reduced, g_scale)."""
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not self.force_exchange):
            return self.flat.n_real, 1.0
        if self.flat.g.is_cuda and dist.get_backend() == "gloo":      # rehearsal on a one-GPU box (several ranks share the
            host = self.flat.g.cpu()                                  # card, RCCL refuses that): reduce through the host
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            self.flat.g.copy_(host)
        else:
            dist.all_reduce(self.flat.g, op=dist.ReduceOp.SUM)
        return self.flat.n_real, 1.0 / dist.get_world_size()

    @staticmethod
    def _world():
        import torch.distributed as dist

        return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1

    def _exchanging(self):
        """True when an update ends in a collective: more than one rank, or force_exchange with an initialised process group."""
        import torch.distributed as dist

        if self._world() > 1:
            return True
        return self.force_exchange and dist.is_available() and dist.is_initialized()

    def _allreduce_bucket(self, lo, hi, async_op=False):
        """Sum of flat.g[lo:hi] over ranks, in place.  nccl (RCCL): returns the work handle when async_op; gloo with CUDA tensors (the
        one-GPU rehearsal: several ranks share the card, RCCL refuses that): through the host, synchronously."""
        import torch.distributed as dist

        buf = self.flat.g[lo:hi]
        if buf.is_cuda and dist.get_backend() == "gloo":
            host = buf.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            buf.copy_(host)
            return None
        return dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=async_op)

    def _ensure_update_buffers(self, M, K):
        from . import _agent_capi as A

        if self._upd is not None and self._upd["M"] == M:
            return self._upd
        H, NA, dev = HIDDEN, self.net.n_action, self.dev
        f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        own = self._fwd is not None and M == self.T * self.env.n_envs      # the rollout's own buffers double as the update's
        ldl = (NA + 15) // 16 * 16
        if own:
            fw = {k: self._fwd[k].view(M, H) for k in ("h1a", "h1c", "h2a")}
            logits_pad = self._logits_pad.view(M, ldl)
        else:
            fw = {"h1a": f(M, H), "h1c": f(M, H), "h2a": f(M, H)}
            logits_pad = torch.zeros((M, ldl), dtype=torch.float32, device=dev)
        self._upd = {"M": M, "own": own, "h1a": fw["h1a"], "h1c": fw["h1c"], "h2a": fw["h2a"], "logits": logits_pad[:, :NA],
                     "logits_pad": logits_pad, "h2c": f(M, H), "dh": f(M, H),
                     "gcat": f(M, 2 * H), "v": f(M), "dv": f(M), "target": f(M), "loss": torch.zeros(3, dtype=torch.float64, device=dev),
                     "ws_loss": A.loss_grad_workspace(NA, dev), "ws_relu": A.relu6_bwd_workspace(H, dev),
                     "ws_rows": A.rows_grad_workspace(M, K, 2 * H, self.net.n_state, dev)}
        if self.hip_gemms:
            self._upd.update({"w3p": torch.zeros((H, ldl), dtype=torch.float32, device=dev),       # a_w3 in rows of ldl, zero tail
                              "ws_tn_h": A.gemm_tn_workspace(M, H, dev), "ws_tn_a": A.gemm_tn_workspace(M, NA, dev),
                              "ws_cs": A.gemm_rows_workspace(M, dev)})
            if self._exchanging() and self.overlap_allreduce:    # one table gradient per trunk: each needs its g rows contiguous
                self._upd.update({"g_a": f(M, H), "g_c": f(M, H)})
            elif self.overlap_dw:                                # the critic's dh2 beside the actor's: both outlive the dX chain
                self._upd["dh_c"] = f(M, H)
        return self._upd

    @torch.no_grad()
    def update_fused(self, idx_buf, act_buf, rew_buf, boot):
        """The update with a hand-derived backward pass (same mathematics as update_reference, main.py:64-74,143-156).  Dense layers:
        libuavagent's float32 MFMA GEMMs (hip_gemms, the default: relu6 backward fused into the dX kernels' epilogue, bias gradients
        taken from the dW kernels' ones column / the dX kernels' column sums) or torch GEMMs + separate relu6-backward passes;
        softmax / loss / its gradient, the value head, the table gradient and RMSProp = libuavagent kernels.  Everything writes
        straight into the flat gradient buffer."""
        from . import _agent_capi as A

        net, fl = self.net, self.flat
        T, N, K = idx_buf.shape
        M, H = T * N, HIDDEN
        b = self._ensure_update_buffers(M, K)
        target = A.nstep_returns(rew_buf.contiguous(), boot.contiguous(), self.gamma, out=b["target"].view(T, N)).reshape(M)
        idx, act = idx_buf.reshape(M, K), act_buf.reshape(M)
        gv = fl.gv
        hip = self.hip_gemms
        # The table gradient's sort needs only idx: on the side stream, beside the forward pass and the dX chain (hip path; one sort
        # serves both trunks' sums when they are exchanged separately).
        early_sort = hip and self.early_sort and idx.is_contiguous()
        if early_sort:
            main = torch.cuda.current_stream(self.dev)
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.dev)
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                A.rows_grad_sort(idx, 2 * H, net.n_state, b["ws_rows"])
            sorted_ev = torch.cuda.Event()
            sorted_ev.record(self._side)

        def table_grad(g, dw0, dw1):
            if early_sort:
                torch.cuda.current_stream(self.dev).wait_event(sorted_ev)
                A.rows_grad_sums((M, K), g, H, net.n_state, dw0, dw1, b["ws_rows"])
            else:
                A.rows_grad(idx, g, H, net.n_state, dw0, dw1, b["ws_rows"])
        # forward: the actor's activations and both first layers were computed by the rollout itself, with these very weights
        reuse = b["own"] and self._fwd_valid and idx_buf.data_ptr() == self.idx_buf.data_ptr()
        if not reuse:
            A.sparse_rows_sum(idx, net.a_w1, net.a_b1, net.c_w1, net.c_b1, relu6=True, out_a=b["h1a"], out_c=b["h1c"])
            if hip:
                self._refresh_transposed()
                A.gemm_rows(b["h1a"], self._wt["a_w2t"], b["h2a"], w_transposed=True, bias=net.a_b2, relu6=True)
                A.gemm_rows(b["h2a"], self._wt["a_w3t"], b["logits_pad"], w_transposed=True, bias=self._wt["a_b3p"])
            else:
                torch.addmm(net.a_b2, b["h1a"], net.a_w2, out=b["h2a"]).clamp_(0.0, 6.0)
                torch.addmm(net.a_b3, b["h2a"], net.a_w3, out=b["logits"])
        self._fwd_valid = False                                        # the backward pass below overwrites logits and h2a
        if hip:
            # one arithmetic whether or not the rollout's forward pass is reused: W2^T through the k-contiguous kernels (collect() /
            # the branch above refreshed the transposed copies from these very weights)
            A.gemm_rows(b["h1c"], self._wt["c_w2t"], b["h2c"], w_transposed=True, bias=net.c_b2, relu6=True)
        else:
            torch.addmm(net.c_b2, b["h1c"], net.c_w2, out=b["h2c"]).clamp_(0.0, 6.0)
        A.rowdot(b["h2c"], net.c_w3, net.c_b3, b["v"])
        # loss and its gradient w.r.t. logits / v (logits are overwritten); d a_b3, d c_b3
        A.a2c_loss_grad(b["logits"], b["v"], target, act, self.beta, b["dv"], gv["a_b3"], b["loss"], b["ws_loss"])
        gv["c_b3"].copy_(b["loss"][2:3].to(torch.float32))
        overlap = hip and ("g_a" in b)
        rows_done = False
        ae = fl.actor_end
        ev = lambda: torch.cuda.Event(enable_timing=True)
        e_c0 = e_c1 = None
        work_c = None
        if overlap:
            # ---- more than one rank: critic trunk first, its half of the gradient on the wire while the actor trunk runs ----
            A.relu6_bwd(None, b["h2c"], b["dh"], H, gv["c_b2"], b["ws_relu"], dv=b["dv"], w3=net.c_w3, dw3_out=gv["c_w3"])
            A.gemm_tn(b["h1c"], b["dh"], gv["c_w2"], b["ws_tn_h"])
            A.gemm_rows(b["dh"], net.c_w2, b["g_c"], w_transposed=True, relu6_mask_h=b["h1c"], colsum_out=gv["c_b1"], workspace=b["ws_cs"])
            table_grad(b["g_c"], gv["c_w1"], None)
            main = torch.cuda.current_stream(self.dev)
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.dev)
            ready = torch.cuda.Event()
            ready.record(main)
            e_c0, e_c1 = ev(), ev()
            with torch.cuda.stream(self._side):
                self._side.wait_event(ready)
                e_c0.record(self._side)
                work_c = self._allreduce_bucket(ae, fl.n_flat, async_op=True)
                if work_c is not None:
                    work_c.wait()                                  # (the SIDE stream waits; the main stream goes on with the actor)
                e_c1.record(self._side)
            b["w3p"][:, :net.n_action].copy_(net.a_w3)
            A.gemm_tn(b["h2a"], b["logits"], gv["a_w3"], b["ws_tn_a"])
            A.gemm_rows(b["logits_pad"], b["w3p"], b["dh"], w_transposed=True, relu6_mask_h=b["h2a"])
            A.gemm_tn(b["h1a"], b["dh"], gv["a_w2"], b["ws_tn_h"], dbias_out=gv["a_b2"])
            A.gemm_rows(b["dh"], net.a_w2, b["g_a"], w_transposed=True, relu6_mask_h=b["h1a"], colsum_out=gv["a_b1"], workspace=b["ws_cs"])
            table_grad(b["g_a"], gv["a_w1"], None)
        elif hip and "dh_c" in b:
            # The dX chain first (every product the table gradient waits for), then the three dW GEMMs on a side stream WHILE the main
            # stream sorts the (row, sample) pairs and sums the indexed rows.  Same kernels on the same operands as the branch below:
            # the gradient is bit-identical (tests/test_learner_kernels_gpu.py).
            b["w3p"][:, :net.n_action].copy_(net.a_w3)
            A.gemm_rows(b["logits_pad"], b["w3p"], b["dh"], w_transposed=True, relu6_mask_h=b["h2a"])
            A.gemm_rows(b["dh"], net.a_w2, b["gcat"][:, :H], w_transposed=True, relu6_mask_h=b["h1a"], colsum_out=gv["a_b1"],
                        workspace=b["ws_cs"])
            A.relu6_bwd(None, b["h2c"], b["dh_c"], H, gv["c_b2"], b["ws_relu"], dv=b["dv"], w3=net.c_w3, dw3_out=gv["c_w3"])
            A.gemm_rows(b["dh_c"], net.c_w2, b["gcat"][:, H:], w_transposed=True, relu6_mask_h=b["h1c"], colsum_out=gv["c_b1"],
                        workspace=b["ws_cs"])
            main = torch.cuda.current_stream(self.dev)
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.dev)
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                A.gemm_tn(b["h2a"], b["logits"], gv["a_w3"], b["ws_tn_a"])
                A.gemm_tn(b["h1a"], b["dh"], gv["a_w2"], b["ws_tn_h"], dbias_out=gv["a_b2"])
                A.gemm_tn(b["h1c"], b["dh_c"], gv["c_w2"], b["ws_tn_h"])
            table_grad(b["gcat"], gv["a_w1"], gv["c_w1"])
            main.wait_stream(self._side)
            rows_done = True
        elif hip:
            # actor trunk backwards: dW3; dh2a = relu6'(h2a) * (dlogits @ W3^T); dW2 + db2; dh1a = relu6'(h1a) * (dh2a @ W2^T) + db1
            b["w3p"][:, :net.n_action].copy_(net.a_w3)
            A.gemm_tn(b["h2a"], b["logits"], gv["a_w3"], b["ws_tn_a"])
            A.gemm_rows(b["logits_pad"], b["w3p"], b["dh"], w_transposed=True, relu6_mask_h=b["h2a"])
            A.gemm_tn(b["h1a"], b["dh"], gv["a_w2"], b["ws_tn_h"], dbias_out=gv["a_b2"])
            A.gemm_rows(b["dh"], net.a_w2, b["gcat"][:, :H], w_transposed=True, relu6_mask_h=b["h1a"], colsum_out=gv["a_b1"],
                        workspace=b["ws_cs"])
            # critic trunk backwards (the value head's outer product + relu6' + db2 + dw3 stay one streaming kernel)
            A.relu6_bwd(None, b["h2c"], b["dh"], H, gv["c_b2"], b["ws_relu"], dv=b["dv"], w3=net.c_w3, dw3_out=gv["c_w3"])
            A.gemm_tn(b["h1c"], b["dh"], gv["c_w2"], b["ws_tn_h"])
            A.gemm_rows(b["dh"], net.c_w2, b["gcat"][:, H:], w_transposed=True, relu6_mask_h=b["h1c"], colsum_out=gv["c_b1"],
                        workspace=b["ws_cs"])
        else:
            # actor trunk backwards
            torch.mm(b["h2a"].t(), b["logits"], out=gv["a_w3"])
            torch.mm(b["logits"], net.a_w3.t(), out=b["dh"])
            A.relu6_bwd(b["dh"], b["h2a"], b["dh"], H, gv["a_b2"], b["ws_relu"])
            torch.mm(b["h1a"].t(), b["dh"], out=gv["a_w2"])
            torch.mm(b["dh"], net.a_w2.t(), out=b["h2a"])                 # h2a is free now: reuse it for d h1a
            A.relu6_bwd(b["h2a"], b["h1a"], b["gcat"], 2 * H, gv["a_b1"], b["ws_relu"])
            # critic trunk backwards
            A.relu6_bwd(None, b["h2c"], b["dh"], H, gv["c_b2"], b["ws_relu"], dv=b["dv"], w3=net.c_w3, dw3_out=gv["c_w3"])
            torch.mm(b["h1c"].t(), b["dh"], out=gv["c_w2"])
            torch.mm(b["dh"], net.c_w2.t(), out=b["h2c"])
            A.relu6_bwd(b["h2c"], b["h1c"], b["gcat"][:, H:], 2 * H, gv["c_b1"], b["ws_relu"])
        # first-layer tables: both in one sorted pass (one rank, or no overlap)
        if not overlap and not rows_done:
            table_grad(b["gcat"], gv["a_w1"], gv["c_w1"])
        # synchronise and step
        ev0, ev1 = ev(), ev()
        ev0.record()
        if overlap:
            self._allreduce_bucket(0, ae)                              # the actor's half, behind its backward pass
            torch.cuda.current_stream(self.dev).wait_stream(self._side)     # ... and the critic's half has landed
            n_red, g_scale = fl.n_real, 1.0 / self._world()
        else:
            n_red, g_scale = self._allreduce()
        ev1.record()
        A.rmsprop_tf1(fl.w[:ae], fl.ms[:ae], fl.g[:ae], self.lr_a, g_scale=g_scale)
        A.rmsprop_tf1(fl.w[ae:], fl.ms[ae:], fl.g[ae:], self.lr_c, g_scale=g_scale)
        loss = b["loss"].cpu()                                        # (synchronises)
        self.stats = {"a_loss": float(loss[0]), "c_loss": float(loss[1]), "mean_reward": float(rew_buf.mean()),
                      "grad_elems": n_red, "running_r": self.running_r, "allreduce_ms": ev0.elapsed_time(ev1),
                      "forward_reused": bool(reuse), "hip_gemms": bool(hip), "dw_on_side_stream": bool(rows_done),
                      # two buckets (critic trunk, then actor trunk): the first one's time is hidden behind the actor's backward pass
                      "allreduce_overlapped_ms": e_c0.elapsed_time(e_c1) if overlap else None,
                      "allreduce_buckets": [4 * (fl.n_flat - ae), 4 * ae] if overlap else None}
        return self.stats

    def update_reference(self, idx_buf, act_buf, rew_buf, boot):
        """The same update through autograd (the form round 1 shipped), chunked to bound activation memory; gradients
        accumulate into the flat buffer, TFRMSProp steps on views of the same flat accumulators."""
        T, N, K = idx_buf.shape
        target = nstep_returns(rew_buf, boot, self.gamma).reshape(T * N, 1)
        idx, act = idx_buf.reshape(T * N, K), act_buf.reshape(T * N)
        M = T * N
        self.flat.zero_grad()
        a_tot = c_tot = 0.0
        for s in range(0, M, self.update_chunk):
            e = min(M, s + self.update_chunk)
            a_prob, v = self.net(idx[s:e])
            a_loss, c_loss = a2c_losses(a_prob, v, act[s:e], target[s:e], self.beta)
            w = (e - s) / M                                   # mean over the whole batch = weighted mean of chunks
            ((a_loss + c_loss) * w).backward()                # disjoint parameter sets: same grads as two backward()s
            a_tot += float(a_loss.detach()) * w
            c_tot += float(c_loss.detach()) * w
        n_red, g_scale = self._allreduce()
        if g_scale != 1.0:
            self.flat.g.mul_(g_scale)
        self.opt_a.step()
        self.opt_c.step()
        self.stats = {"a_loss": a_tot, "c_loss": c_tot, "mean_reward": float(rew_buf.mean()), "grad_elems": n_red,
                      "running_r": self.running_r, "allreduce_ms": None}
        return self.stats

    def train_rollout(self):
        stats = self.update(*self.collect())
        if self.gemm_tuning and not self._tuning_frozen:
            # every GEMM shape of a rollout + update has now run once eagerly: no timing run may start later (inside a training loop, a
            # graph capture of the host application, or with another pick in another process)
            freeze_gemm_tuning()
            self._tuning_frozen = True
        return stats

    # ---- checkpoint / resume (the reference saves the actor only and cannot resume) -------------------------------
    def state_dict(self):
        """Everything a bit-identical continuation needs: both trunks and their RMSProp accumulators (the flat buffers), the env batch
        (state blob + last outputs), the current observation indices, the episode bookkeeping and the action-sampling generator."""
        env = self.env
        blob = torch.empty(env._lay.total_bytes, dtype=torch.uint8, device=self.dev)
        env.copy_state_to(blob)
        return {"format": 1, "n_envs": env.n_envs, "n_bs": env.nBS, "n_ue": env.nUE, "grid_n": env.grid_n, "rollout": self.T,
                "w": self.flat.w.detach().cpu(), "ms": self.flat.ms.detach().cpu(), "env_state": blob.cpu(), "env_out": env._arena.cpu(),
                "idx": self.idx_buf[self.T].cpu(), "ep_r": self.ep_r.cpu(), "running_r": self.running_r,
                "last_episode_return": self.last_episode_return, "gen_state": self.gen.get_state().cpu()}

    def load_state_dict(self, sd):
        env = self.env
        shape = (sd["n_envs"], sd["n_bs"], sd["n_ue"], sd["grid_n"], sd["rollout"])
        if sd.get("format") != 1 or shape != (env.n_envs, env.nBS, env.nUE, env.grid_n, self.T):
            raise ValueError("checkpoint was written for another configuration: %r" % (shape,))
        with torch.no_grad():
            self.flat.w.copy_(sd["w"])
            self.flat.ms.copy_(sd["ms"])
            env.copy_state_from(sd["env_state"].to(self.dev))
            env._arena.copy_(sd["env_out"])
            self.idx_buf[self.T].copy_(sd["idx"])
            self.ep_r.copy_(sd["ep_r"])
        self.running_r, self.last_episode_return = sd["running_r"], sd["last_episode_return"]
        self.gen.set_state(sd["gen_state"].cpu())
        self._fwd_valid = False
        torch.cuda.synchronize(self.dev) if self.dev.type == "cuda" else None


ACTOR_KEYS = ("a_w1", "a_b1", "a_w2", "a_b2", "a_w3", "a_b3")   # TF order: la/kernel, la/bias, la2/kernel, la2/bias, ap/kernel, ap/bias


def save_actor_npz(net, path):
    """Actor parameters, the content of the reference's Global_A_PARA.npz (main.py:265-269: np.savez(path,
    SESS.run(a_params))).  The reference stores the ragged list as ONE pickled object array ('arr_0'); here the same
    six arrays are stored under names, so loading never needs allow_pickle."""
    import numpy as np

    np.savez(path, **{k: getattr(net, k).detach().cpu().numpy() for k in ACTOR_KEYS})


def load_actor_npz(net, path):
    """Inverse of save_actor_npz (main_test.py:11-26 assigns the six arrays to the actor variables in order)."""
    import numpy as np

    with np.load(path, allow_pickle=False) as z:
        missing = [k for k in ACTOR_KEYS if k not in z.files]
        if missing:
            raise ValueError("not an actor checkpoint written by save_actor_npz (missing %s); the reference's own "
                             "Global_A_PARA.npz is a pickled object array and is not loaded" % missing)
        with torch.no_grad():
            for k in ACTOR_KEYS:
                p = getattr(net, k)
                a = torch.as_tensor(z[k])
                if tuple(a.shape) != tuple(p.shape):
                    raise ValueError("%s: checkpoint shape %s != network shape %s" % (k, tuple(a.shape), tuple(p.shape)))
                p.copy_(a.to(p.device, p.dtype))
    return net


def grad_allreduce_bytes(net):
    """Bytes of real gradient per all-reduce (the flat buffer adds < 1 KB of alignment padding)."""
    return 4 * sum(p.numel() for p in net.parameters())


def expected_param_count(n_state, n_action, hidden=HIDDEN):
    """20 206 626 parameters at G = 100, 4 UAVs (80.8 MB float32)."""
    actor = n_state * hidden + hidden + hidden * hidden + hidden + hidden * n_action + n_action
    critic = n_state * hidden + hidden + hidden * hidden + hidden + hidden + 1
    return actor + critic
