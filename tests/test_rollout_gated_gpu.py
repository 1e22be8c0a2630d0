"""uavenv_rollout_gated (include/uavenv.h): T steps + the encoded observation after each, in one persistent launch gated on per-block step
counters.  Here the ENV side alone: the action tape is filled and every action gate opened before the launch, so the kernel runs its T steps
without a partner -- and must leave exactly what T calls of uavenv_step + uavagent_first_layer_from_obs_f32 leave: rewards, outputs, state
blob, index lists, encoded rows, all BIT-IDENTICAL, and the observation gates at T.  With a closed gate it must give up within its spin
budget and poison the handle (UAVENV_E_DEVICE), never hang.  Then the policy side alone (uavagent_actor_head_gated_f32), and the pair on two streams with closed gates.
Reference: mobile_env.py:150-194 (step), :169-170 (observation planes), main.py:147,153 (first dense layer) x T, a2c_single_thread.py:113-133."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _torch():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch


def _env(n, n_ue, grid_n=100, **kw):
    from drl_uav_cellularnet_amd import BatchedMobiEnv

    groups = [n_ue // 4] * 3 + [n_ue - 3 * (n_ue // 4)]
    return BatchedMobiEnv(n, nBS=4, nUE=n_ue, grid_n=grid_n, groups=groups, seed=kw.pop("seed", 31), **kw)


def _tables(torch, env, hid, seed):
    g = torch.Generator().manual_seed(seed)
    rows = (env.nBS + 1) * env.grid_n * env.grid_n
    mk = lambda *s: (torch.rand(*s, generator=g) - 0.3).to(env.device)
    return mk(rows, hid), mk(hid), mk(rows, hid), mk(hid)


def _buffers(torch, env, T, hid):
    N, K = env.n_envs, env.nBS + env.nUE
    dev = env.device
    nb = (N + env.GATE_ROWS - 1) // env.GATE_ROWS
    return dict(out_a=torch.full((T, N, hid), float("nan"), device=dev), out_c=torch.full((T, N, hid), float("nan"), device=dev),
                idx=torch.full((T + 1, N, K), -7, dtype=torch.int64, device=dev), rew=torch.full((T, N), float("nan"), device=dev),
                gate_act=torch.full((nb,), T, dtype=torch.int32, device=dev), gate_obs=torch.zeros(nb, dtype=torch.int32, device=dev), claim=torch.zeros(2, dtype=torch.int32, device=dev))


# (n_envs, n_ue, T, hidden, two tables): whole pairs of blocks; a ragged last block and a lone last block; 40 UEs (one env per wavefront,
# 44 nodes); one table; a node count without an instantiation of its own (36: run-time loop); more pairs than a small grid would hold
SHAPES = [(64, 20, 6, 200, True), (200, 20, 5, 200, True), (16 * 3 + 5, 20, 4, 200, True), (40, 40, 4, 200, True), (96, 20, 3, 64, False), (48, 32, 3, 200, True),
          (8192, 20, 3, 200, True)]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "%denv_%due_T%d_h%d_%s" % (s[0], s[1], s[2], s[3], "two" if s[4] else "one"))
def test_gated_rollout_with_open_gates_equals_steps_plus_first_layer(shape):
    torch = _torch()
    from drl_uav_cellularnet_amd import _agent_capi as A

    n, n_ue, T, hid, two = shape
    env = _env(n, n_ue)
    ref = env.clone()
    wa, ba, wc, bc = _tables(torch, env, hid, 3)
    g = torch.Generator().manual_seed(11)
    act = torch.randint(0, env.action_space_dim, (T, n), generator=g, dtype=torch.int64).to(env.device)
    b = _buffers(torch, env, T, hid)
    env.rollout_gated(act, b["gate_act"], b["gate_obs"], b["claim"][0:1], wa, ba, b["out_a"], wc if two else None, bc if two else None, b["out_c"] if two else None,
                      idx_out=b["idx"], reward_out=b["rew"])
    torch.cuda.synchronize()
    assert env.device_error() == 0
    K = env.nBS + n_ue
    for t in range(T):
        r = ref.step(act[t])
        assert torch.equal(b["rew"][t], ref.out["reward"]), "reward of step %d" % t
        ea = torch.empty((n, hid), device=env.device)
        ec = torch.empty((n, hid), device=env.device)
        ei = torch.empty((n, K), dtype=torch.int64, device=env.device)
        A.first_layer_from_obs(ref.observation(), env.grid_n, wa, ba, wc if two else None, bc if two else None, ea, ec if two else None, idx_out=ei)
        assert torch.equal(b["idx"][t + 1], ei), "index list after step %d" % t
        if t + 1 < T:
            assert torch.equal(b["out_a"][t + 1], ea), "table a, slot %d" % (t + 1)
            if two:
                assert torch.equal(b["out_c"][t + 1], ec), "table c, slot %d" % (t + 1)
        del r
    for k, v in ref.out.items():
        if k != "reward":
            assert torch.equal(env.out[k], v), k
    assert np.array_equal(env.get_state(), ref.get_state())
    assert bool((b["gate_obs"] == (T if T > 1 else 0)).all())
    assert bool(torch.isnan(b["out_a"][0]).all()) and bool((b["idx"][0] == -7).all())      # slot 0 is the caller's
    # and the API continues from there
    a = torch.randint(0, env.action_space_dim, (n,), generator=g, dtype=torch.int64).to(env.device)
    env.step(a); ref.step(a)
    assert np.array_equal(env.get_state(), ref.get_state())


def test_a_gate_that_never_opens_is_an_error_code_not_a_hang(monkeypatch):
    torch = _torch()
    from drl_uav_cellularnet_amd import UavEnvError

    monkeypatch.setenv("UAVENV_HANDOFF_SPIN_US", "20000")            # 20 ms (read once in uavenv_create)
    env = _env(64, 20)
    state = env.get_state()
    T, hid = 3, 200
    wa, ba, wc, bc = _tables(torch, env, hid, 3)
    act = torch.zeros((T, 64), dtype=torch.int64, device=env.device)
    b = _buffers(torch, env, T, hid)
    b["gate_act"].fill_(1)                                           # step 0 may run; the actions of step 1 never come
    env.rollout_gated(act, b["gate_act"], b["gate_obs"], b["claim"][0:1], wa, ba, b["out_a"], wc, bc, b["out_c"], idx_out=b["idx"], reward_out=b["rew"])
    torch.cuda.synchronize()
    assert env.device_error() == 0x47415445
    with pytest.raises(UavEnvError):
        env.step(act[0])
    env.set_state(state)                                             # a whole state again: the handle works
    assert env.device_error() == 0
    env.step(act[0])
    torch.cuda.synchronize()


def test_gated_rollout_refuses_what_it_was_not_built_for():
    torch = _torch()
    from drl_uav_cellularnet_amd import BatchedMobiEnv, UavEnvError

    env = BatchedMobiEnv(32, nBS=3, nUE=20, grid_n=100, groups=[5, 5, 5, 5], seed=1)
    T, hid = 2, 200
    rows = 4 * 100 * 100
    wa = torch.zeros((rows, hid), device=env.device)
    b = _buffers(torch, env, T, hid)
    act = torch.zeros((T, 32), dtype=torch.int64, device=env.device)
    with pytest.raises(UavEnvError):
        env.rollout_gated(act, b["gate_act"], b["gate_obs"], b["claim"][0:1], wa, None, b["out_a"])
    env4 = _env(32, 20)
    with pytest.raises(ValueError):
        env4.rollout_gated(act, b["gate_act"][:1], b["gate_obs"], b["claim"][0:1], wa, None, b["out_a"])


def _head_weights(torch, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    rnd = lambda *s: torch.rand(s, device="cuda", generator=g) * 2.0 - 1.0
    H, NA = 200, 625
    w2, b2, w3, b3 = rnd(H, H) * 0.2, rnd(H), rnd(H, NA) * 0.3, rnd(NA)
    w3t, b3p = torch.zeros((640, H), device="cuda"), torch.zeros(640, device="cuda")
    w3t[:NA], b3p[:NA] = w3.t(), b3
    return w2.t().contiguous(), b2, w3t, b3p, NA, g


@pytest.mark.parametrize("shape", [(64, 5), (200, 3), (52, 4), (8192, 3)], ids=lambda s: "%drows_T%d" % s)
def test_gated_head_with_open_gates_equals_the_head_of_every_step(shape):
    """uavagent_actor_head_gated_f32 alone: h1 of all T steps is there and every observation gate open before the launch; the persistent kernel
    must leave the h2, logits and actions of T calls of uavagent_actor_head_f32, bit for bit, and the action gates at T."""
    torch = _torch()
    from drl_uav_cellularnet_amd import _agent_capi as A

    n, T = shape
    w2t, b2, w3t, b3p, NA, g = _head_weights(torch, n)
    h1 = ((torch.rand((T, n, 200), device="cuda", generator=g) * 2.0 - 1.0) * 4.0 + 2.0).clamp_(0.0, 6.0)
    u = torch.rand((T, n), device="cuda", generator=g)
    nb = (n + 15) // 16
    h2, lg = torch.full((T, n, 200), float("nan"), device="cuda"), torch.full((T, n, 640), float("nan"), device="cuda")
    act = torch.full((T, n), -1, dtype=torch.int64, device="cuda")
    gate_obs, gate_act = torch.full((nb,), T, dtype=torch.int32, device="cuda"), torch.zeros(nb, dtype=torch.int32, device="cuda")
    claim = torch.zeros(1, dtype=torch.int32, device="cuda")
    A.gate_prepare()
    A.device_error_clear()
    A.actor_head_gated(h1, w2t, b2, w3t, b3p, u, NA, h2, lg, act, gate_obs, gate_act, claim)
    torch.cuda.synchronize()
    assert A.device_error() == 0
    for t in range(T):
        h2r, lgr = torch.empty((n, 200), device="cuda"), torch.empty((n, 640), device="cuda")
        ar = torch.empty(n, dtype=torch.int64, device="cuda")
        A.actor_head(h1[t], w2t, b2, w3t, b3p, u[t], NA, h2r, lgr, ar)
        assert torch.equal(h2[t], h2r) and torch.equal(lg[t], lgr) and torch.equal(act[t], ar), "step %d" % t
    assert bool((gate_act == T).all())


def test_gated_head_gives_up_on_a_gate_that_never_opens():
    torch = _torch()
    from drl_uav_cellularnet_amd import _agent_capi as A

    n, T = 64, 3
    w2t, b2, w3t, b3p, NA, g = _head_weights(torch, 5)
    h1 = torch.rand((T, n, 200), device="cuda", generator=g)
    u = torch.rand((T, n), device="cuda", generator=g)
    h2, lg = torch.empty((T, n, 200), device="cuda"), torch.empty((T, n, 640), device="cuda")
    act = torch.empty((T, n), dtype=torch.int64, device="cuda")
    gate_obs, gate_act = torch.ones(4, dtype=torch.int32, device="cuda"), torch.zeros(4, dtype=torch.int32, device="cuda")
    claim = torch.zeros(1, dtype=torch.int32, device="cuda")
    A.gate_prepare()
    A.device_error_clear()
    A.actor_head_gated(h1, w2t, b2, w3t, b3p, u, NA, h2, lg, act, gate_obs, gate_act, claim, spin_us=20000)
    torch.cuda.synchronize()
    assert A.device_error() == 0x47415445
    assert bool((gate_act == 1).all())                               # step 0 ran
    with pytest.raises(A.UavAgentError):
        A.actor_head_gated(h1, w2t, b2, w3t, b3p, u, NA, h2, lg, act, gate_obs, gate_act, claim)
    A.device_error_clear()
    assert A.device_error() == 0


@pytest.mark.parametrize("shape", [(64, 20, 6), (8192, 20, 8), (200, 20, 5), (4096 + 16, 20, 4)], ids=lambda s: "%denv_%due_T%d" % s)
def test_the_two_persistent_kernels_together_equal_the_step_by_step_rollout(shape):
    """The pair: uavenv_rollout_gated on one stream, uavagent_actor_head_gated_f32 on another, gates closed -- each kernel waits for the other,
    step by step and block by block.  Against the same rollout made of single launches (first layer from the observation, actor head, env step):
    indices, first-layer rows, h2, logits, actions, rewards, env outputs and state bit for bit."""
    torch = _torch()
    from drl_uav_cellularnet_amd import _agent_capi as A

    n, n_ue, T = shape
    env = _env(n, n_ue)
    ref = env.clone()
    dev = env.device
    hid, K = 200, env.nBS + n_ue
    wa, ba, wc, bc = _tables(torch, env, hid, 3)
    w2t, b2, w3t, b3p, NA, g = _head_weights(torch, n + T)
    u = torch.rand((T, n), device=dev, generator=g)
    b = _buffers(torch, env, T, hid)
    h2, lg = torch.full((T, n, hid), float("nan"), device=dev), torch.full((T, n, 640), float("nan"), device=dev)
    act = torch.full((T, n), -1, dtype=torch.int64, device=dev)
    # slot 0: the observation the rollout starts from
    A.first_layer_from_obs(env.observation(), env.grid_n, wa, ba, wc, bc, b["out_a"][0], b["out_c"][0], idx_out=b["idx"][0])
    b["gate_obs"].fill_(1); b["gate_act"].zero_()
    A.gate_prepare()
    A.device_error_clear()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev, priority=-1)    # its own hardware queue (priority levels have separate queues): the two kernels wait for each other
    side.wait_stream(torch.cuda.current_stream(dev))
    A.actor_head_gated(b["out_a"], w2t, b2, w3t, b3p, u, NA, h2, lg, act, b["gate_obs"], b["gate_act"], b["claim"][1:2])
    with torch.cuda.stream(side):
        env.rollout_gated(act, b["gate_act"], b["gate_obs"], b["claim"][0:1], wa, ba, b["out_a"], wc, bc, b["out_c"], idx_out=b["idx"], reward_out=b["rew"])
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    assert A.device_error() == 0 and env.device_error() == 0
    ea, ec = torch.empty((n, hid), device=dev), torch.empty((n, hid), device=dev)
    ei = torch.empty((n, K), dtype=torch.int64, device=dev)
    h2r, lgr, ar = torch.empty((n, hid), device=dev), torch.empty((n, 640), device=dev), torch.empty(n, dtype=torch.int64, device=dev)
    for t in range(T):
        A.first_layer_from_obs(ref.observation(), env.grid_n, wa, ba, wc, bc, ea, ec, idx_out=ei)
        assert torch.equal(b["idx"][t], ei) and torch.equal(b["out_a"][t], ea) and torch.equal(b["out_c"][t], ec), "first layer, step %d" % t
        A.actor_head(ea, w2t, b2, w3t, b3p, u[t], NA, h2r, lgr, ar)
        assert torch.equal(h2[t], h2r) and torch.equal(lg[t], lgr) and torch.equal(act[t], ar), "head, step %d" % t
        ref.step(ar)
        assert torch.equal(b["rew"][t], ref.out["reward"]), "reward, step %d" % t
    A.obs_indices(ref.observation(), env.grid_n, env.nBS, out=ei)
    assert torch.equal(b["idx"][T], ei)
    for k, v in ref.out.items():
        if k != "reward":
            assert torch.equal(env.out[k], v), k
    assert np.array_equal(env.get_state(), ref.get_state())


def test_rows_beyond_the_table_contribute_nothing():
    """include/uavenv.h: a node whose row index is >= enc_rows contributes nothing (like a node off the grid) and is never dereferenced.  A table
    that holds planes 0 and 1 only (the UAVs, and the UEs served by UAV 0): against the same sum formed step by step in torch, float32, node
    order -- exactly."""
    torch = _torch()
    n, T, hid = 48, 3, 64
    env = _env(n, 20)
    ref = env.clone()
    G, K = env.grid_n, env.nBS + env.nUE
    rows = 2 * G * G
    g = torch.Generator().manual_seed(5)
    wa = (torch.rand(rows, hid, generator=g) - 0.5).to(env.device)
    ba = (torch.rand(hid, generator=g) - 0.5).to(env.device)
    act = torch.randint(0, env.action_space_dim, (T, n), generator=g, dtype=torch.int64).to(env.device)
    b = _buffers(torch, env, T, hid)
    env.rollout_gated(act, b["gate_act"], b["gate_obs"], b["claim"][0:1], wa, ba, b["out_a"], idx_out=b["idx"], reward_out=b["rew"], relu6=False)
    torch.cuda.synchronize()
    assert env.device_error() == 0
    seen_beyond = 0
    for t in range(T - 1):
        ref.step(act[t])
        idx = b["idx"][t + 1]                                       # checked against the learner's index kernel in the tests above
        s = torch.zeros((n, hid), device=env.device)
        for k in range(K):
            ok = (idx[:, k] >= 0) & (idx[:, k] < rows)
            s = s + torch.where(ok[:, None], wa[idx[:, k].clamp(0, rows - 1)], torch.zeros((), device=env.device))
        seen_beyond += int((idx >= rows).sum())
        assert torch.equal(b["out_a"][t + 1], s + ba), "slot %d" % (t + 1)
    assert seen_beyond > 0                                          # UEs served by UAVs 1-3 do occur: the case was exercised
