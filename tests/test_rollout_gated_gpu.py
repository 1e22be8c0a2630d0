"""uavenv_rollout_gated (include/uavenv.h): T steps + the encoded observation after each, in one persistent launch gated on per-block step
counters.  Here the ENV side alone: the action tape is filled and every action gate opened before the launch, so the kernel runs its T steps
without a partner -- and must leave exactly what T calls of uavenv_step + uavagent_first_layer_from_obs_f32 leave: rewards, outputs, state
blob, index lists, encoded rows, all BIT-IDENTICAL, and the observation gates at T.  With a closed gate it must give up within its spin
budget and poison the handle (UAVENV_E_DEVICE), never hang.  The pair of kernels is tested in tests/test_learner_kernels_gpu.py.
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
                gate_act=torch.full((nb,), T, dtype=torch.int32, device=dev), gate_obs=torch.zeros(nb, dtype=torch.int32, device=dev))


# (n_envs, n_ue, T, hidden, two tables): whole pairs of blocks; a ragged last block and a lone last block; 40 UEs (one env per wavefront,
# run-time node count); one table; more pairs than a small grid would hold
SHAPES = [(64, 20, 6, 200, True), (200, 20, 5, 200, True), (16 * 3 + 5, 20, 4, 200, True), (40, 40, 4, 200, True), (96, 20, 3, 64, False),
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
    env.rollout_gated(act, b["gate_act"], b["gate_obs"], wa, ba, b["out_a"], wc if two else None, bc if two else None, b["out_c"] if two else None,
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
    env.rollout_gated(act, b["gate_act"], b["gate_obs"], wa, ba, b["out_a"], wc, bc, b["out_c"], idx_out=b["idx"], reward_out=b["rew"])
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
        env.rollout_gated(act, b["gate_act"], b["gate_obs"], wa, None, b["out_a"])
    env4 = _env(32, 20)
    with pytest.raises(ValueError):
        env4.rollout_gated(act, b["gate_act"][:1], b["gate_obs"], wa, None, b["out_a"])
