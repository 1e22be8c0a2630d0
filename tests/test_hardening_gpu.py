"""GPU tests of the C-ABI's defensive behaviour (no reference counterpart: the reference has no FFI layer)."""
import pytest

pytestmark = pytest.mark.gpu


def _torch():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch


def test_obs_dense_update_refuses_a_buffer_it_did_not_write():
    """uavenv_obs_dense_update applies +-1 deltas against the cell list of the LAST full write: on any other buffer that would
    silently give wrong (even negative) counts, so the handle remembers the pointer and refuses."""
    torch = _torch()
    from drl_uav_cellularnet_amd import BatchedMobiEnv, UavEnvError

    env = BatchedMobiEnv(8, nBS=4, nUE=20, grid_n=100)
    a = env.dense_obs()
    env.step(torch.zeros(8, dtype=torch.int64, device=env.device))
    env.dense_obs_update(a)                                   # the buffer dense_obs wrote: fine
    b = torch.zeros_like(a)
    with pytest.raises(UavEnvError, match="not the buffer"):
        env.dense_obs_update(b)
    env.dense_obs(out=b)                                      # a full write re-targets the handle
    env.step(torch.ones(8, dtype=torch.int64, device=env.device))
    env.dense_obs_update(b)
    assert torch.equal(b, env.dense_obs())
    with pytest.raises(UavEnvError, match="not the buffer"):
        env.dense_obs_update(a)


def test_create_and_destroy_keep_the_callers_current_device():
    """uavenv_create / uavenv_destroy run under a device guard (destroy is called from __del__ at GC time)."""
    torch = _torch()
    from drl_uav_cellularnet_amd import BatchedMobiEnv

    before = torch.cuda.current_device()
    env = BatchedMobiEnv(4, nBS=4, nUE=20, grid_n=100, device="cuda:0")
    assert torch.cuda.current_device() == before
    env.close()
    assert torch.cuda.current_device() == before


def test_first_state_zeros_gives_the_bias():
    """a2c_single_thread.py:155: the first sample of training is the all-zero env.state, so the first layer returns its bias."""
    torch = _torch()
    from drl_uav_cellularnet_amd import BatchedMobiEnv
    from drl_uav_cellularnet_amd.agent import A2CRunner

    env = BatchedMobiEnv(16, nBS=4, nUE=20, grid_n=100)
    r = A2CRunner(env, rollout=2, first_state="zeros")
    assert bool((r.idx == -1).all())
    with torch.no_grad():
        r.net.c_b1.add_(0.25)
        v0 = r.net.critic_only(r.idx)
        h = torch.nn.functional.relu6(r.net.c_b1)
        want = torch.nn.functional.relu6(h @ r.net.c_w2 + r.net.c_b2) @ r.net.c_w3 + r.net.c_b3
    torch.testing.assert_close(v0, want.expand_as(v0), rtol=1e-6, atol=1e-6)
    stats = r.train_rollout()
    assert all(map(lambda k: stats[k] == stats[k], ("a_loss", "c_loss")))
    assert bool((r.idx >= 0).all())
