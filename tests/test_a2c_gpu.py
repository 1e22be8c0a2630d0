"""GPU: the A2C learner end to end on the HIP env (BASELINE config 3 shape, scaled down): rollouts through
BatchedMobiEnv.step (FAST kernel), sparse first layer on the compact observation, one update per rollout."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_a2c_rollouts_and_updates_on_the_hip_env():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from drl_uav_cellularnet_amd import BatchedMobiEnv
    from drl_uav_cellularnet_amd.agent import A2CRunner, ACNet, obs_to_indices

    N, T = 512, 10
    env = BatchedMobiEnv(N, nBS=4, nUE=20, grid_n=100, max_step=25)        # short episodes: exercises done + masked reset
    runner = A2CRunner(env, rollout=T, update_chunk=2048)
    assert sum(p.numel() for p in runner.net.parameters()) == 20206626
    # sparse path == dense path on the env's own dense observation
    idx = obs_to_indices(env.observation(), 100, 4)
    dense = env.dense_obs().reshape(N, -1)
    with torch.no_grad():
        p_s, v_s = runner.net(idx[:32])
        p_d, v_d = runner.net.forward_dense(dense[:32])
    torch.testing.assert_close(p_s, p_d, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(v_s, v_d, rtol=1e-4, atol=1e-4)
    w0 = runner.net.a_w3.detach().clone()
    steps = []
    for it in range(4):
        st = runner.train_rollout()
        assert np.isfinite(st["a_loss"]) and np.isfinite(st["c_loss"]) and st["grad_elems"] == 20206626
        steps.append(int(env.out["step_n"].max()))
    # done turns true at step 25 inside rollout 3 and stays true (step_n >= MAXSTEP, no auto-reset); like Worker.work
    # (a2c_single_thread.py:158-172) the rollout runs on to step 30 and the env is reset AFTER it -> step_n 0, then 10
    assert steps == [10, 20, 0, 10]
    assert runner.running_r is not None        # an episode finished: GLOBAL_RUNNING_R bookkeeping ran
    assert not torch.equal(w0, runner.net.a_w3.detach())


def test_training_tool_writes_the_reference_artifacts(tmp_path):
    """tools/train_a2c.py = the loop of a2c_single_thread.py:107-137: one episode, then Global_return.npy + the actor parameters,
    which tools/run_eval.py's loader accepts."""
    import os
    import subprocess
    import sys

    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(tmp_path, "run")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "train_a2c.py"), "--out", out, "--workers", "96", "--episodes", "1",
                        "--rollout", "100"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    ret = np.load(os.path.join(out, "Global_return.npy"))
    assert ret.shape == (1,) and np.isfinite(ret).all()                    # one finished episode -> one running-return entry
    from drl_uav_cellularnet_amd.agent import ACNet, load_actor_npz

    net = load_actor_npz(ACNet(50000, 625), os.path.join(out, "Global_A_PARA.npz"))
    assert all(bool(torch.isfinite(q).all()) for q in net.actor_params())


def test_checkpoint_resume_is_bit_identical(tmp_path):
    """A2CRunner.state_dict / load_state_dict ("Checkpoint / resume": the reference saves the actor only): a run
    restored from a checkpoint continues exactly like the uninterrupted one -- weights, RMSProp accumulators, env batch, episode
    bookkeeping and the sampling generator all come back (every kernel on the path is deterministic)."""
    import os

    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from drl_uav_cellularnet_amd import BatchedMobiEnv
    from drl_uav_cellularnet_amd.agent import A2CRunner

    def fresh():
        env = BatchedMobiEnv(256, nBS=4, nUE=20, grid_n=100, max_step=12)       # episodes end inside the third rollout
        return A2CRunner(env, rollout=5)

    a = fresh()
    for _ in range(3):
        a.train_rollout()
    path = os.path.join(tmp_path, "ckpt.pt")
    torch.save(a.state_dict(), path)
    for _ in range(3):
        sa = a.train_rollout()
    b = fresh()
    b.load_state_dict(torch.load(path, weights_only=True))
    for _ in range(3):
        sb = b.train_rollout()
    assert torch.equal(a.flat.w, b.flat.w) and torch.equal(a.flat.ms, b.flat.ms)
    assert np.array_equal(a.env.get_state(), b.env.get_state()) and torch.equal(a.idx, b.idx)
    assert sa["a_loss"] == sb["a_loss"] and a.running_r == b.running_r and torch.equal(a.ep_r, b.ep_r)
