"""Oracle parity AT the benchmark's real grid sizes ("parity on first 64 steps" at N = 4096; VERDICT r1
"What's missing" #2).  The exact launch shapes bench.py times -- 4096 envs (the PINNED FAST kernel, 1366 wavefronts with a ragged
last one), 8192 and 65536 envs (the unpinned FAST kernel) -- run 64 steps on the device; the oracle replays 64-env WINDOWS of the
big batch (OracleEnv(..., 64, env_id_base=k) owns the same Philox streams as envs k..k+63 of one big batch) at the start, in the
middle and over the ragged last wavefront.  Integers exact, float32 within 1e-5 relative (north_star).
Also: the same comparison through uavenv_step_many and through a replayed hipGraph at 4096 envs, and BASELINE config 3 at its
full size (8192 envs x 50-step rollout + one update)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

STEPS, WIN = 64, 64
INT_KEYS = ("ue_xy", "bs_xy", "serving", "n_out", "step_n", "done")
F32_KEYS = ("cur_sinr", "mean_sinr", "reward")


def _torch():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch


def _windows(n):
    last = n - WIN                       # covers the last (ragged: 3 envs per wavefront, n % 3 != 0) wavefront
    return sorted({0, ((n // 2) // 3) * 3 + 1, last})     # the middle window starts inside a wavefront (slot 1)


def _oracles(n):
    from oracle import oracle as O

    cfg = O.make_config(4, 20, 100, groups=[5, 5, 5, 5])
    out = []
    for k in _windows(n):
        o = O.OracleEnv(cfg, WIN, seed=0x5EED, env_id_base=k)
        first = o.construct()
        out.append((k, o, first))
    return out


def _compare(got, want, k, what):
    for key in INT_KEYS:
        assert np.array_equal(got[key][k:k + WIN].cpu().numpy(), want[key]), "%s: %s differs, window %d" % (what, key, k)
    for key in F32_KEYS:
        np.testing.assert_allclose(got[key][k:k + WIN].cpu().numpy(), want[key], rtol=1e-5, atol=0,
                                   err_msg="%s: %s, window %d" % (what, key, k))


@pytest.mark.parametrize("n_envs", [4096, 8192, 65536])
def test_bench_sized_batches_match_the_oracle_on_windows(n_envs):
    torch = _torch()
    from drl_uav_cellularnet_amd import BatchedMobiEnv

    env = BatchedMobiEnv(n_envs, nBS=4, nUE=20, grid_n=100, groups=[5, 5, 5, 5], seed=0x5EED)
    orcs = _oracles(n_envs)
    for k, _, first in orcs:
        _compare(env.out, first, k, "constructor")
    gen = torch.Generator().manual_seed(1234)
    for t in range(STEPS):
        a = torch.randint(0, 625, (n_envs,), generator=gen, dtype=torch.int64)
        env.step(a.to(env.device))
        a_np = a.numpy()
        for k, o, _ in orcs:
            _compare(env.out, o.step(a_np[k:k + WIN]), k, "step %d" % t)


@pytest.mark.parametrize("n,rotate", [(4096, None), (4096, "0"), (8192, None), (8192, "0"), (65536, None)],
                         ids=["4096_auto_one_launch_rotation", "4096_plain_launch", "8192_auto_one_launch_rotation", "8192_plain_launch", "65536"])
def test_step_many_and_graph_at_bench_sizes_match_the_oracle_on_windows(n, rotate, monkeypatch):
    """4096 envs: the pinned multi-step kernel, as the library runs it by default (1366 env-wavefronts on 1024 SIMDs: the ONE-launch
    rotation schedule, 1024 persistent wavefronts with hand-offs) and as ONE plain launch (UAVENV_ROTATE=0).  8192 envs: 2731 env-wavefronts as 2048 slots of the pinned kernel (two per SIMD) by default, and the
    UNPINNED kernel as a plain launch; 65536: the unpinned kernel, the instantiation behind DESIGN section 5's 65 536-env `step_many` figure."""
    torch = _torch()
    import ctypes as C

    from drl_uav_cellularnet_amd import BatchedMobiEnv, _capi

    if rotate is not None:
        monkeypatch.setenv("UAVENV_ROTATE", rotate)
    census0 = {name: cnt for name, _, cnt in _capi.launch_census()}
    env_m = BatchedMobiEnv(n, nBS=4, nUE=20, grid_n=100, groups=[5, 5, 5, 5], seed=0x5EED)
    env_g = env_m.clone()
    orcs = _oracles(n)
    gen = torch.Generator().manual_seed(99)
    tape = torch.randint(0, 625, (STEPS, n), generator=gen, dtype=torch.int64)
    dev_tape = tape.to(env_m.device)
    nl = C.c_int(-1)
    assert env_m._lib.uavenv_debug_rotation_info(env_m._h, STEPS, C.byref(nl), None) == 0
    if torch.cuda.get_device_properties(0).multi_processor_count == 256:          # (MI355X: 1024 SIMDs)
        want = {(4096, None): 1, (4096, "0"): 0, (8192, None): 1, (8192, "0"): 0, (65536, None): 0}[(n, rotate)]
        assert nl.value == want, nl.value
    many = env_m.step_many(dev_tape)
    g = env_g.capture_steps(dev_tape)
    g.replay()
    torch.cuda.synchronize()
    a_np = tape.numpy()
    for t in range(STEPS):
        for k, o, _ in orcs:
            want = o.step(a_np[t, k:k + WIN])
            _compare({key: many[key][t] for key in INT_KEYS + F32_KEYS}, want, k, "step_many step %d" % t)
            if t == STEPS - 1:
                _compare(env_g.out, want, k, "graph replay, last step")
    assert np.array_equal(env_m.get_state(), env_g.get_state())
    assert env_m.device_error() == 0
    ran = [name for name, _, cnt in _capi.launch_census() if cnt > census0[name] and "MANY=1" in name]
    pinned = n == 4096 or (n == 8192 and rotate is None)          # resident wavefronts <= 2 per SIMD
    if torch.cuda.get_device_properties(0).multi_processor_count == 256:
        sched = ", SCHED=1" if (rotate is None and n in (4096, 8192)) else ""          # the schedule's own kernel instantiation
        assert ran == ["env_kernel_packed<BT=4, STEP, PLC=1, FAST=1, PIN=%d, MANY=1%s>" % (1 if pinned else 0, sched)], ran


@pytest.mark.parametrize("n_envs", [8192])
def test_config5_at_full_size_matches_the_oracle_on_windows(n_envs):
    """BASELINE configs[4]: 16 UAV x 200 UE, 8192 envs -- env_kernel_multipass<16, STEP, cube, FAST> at the 8192 wavefronts
    profiles/r02e_config5_kernel_stats.csv times -- 40 steps (the handover FIFO is full from step 2 on: ~38 steps of handover decisions)
    against 16-env oracle windows at the start, in the middle and at the end of the batch.  Integers (cells, serving UAV, outage count, step counter) EXACT over every window and step: the multi-pass
    kernel sums the interference of the best UAV in a different order than the reference, which could only
    show as a flipped handover / outage decision."""
    torch = _torch()
    from drl_uav_cellularnet_amd import BatchedMobiEnv, _capi
    from oracle import oracle as O

    B, U, Gd, W, T = 16, 200, 100, 16, 40
    bs_init = [(Gd // 8 + (b // 4) * (Gd // 4), Gd // 8 + (b % 4) * (Gd // 4)) for b in range(B)]      # 4 x 4 lattice
    census0 = {name: cnt for name, _, cnt in _capi.launch_census()}
    env = BatchedMobiEnv(n_envs, nBS=B, nUE=U, grid_n=Gd, groups=[50] * 4, bs_init=bs_init, seed=0x5EED)
    cfg = O.make_config(B, U, Gd, groups=[50] * 4, bs_init=bs_init)
    orcs = []
    for k in sorted({0, n_envs // 2 + 3, n_envs - W}):
        o = O.OracleEnv(cfg, W, seed=0x5EED, env_id_base=k)
        orcs.append((k, o, {kk: v.copy() for kk, v in o.construct().items()}))

    def cmp(want, k, what):
        for key in INT_KEYS:
            assert np.array_equal(env.out[key][k:k + W].cpu().numpy(), want[key]), "%s: %s differs, window %d" % (what, key, k)
        for key in F32_KEYS:
            np.testing.assert_allclose(env.out[key][k:k + W].cpu().numpy(), want[key], rtol=1e-5, atol=0,
                                       err_msg="%s: %s, window %d" % (what, key, k))

    for k, _, first in orcs:
        cmp(first, k, "constructor")
    gen = torch.Generator().manual_seed(4321)
    for t in range(T):
        digits = torch.randint(0, 5, (n_envs, B), generator=gen, dtype=torch.int64)
        a = torch.zeros(n_envs, dtype=torch.int64)
        for b in range(B):
            a = a * 5 + digits[:, b]                                  # joint action in [0, 5^16): needs the int64 decode
        env.step(a.to(env.device))
        a_np = a.numpy()
        for k, o, _ in orcs:
            cmp(o.step(a_np[k:k + W]), k, "step %d" % t)
    ran = [name for name, _, cnt in _capi.launch_census() if cnt > census0[name] and "STEP" in name]
    assert ran == ["env_kernel_multipass<BT=16, STEP, PLC=1, FAST=1>"], ran


def test_a2c_config3_at_full_size():
    """BASELINE configs[2]: 8192 envs + MLP actor-critic, one 50-step rollout and one update at full size."""
    torch = _torch()
    from drl_uav_cellularnet_amd import BatchedMobiEnv
    from drl_uav_cellularnet_amd.agent import A2CRunner

    env = BatchedMobiEnv(8192, nBS=4, nUE=20, grid_n=100, groups=[5, 5, 5, 5])
    runner = A2CRunner(env, rollout=50)
    w0 = runner.net.c_w2.detach().clone()
    for it in range(2):
        st = runner.train_rollout()
        assert np.isfinite(st["a_loss"]) and np.isfinite(st["c_loss"]) and np.isfinite(st["mean_reward"])
        assert st["grad_elems"] == 20206626
        assert int(env.out["step_n"].min()) == int(env.out["step_n"].max()) == 50 * (it + 1)
    assert not torch.equal(w0, runner.net.c_w2.detach())
    assert all(bool(torch.isfinite(p).all()) for p in runner.net.parameters())
