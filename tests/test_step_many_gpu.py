"""uavenv_step_many (T steps in one launch, state carried in registers) and hipGraph replay of T step() launches against
T plain uavenv_step calls from the same state: every output of every step and the final state blob BIT-IDENTICAL.
(uavenv_step itself is checked against the reference fixtures and the oracle in test_hip_parity.py; this file adds the
oracle at a few points so that the multi-step path is also pinned directly.)  Reference: mobile_env.py:150-194 called T times."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _torch():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch


def _env(n, n_bs, n_ue, **kw):
    from drl_uav_cellularnet_amd import BatchedMobiEnv

    groups = [n_ue // 4] * 3 + [n_ue - 3 * (n_ue // 4)]
    return BatchedMobiEnv(n, nBS=n_bs, nUE=n_ue, grid_n=kw.pop("grid_n", 100), groups=groups, seed=kw.pop("seed", 77), **kw)


def _actions(torch, env, T, seed):
    g = torch.Generator().manual_seed(seed)
    hi = min(env.action_space_dim, 2 ** 62)
    return torch.randint(0, hi, (T, env.n_envs), generator=g, dtype=torch.int64).to(env.device)


# (n_envs, n_bs, n_ue, T): packed FAST (3 envs / wavefront, ragged tail), 1 env / wavefront, B != BT (checked variant),
# B > 8 (cooperative UAV move through LDS inside the step loop), multi-pass (host loop of single-step launches)
SHAPES = [(100, 4, 20, 1), (100, 4, 20, 2), (100, 4, 20, 7), (1366 * 3 - 1, 4, 20, 5), (50, 4, 40, 9), (33, 3, 20, 6), (20, 8, 20, 5),
          (10, 16, 60, 6), (4, 16, 200, 3)]


@pytest.mark.parametrize("pin", [None, "0", "1"], ids=["auto", "unpinned", "pinned"])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "%denv_%dx%d_T%d" % s)
def test_step_many_is_bit_identical_to_single_steps(shape, pin, monkeypatch):
    """`pin`: the launcher's own choice, or UAVENV_FORCE_PIN (read once in uavenv_create, so set BEFORE the env exists): "0" runs
    env_kernel_packed<..., FAST, PIN = false, MANY = true>, the instantiation uavenv_step_many selects above two wavefronts per
    SIMD (> 6144 envs at 20 UEs) -- here on small batches; tests/test_full_size_parity_gpu.py runs it at 8192 / 65536 envs."""
    torch = _torch()
    if pin is not None:
        monkeypatch.setenv("UAVENV_FORCE_PIN", pin)
    n, n_bs, n_ue, T = shape
    env = _env(n, n_bs, n_ue, f64_outputs=(n_bs == 3))
    ref = env.clone()
    act = _actions(torch, env, T, 5)
    many = env.step_many(act)
    for t in range(T):
        ref.step(act[t])
        for k, v in ref.out.items():
            assert torch.equal(many[k][t], v), "%s differs at step %d" % (k, t)
    for k, v in ref.out.items():
        assert torch.equal(env.out[k], v), k                       # self.out refreshed with the last block
    assert np.array_equal(env.get_state(), ref.get_state())
    # and the API continues from there
    a = _actions(torch, env, 1, 9)[0]
    env.step(a); ref.step(a)
    for k, v in ref.out.items():
        assert torch.equal(env.out[k], v), k


# (n_envs, n_bs, n_ue, T, slots): W = ceil(n_envs / envs per wavefront) env-wavefronts planned onto k x `slots` slots, k = W // slots
ROT_SHAPES = [(100, 4, 20, 7, 24), (100, 4, 20, 50, 26), (301, 4, 20, 33, 80), (50, 4, 40, 9, 37), (33, 3, 20, 6, 8), (20, 8, 20, 5, 5),
              (10, 16, 60, 6, 7), (64, 4, 20, 100, 16),
              (100, 4, 20, 21, 13), (301, 4, 20, 40, 29), (64, 4, 20, 12, 5)]      # k = 2, 3, 4 resident wavefronts per pretend-SIMD


@pytest.mark.parametrize("shape", ROT_SHAPES, ids=lambda s: "%denv_%dx%d_T%d_S%d" % s)
def test_rotation_schedule_is_bit_identical_to_the_plain_launch(shape, monkeypatch):
    """A multi-step call on S < W < 2 S wavefronts runs as ONE launch of S persistent wavefronts, each working through up to three pieces
    (env-wavefront, first step, steps) of a wrap-around schedule (csrc/uavenv_capi.hip: rotation_plan); the two wavefronts that share
    a split job hand its state over through memory + a flag.  Same steps, same order per env: every output of every step and the final state must equal the
    single plain launch.  UAVENV_ROTATE_SLOTS makes small batches plan as if the device had that few SIMDs; at BASELINE's 4096 envs
    the schedule is chosen automatically (tests/test_full_size_parity_gpu.py compares that run with the oracle)."""
    torch = _torch()
    import ctypes as C

    n, n_bs, n_ue, T, slots = shape
    monkeypatch.setenv("UAVENV_ROTATE", "1")
    monkeypatch.setenv("UAVENV_ROTATE_SLOTS", str(slots))
    env = _env(n, n_bs, n_ue)
    monkeypatch.setenv("UAVENV_ROTATE", "0")
    ref = env.clone()
    W = -(-n // max(1, 64 // n_ue)) if n_ue <= 64 else n
    nl, sl = C.c_int(-1), C.c_longlong(-1)
    assert env._lib.uavenv_debug_rotation_info(env._h, T, C.byref(nl), C.byref(sl)) == 0
    assert sl.value == (W // slots) * slots, (nl.value, sl.value)
    assert nl.value == 1, nl.value                                              # the rotated handle really rotates ...
    assert ref._lib.uavenv_debug_rotation_info(ref._h, T, C.byref(nl), C.byref(sl)) == 0 and nl.value == 0    # ... the reference does not
    act = _actions(torch, env, T, 8)
    for rep in range(3):                                                        # later calls: cached schedule, flags cleared by their consumers
        got, want = env.step_many(act), ref.step_many(act)
        for k in want:
            assert torch.equal(got[k], want[k]), "%s differs (call %d)" % (k, rep)
        assert np.array_equal(env.get_state(), ref.get_state())
        assert env.device_error() == 0


def test_one_launch_rotation_replays_from_a_captured_graph(monkeypatch):
    """The hand-off flags are cleared by the wavefront that consumed them, so a captured multi-step launch replays correctly (a
    per-call epoch in the kernel arguments would not); a schedule that does not exist yet is not built inside a capture (the plain
    launch is captured instead)."""
    torch = _torch()
    import ctypes as C

    monkeypatch.setenv("UAVENV_ROTATE", "1")
    monkeypatch.setenv("UAVENV_ROTATE_SLOTS", "24")
    env = _env(100, 4, 20)
    cold = env.clone()                                            # never prepared: its capture must fall back to the plain launch
    monkeypatch.setenv("UAVENV_ROTATE", "0")
    ref = env.clone()
    T = 9
    act = _actions(torch, env, T, 5)
    assert env._lib.uavenv_step_many_prepare(env._h, T) == 0
    outs = {}
    graphs = {}
    for name, e in (("warm", env), ("cold", cold)):
        outs[name] = e.step_many(act) if name == "warm" else {k: torch.empty((T,) + tuple(v.shape), dtype=v.dtype, device=e.device) for k, v in e.out.items()}
        if name == "warm":
            want = ref.step_many(act)
            for k in want:
                assert torch.equal(outs[name][k], want[k])
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                e.step_many(act, out=outs[name], refresh_out=False)
        graphs[name] = g
    nl = C.c_int(-1)
    assert cold._lib.uavenv_debug_rotation_info(cold._h, T, C.byref(nl), None) == 0 and nl.value == 1     # (built now, outside the capture)
    cold_ref = ref.clone()
    cold_ref.set_state(cold.get_state())
    for rep in range(3):
        graphs["warm"].replay()
        want = ref.step_many(act)
        torch.cuda.synchronize()
        for k in want:
            assert torch.equal(outs["warm"][k], want[k]), "%s differs (replay %d)" % (k, rep)
        assert np.array_equal(env.get_state(), ref.get_state())
        graphs["cold"].replay()
        want = cold_ref.step_many(act)
        torch.cuda.synchronize()
        for k in want:
            assert torch.equal(outs["cold"][k], want[k]), "%s differs (cold replay %d)" % (k, rep)
    assert env.device_error() == 0 and cold.device_error() == 0


def test_a_hand_off_that_is_never_signalled_becomes_an_error_code_not_a_hang(monkeypatch):
    """UAVENV_DEBUG_DROP_PUBLISH builds a schedule whose publishing pieces never set their flag.  The waiting wavefronts give up after
    the spin budget (2 ms here), leave UAVENV_DEV_ERR_HANDOFF in the handle's host-mapped error word and exit; the launch ends, every
    later call on the handle fails with UAVENV_E_DEVICE, and set_state() makes it usable again."""
    torch = _torch()
    from drl_uav_cellularnet_amd import _capi

    monkeypatch.setenv("UAVENV_ROTATE", "1")
    monkeypatch.setenv("UAVENV_ROTATE_SLOTS", "24")
    good = _env(100, 4, 20)
    monkeypatch.setenv("UAVENV_DEBUG_DROP_PUBLISH", "1")
    monkeypatch.setenv("UAVENV_HANDOFF_SPIN_US", "2000")
    bad = good.clone()
    monkeypatch.delenv("UAVENV_DEBUG_DROP_PUBLISH")
    state0 = good.get_state()
    act = _actions(torch, good, 7, 2)
    want = good.step_many(act)
    assert good.device_error() == 0
    bad.step_many(act, refresh_out=False)                        # the launch itself is asynchronous: it returns 0
    torch.cuda.synchronize()                                      # ... and ENDS (bounded wait), with the error word set
    assert bad.device_error() == 0x48414E44
    with pytest.raises(_capi.UavEnvError, match="hand-off"):
        bad.step(act[0])
    with pytest.raises(_capi.UavEnvError):
        bad.get_state()
    bad.set_state(state0)                                         # a whole state again: the error is cleared ...
    assert bad.device_error() == 0
    monkeypatch.setenv("UAVENV_ROTATE", "0")
    ok = bad.clone()                                              # ... and a handle without the broken schedule continues from it
    got = ok.step_many(act)
    for k in want:
        assert torch.equal(got[k], want[k]), k


def test_rotation_is_automatic_where_it_pays_and_off_elsewhere():
    """Automatic use (no UAVENV_ROTATE): k = W // SIMDs resident wavefronts per SIMD with 1 <= k <= 2 (the pinned kernel's occupancy),
    W not a multiple of the SIMD count, W / SIMDs <= 1.45 when k = 1, at least 20 steps per call."""
    torch = _torch()
    import ctypes as C

    n_simd = 4 * torch.cuda.get_device_properties(0).multi_processor_count
    nl, sl = C.c_int(-1), C.c_longlong(-1)
    for n, T in ((4096, 100), (4096, 48), (4096, 20), (4096, 19), (4096, 8), (3072, 100), (3500, 100), (5400, 100), (6144, 100), (8192, 100), (8192, 20), (9100, 100),
                 (12288, 100), (1536, 100)):
        env = _env(n, 4, 20)
        assert env._lib.uavenv_debug_rotation_info(env._h, T, C.byref(nl), C.byref(sl)) == 0
        waves = (n + 2) // 3
        k = waves // n_simd
        want = 1 <= k <= 2 and waves > k * n_simd and (k == 2 or 100 * waves <= 145 * n_simd) and T >= 20
        assert (nl.value == 1) == want and nl.value in (0, 1), (n, T, nl.value, n_simd)
        if nl.value:
            assert sl.value == k * n_simd


@pytest.mark.parametrize("shape", [(100, 4, 20, 33), (100, 4, 20, 32), (100, 4, 20, 49), (50, 4, 40, 7), (10, 16, 200, 4), (20, 8, 20, 9), (4097, 4, 20, 2048)],
                         ids=lambda s: "%denv_%dx%d_cut%d" % s)
def test_step_range_over_two_ranges_equals_step(shape):
    """uavenv_step_range on [0, cut) and [cut, N) -- on two streams, concurrently -- leaves the outputs and the state one uavenv_step of
    the whole batch leaves (what the A2C rollout's two half-batches rely on), also when the cut falls inside a wavefront's envs (that
    wavefront then runs in both launches, each with its own envs live)."""
    torch = _torch()
    from drl_uav_cellularnet_amd import _capi

    n, n_bs, n_ue, cut = shape
    env = _env(n, n_bs, n_ue)
    ref = env.clone()
    act = _actions(torch, env, 5, 11)
    s1 = torch.cuda.Stream()
    for t in range(5):
        ref.step(act[t])
        s1.wait_stream(torch.cuda.current_stream())
        env.step_range(act[t], 0, cut)
        with torch.cuda.stream(s1):
            env.step_range(act[t], cut, n - cut)
        torch.cuda.current_stream().wait_stream(s1)
        for k in ref.out:
            assert torch.equal(env.out[k], ref.out[k]), "%s differs at step %d" % (k, t)
    assert np.array_equal(env.get_state(), ref.get_state())
    with pytest.raises(_capi.UavEnvError, match="range"):
        env.step_range(act[0], 0, n + 1)
    with pytest.raises(_capi.UavEnvError, match="range"):
        env.step_range(act[0], n, 1)
    with pytest.raises(_capi.UavEnvError, match="range"):
        env.step_range(act[0], 0, 0)


def test_one_launch_rotation_under_uneven_load_many_calls(monkeypatch):
    """Hand-offs under UNEVEN load (MI355X_MICROARCH.md: 'test every hand-off under uneven load ... checking every word'): 60 scheduled 20-step
    calls at BASELINE's 4096 envs while a second stream keeps part of the chip busy with large GEMMs and copies; every output word of every
    call and the state after the last one must equal the plain launch's (which ran alone).  A stale or early hand-off would show as a
    difference in some env's trajectory from that call on."""
    torch = _torch()
    env = _env(4096, 4, 20)                                       # default handle: the schedule is automatic from 20 steps per call on
    monkeypatch.setenv("UAVENV_ROTATE", "0")
    ref = env.clone()
    import ctypes as C

    nl = C.c_int(-1)
    assert env._lib.uavenv_debug_rotation_info(env._h, 20, C.byref(nl), None) == 0
    if torch.cuda.get_device_properties(0).multi_processor_count == 256:
        assert nl.value == 1
    acts = [_actions(torch, env, 20, 100 + i) for i in range(6)]
    want = []
    for i in range(60):
        want.append({k: v.clone() for k, v in ref.step_many(acts[i % 6]).items()})
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device="cuda")
    big = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    stop = 60
    for i in range(stop):
        with torch.cuda.stream(side):                             # load that comes and goes: a GEMM, then a 64 MB copy
            if i % 3 != 2:
                torch.mm(a, a)
            if i % 2 == 0:
                big.copy_(big.flip(0)[: big.numel()])
        got = env.step_many(acts[i % 6])
        for k in want[i]:
            assert torch.equal(got[k], want[i][k]), "%s differs in call %d" % (k, i)
    torch.cuda.synchronize()
    assert np.array_equal(env.get_state(), ref.get_state()) and env.device_error() == 0
