"""Every kernel instantiation the dispatch logic of libuavenv can select, launched against the oracle -- and a census that
proves none was left out (VERDICT r2 "What's missing" #1: the unpinned multi-step kernel carried quoted numbers and no test).

The env kernels are templates: (family packed / multi-pass) x (bound on n_bs: 4 / 8 / 16 / 32) x (mode: warm-up, reset, step,
trace step, trace reset) x (path-loss form: cube / generic) x (variant: checked / fast / pinned) x (multi-step or not);
`launch_env` (csrc/uavenv_capi.hip) picks one per call and counts it (uavenv_debug_variant_info).  The matrix below builds, for
every (family, n_bs bound, path-loss form), a small batch in each variant and drives constructor, step, uavenv_step_many,
masked reset, trace reset and trace step against OracleEnv on identical Philox streams: integers exact, float32 within 1e-5
relative (north_star), float64 copies within 1e-9.  The last test reads the census: every selectable instantiation has been
launched by THIS module, and nothing outside the selectable set ever ran.
Reference semantics: mobile_env.py:37-108 (ctor), :115-148 (reset), :150-194 (step), :196-233 (step_test / read_trace)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F32_RTOL = 1e-5
INT_KEYS = ("ue_xy", "bs_xy", "serving", "n_out", "step_n", "done")
F32_KEYS = ("cur_sinr", "mean_sinr", "reward")
F64_KEYS = ("cur_sinr_f64", "mean_sinr_f64", "reward_f64")

# (family, n_bs, n_ue, n_envs): packed needs n_ue <= 64 and n_ue >= n_bs; 7 envs at 20 UEs = 3 wavefronts, the last one ragged
SHAPES = [("packed", 4, 20, 7), ("packed", 8, 24, 5), ("packed", 16, 32, 5), ("packed", 32, 64, 3),
          ("multipass", 4, 72, 3), ("multipass", 8, 80, 3), ("multipass", 16, 72, 3), ("multipass", 32, 66, 3),
          # n_bs below the template bound: the checked kernels with a run-time UAV count
          ("packed", 3, 20, 7), ("packed", 6, 24, 5), ("packed", 12, 32, 5), ("packed", 20, 64, 3)]
G = 40


def _torch():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch


@pytest.fixture(scope="module", autouse=True)
def _fresh_census():
    _torch()
    from drl_uav_cellularnet_amd import _capi

    _capi.load().uavenv_debug_variant_reset()       # what the last test counts is what THIS module launched
    yield


def _lattice(B):
    side = int(np.ceil(np.sqrt(B)))
    return [(G // (2 * side) + (b // side) * (G // side), G // (2 * side) + (b % side) * (G // side)) for b in range(B)]


def _compare(got, want, what, f64):
    for k in INT_KEYS:
        np.testing.assert_array_equal(got[k], want[k], err_msg="%s: %s" % (what, k))
    for k in F32_KEYS:
        np.testing.assert_allclose(got[k], want[k], rtol=F32_RTOL, atol=0, err_msg="%s: %s" % (what, k))
    if f64:
        for k in F64_KEYS:
            np.testing.assert_allclose(got[k], want[k], rtol=1e-9, atol=1e-9, err_msg="%s: %s" % (what, k))


@pytest.mark.parametrize("plc", [True, False], ids=["cube", "generic"])
@pytest.mark.parametrize("variant", ["checked", "fast", "pin"])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "%s_B%d_U%d" % s[:3])
def test_every_mode_of_one_instantiation_family_matches_the_oracle(shape, variant, plc, monkeypatch):
    torch = _torch()
    from drl_uav_cellularnet_amd import BatchedMobiEnv
    from oracle import oracle as O

    fam, B, U, N = shape
    if fam == "multipass" and variant == "pin":
        pytest.skip("the multi-pass kernel has no pinned variant")
    if B not in (4, 8, 16, 32) and variant != "checked":
        pytest.skip("n_bs below the template bound always runs the checked kernels")
    monkeypatch.setenv("UAVENV_FORCE_PIN", "1" if variant == "pin" else "0")     # read once, in uavenv_create
    n_act = 5 if B <= 16 else 2                                                 # n_act^B must fit the int64 joint action
    over = {"n_act": n_act, "max_step": 12}
    if not plc:
        over["pl_b"] = 27.5                                                      # any exponent but 30: the exp2 / log path-loss form
    groups = [U // 4] * 3 + [U - 3 * (U // 4)]
    bs_init = None if B == 4 else _lattice(B)
    f64 = variant == "checked"                                                   # float64 copies requested -> the checked kernels
    env = BatchedMobiEnv(N, nBS=B, nUE=U, grid_n=G, groups=groups, bs_init=bs_init, seed=31337, env_id_base=11, f64_outputs=f64,
                         **over)
    ocfg = O.make_config(B, U, G, groups=groups, bs_init=bs_init, **over)
    orc = O.OracleEnv(ocfg, N, seed=31337, env_id_base=11)
    want = orc.construct()

    def got():
        torch.cuda.synchronize()
        return {k: v.cpu().numpy() for k, v in env.out.items()}

    _compare(got(), want, "constructor (init + 200 warm-up ticks + reset)", f64)
    rs = np.random.RandomState(B * 1000 + U)

    def actions(T):
        d = rs.randint(0, n_act, size=(T, N, B)).astype(np.int64)
        a = np.zeros((T, N), np.int64)
        for b in range(B):
            a = a * n_act + d[:, :, b]
        return a

    a = actions(3)
    for t in range(3):                                                           # MODE_STEP, FIFO depth 1 -> 3
        env.step(torch.as_tensor(a[t], device=env.device))
        _compare(got(), orc.step(a[t]), "step %d" % t, f64)
    a = actions(4)                                                               # uavenv_step_many: MANY kernel (packed) / host loop
    many = env.step_many(torch.as_tensor(a, device=env.device))
    torch.cuda.synchronize()
    for t in range(4):
        _compare({k: v[t].cpu().numpy() for k, v in many.items()}, orc.step(a[t]), "step_many block %d" % t, f64)
    mask = (np.arange(N) % 2 == 0).astype(np.uint8)                              # MODE_RESET on a subset
    env.reset(mask=mask)
    sel = mask.astype(bool)
    _compare({k: v[sel] for k, v in got().items()}, {k: v[sel] for k, v in orc.reset(mask=mask).items()}, "masked reset", f64)
    a = actions(6)                                                               # passes max_step = 12 on the envs not reset: done = 1
    for t in range(6):
        env.step(torch.as_tensor(a[t], device=env.device))
        _compare(got(), orc.step(a[t]), "step after reset %d" % t, f64)
    assert int(env.out["done"].max()) == 1 and int(env.out["done"].min()) == 0
    cells = rs.randint(0, G, size=(4, N, U, 2)).astype(np.int16)                 # MODE_RESET_TRACE / MODE_TRACE (read_trace)
    env.reset_trace(cells[0])
    _compare(got(), orc.reset_trace(cells[0]), "trace reset", f64)
    a = actions(3)
    for t in range(3):
        env.step_trace(torch.as_tensor(a[t], device=env.device), cells[1 + t])
        _compare(got(), orc.step_trace(a[t], cells[1 + t]), "trace step %d" % t, f64)
    s = env.state_fields()
    for k in ("ue_x", "ue_y", "g_x", "g_y", "g_fl", "g_v", "g_cos", "g_sin"):
        np.testing.assert_allclose(s[k], orc.s[k], rtol=0, atol=1e-9, err_msg=k)
    for k in ("agg", "deagg", "tick", "bs_xy", "serving", "fifo_depth", "out_bits", "step_n", "ue_xy"):
        np.testing.assert_array_equal(s[k], orc.s[k], err_msg=k)
    if fam == "packed":
        # The SCHED kernels (a multi-step call under a rotation schedule) of the same (n_bs bound, path loss, variant): a clone
        # whose handle plans its 3 env-wavefronts onto 2 slots, against this env's plain multi-step launches -- bit for bit (the plain
        # launches were compared with the oracle above).
        import ctypes as C

        monkeypatch.setenv("UAVENV_ROTATE", "1")
        monkeypatch.setenv("UAVENV_ROTATE_SLOTS", "2")
        env_r = env.clone()
        nl = C.c_int(-1)
        assert env_r._lib.uavenv_debug_rotation_info(env_r._h, 4, C.byref(nl), None) == 0 and nl.value == 1
        a = torch.as_tensor(actions(4), device=env.device)
        want, got_r = env.step_many(a), env_r.step_many(a)
        for k in want:
            assert torch.equal(got_r[k], want[k]), "scheduled step_many: %s" % k
        assert np.array_equal(env.get_state(), env_r.get_state()) and env_r.device_error() == 0


def test_every_selectable_instantiation_was_launched_and_nothing_else():
    _torch()
    from drl_uav_cellularnet_amd import _capi

    census = _capi.launch_census()
    selectable = [c for c in census if c[1]]
    assert len(selectable) == 188, len(selectable)      # 144 packed (incl. 24 SCHED) + 40 multi-pass + 4 warm-up (csrc/uavenv_capi.hip: variant_selectable)
    never = [name for name, sel, n in census if sel and n == 0]
    assert not never, "instantiations launch_env can select but no test of this module launched:\n  " + "\n  ".join(never)
    stray = [name for name, sel, n in census if not sel and n != 0]
    assert not stray, "launched although variant_selectable() excludes them:\n  " + "\n  ".join(stray)
