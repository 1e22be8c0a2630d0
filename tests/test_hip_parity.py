"""GPU parity tests: the HIP path, called through the C ABI (libuavenv.so), against
 (1) the golden vectors captured from the real reference (injected randomness), and
 (2) the CPU oracle on the same Philox streams (on-device randomness), and
 (3) size-independent properties at BASELINE.json's full sizes.
Bar: integers (UE/UAV cells, serving UAV, FIFO, outage set/count, step counter, done) bit-exact;
float32 SINR / mean / reward within 1e-5 relative (north_star); float64 copies within 1e-9."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F32_RTOL = 1e-5


def _torch():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch


def _make(n, fx=None, **kw):
    _torch()
    from drl_uav_cellularnet_amd import BatchedMobiEnv

    if fx is not None:
        kw.update(nBS=fx["n_bs"], nUE=fx["n_ue"], grid_n=fx["grid"], groups=list(fx["groups"]),
                  bs_init=[tuple(r) for r in fx["bs_init"]], max_step=int(fx["max_step"]))
    return BatchedMobiEnv(n, **kw)


def test_hip_replays_reference(golden):
    """Injected draws: every event of every fixture, 3 identical envs per fixture."""
    from hip_adapter import HipEnvAdapter
    from replay import make_checker, replay_fixture

    fx = golden
    N = 3
    env = _make(N, fx, f64_outputs=True, construct=False)
    ad = HipEnvAdapter(env)
    U, Gr, W = fx["n_ue"], fx["n_groups"], fx["warmup_ticks"]
    stats = {}
    base = make_checker(fx, N, f64_tol=1e-9, f32_rtol=F32_RTOL, stats=stats)
    long_fx = len(fx["ev_kind"]) > 400

    def check(e, kind, out, ad):
        base(e, kind, out, ad)
        if long_fx and e % 50 and e < len(fx["ev_kind"]) - 20:
            return  # state blob download is slow-ish; sample it on the 2000-event episode
        s = ad.s
        if kind == "ctor":
            ref = fx["mob_after_warmup"]
            for n in range(N):
                np.testing.assert_allclose(s["ue_x"][n], ref[:U], rtol=0, atol=1e-9)
                np.testing.assert_allclose(s["ue_y"][n], ref[U:2 * U], rtol=0, atol=1e-9)
                o = 4 * U
                for i, k in enumerate(("g_x", "g_y", "g_fl", "g_v", "g_cos", "g_sin")):
                    np.testing.assert_allclose(s[k][n], ref[o + i * Gr:o + (i + 1) * Gr], rtol=0, atol=1e-9, err_msg=k)
                assert s["agg"][n] == int(ref[o + 6 * Gr]) and s["deagg"][n] == int(ref[o + 6 * Gr + 1])
                np.testing.assert_array_equal(_bits(s["out_bits"][n], U), fx["init_out_mask"])
            return
        for n in range(N):
            if e < fx["tick_pos"].shape[0]:
                np.testing.assert_allclose(np.stack([s["ue_x"][n], s["ue_y"][n]], 1), fx["tick_pos"][e], rtol=0,
                                           atol=1e-9)
            np.testing.assert_allclose(s["g_fl"][n], fx["tick_g_fl"][W + e], rtol=0, atol=1e-9)
            depth = int(fx["fifo_depth"][e])
            assert int(s["fifo_depth"][n]) == depth
            np.testing.assert_array_equal(s["fifo"][n][:depth], fx["fifo"][e][:depth])
            np.testing.assert_array_equal(_bits(s["out_bits"][n], U), fx["out_mask"][e])
        if e % 16 == 0 or e < 4:  # dense observation == the reference's state tensor
            obs = env.dense_obs().cpu().numpy()
            for n in range(N):
                nz = np.argwhere(obs[n] != 0)
                got = sorted((int(p), int(x), int(y), int(obs[n][p, x, y])) for p, x, y in nz)
                want = sorted(tuple(int(v) for v in r) for r in fx["state_nz"][e] if r[0] >= 0)
                assert got == want

    replay_fixture(ad, fx, N, check)
    assert stats["events"] == len(fx["ev_kind"]) + 1


def test_hip_replays_reference_read_trace():
    """BASELINE config 1 on the HIP path: reset_trace / step_trace against the reference's step_test run."""
    import os

    from conftest import GOLDEN_DIR
    from fixture_io import load_fixture
    from hip_adapter import HipEnvAdapter
    from replay import make_checker, replay_trace_fixture

    fx = load_fixture(os.path.join(GOLDEN_DIR, "ref_trace_4x40_g100_seed6.npz"))
    fx["name"] = "ref_trace_4x40_g100_seed6"
    N = 2
    env = _make(N, fx, f64_outputs=True, construct=False)
    ad = HipEnvAdapter(env)
    U = fx["n_ue"]
    base = make_checker(fx, N, f64_tol=1e-9, f32_rtol=F32_RTOL)
    seen = {"n": 0}

    def check(e, kind, out, ad):
        base(e, kind, out, ad)
        seen["n"] += 1
        if kind == "ctor" or e % 8:
            return
        s = ad.s
        for n in range(N):
            depth = int(fx["fifo_depth"][e])
            assert int(s["fifo_depth"][n]) == depth
            np.testing.assert_array_equal(s["fifo"][n][:depth], fx["fifo"][e][:depth])
            np.testing.assert_array_equal(_bits(s["out_bits"][n], U), fx["out_mask"][e])

    replay_trace_fixture(ad, fx, N, check)
    assert seen["n"] == len(fx["ev_kind"]) + 1


def _bits(words, U):
    return np.array([(int(words[u // 64]) >> (u % 64)) & 1 for u in range(U)], bool)


# (B, U, G, N, T, config overrides).  Every env has its own Philox stream, so a lane / slot mix-up in the packed
# kernel cannot hide behind identical neighbours.  Kernel variant each row exercises:
PHILOX_SHAPES = [
    (4, 20, 100, 64, 96, {}),                  # packed, 3 envs per wavefront, N not a multiple of 3 (tail slot)
    (4, 20, 100, 65, 24, {}),                  # packed, last wavefront holds 2 of 3 slots
    (4, 40, 200, 16, 40, {}),                  # packed, 1 env per wavefront, class-default grid
    (2, 8, 32, 37, 40, {}),                    # packed, 8 envs per wavefront (kMaxEpw), 2 UAVs
    (7, 33, 64, 8, 24, {}),                    # packed, 1 env per wavefront, BT = 8 with B = 7, ragged groups
    (16, 200, 100, 8, 24, {}),                 # multi-pass (4 passes), BT = 16, 64-bit action digits
    (16, 12, 64, 8, 24, {}),                   # multi-pass because U < B (owner lanes would not fit a slot)
    (4, 64, 100, 5, 24, {}),                   # packed, U = 64 exactly: full-wavefront slot mask
    (4, 65, 100, 5, 24, {}),                   # multi-pass with ONE walker in the second pass
    (4, 20, 100, 1, 24, {}),                   # a single env (one live slot of three)
    (1, 8, 32, 9, 24, {}),                     # one UAV: HB = 1 (32-bit heading word), no interference terms
    (3, 20, 64, 12, 24, {}),                   # B = 3 < BT = 4: checked variant (FAST needs B == BT)
    (4, 20, 100, 32, 40, {"pl_b": 37.6, "pl_a": 15.3}),  # generic path-loss exponent (PLC = false), packed
    (16, 200, 100, 4, 12, {"pl_b": 37.6}),     # generic path-loss exponent, multi-pass
    # n_act = 9: digits 5..8 are the double steps of BS_move (ue_mobility.py:238-253); mobile_env.py only ever uses N_ACT = 5
    (4, 20, 100, 33, 40, {"n_act": 9}),        # packed FAST kernel, all nine digits
    (4, 20, 16, 33, 80, {"n_act": 9}),         # 16 x 16 grid: walls and the min-distance rule fire on most steps
    (4, 20, 16, 33, 80, {"n_act": 9, "bs_init": [(1, 1), (1, 15), (15, 1), (8, 8)]}),  # UAVs start ON the boundary: only the
                                               # MOVED coordinate is range-checked (:221-253), so they may slide along a wall
    (7, 33, 64, 8, 40, {"n_act": 9}),          # checked variant (B = 7 < BT = 8)
    (16, 200, 100, 4, 12, {"n_act": 9}),       # multi-pass, 9^16 needs the 64-bit digit path
    (4, 20, 100, 6, 2100, {}),                 # a whole episode and beyond: `done` turns on at step 2000 and stays on (no auto-reset),
                                               # ~20 aggregation phase changes, hundreds of group arrivals, masked reset at step 1050
    # the extremes check_config() allows
    (2, 4096, 100, 2, 3, {}),                  # n_ue at its maximum: 64 passes, 64-word outage mask per env
    (4, 1000, 100, 3, 4, {}),                  # 16 passes, last one with 40 of 64 lanes
    (1, 1, 16, 5, 30, {}),                     # ONE walker: groups [0, 0, 0, 1] (three empty groups), multi-pass because U < Gr
    (4, 20, 8, 7, 40, {}),                     # the smallest grid (8 x 8): UAVs 4 cells apart, frozen by the distance rule
    (4, 4, 32, 40, 40, {}),                    # U == B == Gr: every lane of a slot is walker, group owner and UAV owner; 8 envs per wave (kMaxEpw caps 16)
    (8, 64, 100, 3, 12, {}),                   # packed, BT = 8 with B = 8 (FAST), full-wavefront slots
    # quad draws (B > 8: one Philox call per four UAVs) away from the 16 x 200 shape, and BT = 32
    (9, 20, 64, 6, 12, {}),                    # packed, BT = 16 checked (B = 9): the third call serves ONE UAV
    (10, 70, 64, 4, 8, {}),                    # multi-pass, HB = 5 (not a power of two: plain tail pass), last call half used
    (20, 200, 100, 2, 6, {}),                  # multi-pass, B = 20 in the BT-agnostic checked variant, HB = 10, 5^20 actions
    (32, 64, 100, 3, 6, {"n_act": 2}),         # packed, BT = 32 FAST (2^32 joint actions), full-wavefront slots
    (32, 200, 100, 2, 4, {"n_act": 2}),        # multi-pass FAST, HB = 16: 8 tail walkers x 16 lanes > 64 -> plain tail
    (32, 66, 100, 3, 6, {"n_act": 2}),         # multi-pass FAST, 2 tail walkers x 16 lanes: item-layout tail with 16-lane groups
]


@pytest.mark.parametrize("shape", PHILOX_SHAPES, ids=lambda s: "B%dU%dG%dN%d%s" % (s[0], s[1], s[2], s[3], "".join("_" + k for k in sorted(s[5]))))
def test_hip_matches_oracle_on_philox_streams(shape):
    """No injection: device Philox/Box-Muller vs the oracle's, construct + reset + steps + masked reset."""
    torch = _torch()
    from oracle import oracle as O

    B, U, G, N, T, over = shape
    over = dict(over)
    n_act = over.get("n_act", 5)
    groups = None
    if U % 4:
        groups = [U // 4] * 3 + [U - 3 * (U // 4)]
    bs_init = None
    if B != 4:
        side = max(1, int(np.ceil(np.sqrt(B))))
        bs_init = [(G // (2 * side) + (b // side) * (G // side), G // (2 * side) + (b % side) * (G // side))
                   for b in range(B)]
    bs_init = over.pop("bs_init", bs_init)
    seed, base = 0xC0FFEE1234, 1000
    env = _make(N, nBS=B, nUE=U, grid_n=G, groups=groups, bs_init=bs_init, seed=seed, env_id_base=base,
                f64_outputs=True, **over)
    ocfg = O.make_config(B, U, G, groups=groups if groups else [U // 4] * 4, bs_init=bs_init, **over)
    orc = O.OracleEnv(ocfg, N, seed=seed, env_id_base=base)
    oo = orc.construct()
    rs = np.random.RandomState(7)

    def compare(tag):
        torch.cuda.synchronize()
        g = {k: v.cpu().numpy() for k, v in env.out.items()}
        for k in ("ue_xy", "bs_xy", "serving", "step_n", "n_out", "done"):
            np.testing.assert_array_equal(g[k], oo[k], err_msg="%s %s" % (tag, k))
        for k in ("cur_sinr", "mean_sinr", "reward"):
            np.testing.assert_allclose(g[k], oo[k], rtol=F32_RTOL, atol=0, err_msg="%s %s" % (tag, k))
        for k in ("cur_sinr_f64", "mean_sinr_f64", "reward_f64"):
            np.testing.assert_allclose(g[k], oo[k], rtol=1e-9, atol=1e-9, err_msg="%s %s" % (tag, k))

    compare("ctor")
    for t in range(T):
        a = rs.randint(0, n_act, size=(N, B)).astype(np.int64)
        act = np.zeros(N, np.int64)
        for b in range(B):
            act = act * n_act + a[:, b]
        if t == T // 2:  # masked reset of every other env, as a caller would do on `done`
            mask = (np.arange(N) % 2).astype(np.uint8)
            env.reset(mask=mask)
            oo = orc.reset(mask=mask)
            torch.cuda.synchronize()
            g = {k: v.cpu().numpy() for k, v in env.out.items()}
            sel = mask.astype(bool)
            for k in ("ue_xy", "bs_xy", "serving", "step_n"):
                np.testing.assert_array_equal(g[k][sel], oo[k][sel], err_msg="masked reset " + k)
        env.step(torch.as_tensor(act, device=env.device))
        oo = orc.step(act)
        compare("step %d" % t)
    s = env.state_fields()
    for k in ("ue_x", "ue_y", "g_x", "g_y", "g_fl", "g_v", "g_cos", "g_sin"):
        np.testing.assert_allclose(s[k], orc.s[k], rtol=0, atol=1e-9, err_msg=k)
    for k in ("agg", "deagg", "tick", "bs_xy", "serving", "fifo_depth", "out_bits", "step_n", "ue_xy"):
        np.testing.assert_array_equal(s[k], orc.s[k], err_msg=k)
    np.testing.assert_array_equal(env.dense_obs().cpu().numpy(), orc.obs_dense())


def test_full_size_properties_4096_envs():
    """BASELINE config 2 (4096 envs, 4 UAV x 20 UE): invariants the reference guarantees."""
    torch = _torch()
    N, B, U, G = 4096, 4, 20, 100
    env = _make(N, nBS=B, nUE=U, grid_n=G)
    gen = torch.Generator(device="cpu").manual_seed(1234)
    bs0 = env.out["bs_xy"].clone()
    for t in range(40):
        a = torch.randint(0, 625, (N,), generator=gen).to(env.device)
        obs, reward, done, info = env.step(a)
        assert int(info["step_n"].min()) == t + 1 == int(info["step_n"].max())
        assert float(reward.min()) >= -1.0
        assert not bool(done.any())
        ue = obs["ue_xy"]
        assert int(ue.min()) >= 0 and int(ue.max()) <= G - 1
        bs = obs["bs_xy"]
        # ue_mobility.py:221-235: +2 only while x+2 < G, -2 only while x-2 > 1  =>  2 <= x <= G-1
        # (SURVEY.md section 4 says [2, G-2]; the reference's own guards allow G-1, e.g. 97+2=99 < 100)
        assert int(bs.min()) >= 2 and int(bs.max()) <= G - 1
        assert bool(((bs - bs0) % 2 == 0).all())                            # BS_STEP = 2 keeps parity
        assert int(obs["serving"].min()) >= 0 and int(obs["serving"].max()) < B
        assert int(info["n_out"].min()) >= 0 and int(info["n_out"].max()) <= U
        assert bool(torch.isfinite(info["cur_sinr"]).all())
    dense = env.dense_obs()
    assert torch.equal(dense[:, 0].sum(dim=(1, 2)), torch.full((N,), float(B), device=env.device))
    assert torch.equal(dense[:, 1:].sum(dim=(1, 2, 3)), torch.full((N,), float(U), device=env.device))
    # action 624 = "4444" is a no-op for every UAV (ue_mobility.py:237)
    before = env.out["bs_xy"].clone()
    env.step(torch.full((N,), 624, dtype=torch.int64, device=env.device))
    assert torch.equal(before, env.out["bs_xy"])
    # mean_sinr really is the mean of cur_sinr; reward formula (mobile_env.py:163-189)
    m = env.out["cur_sinr"].double().mean(dim=1)
    assert torch.allclose(m, env.out["mean_sinr"].double(), rtol=1e-5, atol=1e-5)
    r = torch.clamp(env.out["mean_sinr"].double() / 20 - env.out["n_out"].double() / U, min=-1.0)
    assert torch.allclose(r, env.out["reward"].double(), rtol=1e-5, atol=1e-6)


def test_sharding_and_determinism():
    """Env e of a big batch == env 0 of a batch created with env_id_base = e (what a rank's shard is);
    and two handles with the same seed produce identical trajectories."""
    torch = _torch()
    N = 512
    big = _make(N, nBS=4, nUE=20, grid_n=100, seed=99)
    again = _make(N, nBS=4, nUE=20, grid_n=100, seed=99)
    shard = _make(128, nBS=4, nUE=20, grid_n=100, seed=99, env_id_base=256)
    gen = torch.Generator(device="cpu").manual_seed(5)
    for t in range(12):
        a = torch.randint(0, 625, (N,), generator=gen).to(big.device)
        big.step(a)
        again.step(a)
        shard.step(a[256:384].contiguous())
        for k in ("ue_xy", "bs_xy", "serving", "cur_sinr", "reward", "n_out"):
            assert torch.equal(big.out[k], again.out[k]), k
            assert torch.equal(big.out[k][256:384], shard.out[k]), k
    other = _make(N, nBS=4, nUE=20, grid_n=100, seed=100)
    assert not torch.equal(other.out["ue_xy"], big.out["ue_xy"])


def test_done_and_state_roundtrip_and_clone():
    torch = _torch()
    N = 64
    env = _make(N, nBS=4, nUE=20, grid_n=100, max_step=5)
    a = torch.full((N,), 624, dtype=torch.int64, device=env.device)
    for t in range(5):
        _, _, done, info = env.step(a)
        assert bool(done.all()) == (t == 4)                                 # mobile_env.py:186-187
    env.reset()
    assert int(env.out["step_n"].max()) == 0
    blob = env.get_state()
    twin = env.clone()                                                      # gradient.py:15 deepcopy(env)
    outs = []
    for t in range(3):
        env.step(a)
        outs.append({k: v.clone() for k, v in env.out.items()})
    for t in range(3):
        twin.step(a)
        for k, v in twin.out.items():
            assert torch.equal(v, outs[t][k]), k
    env.set_state(blob)
    for t in range(3):
        env.step(a)
        for k, v in env.out.items():
            assert torch.equal(v, outs[t][k]), k


def test_bad_arguments_fail_loudly():
    torch = _torch()
    from drl_uav_cellularnet_amd import BatchedMobiEnv, UavEnvError

    with pytest.raises(UavEnvError):
        BatchedMobiEnv(8, nBS=40)
    with pytest.raises(ValueError):
        BatchedMobiEnv(8, nBS=4, nUE=20, groups=[5, 5, 5])
    env = BatchedMobiEnv(8)
    with pytest.raises(ValueError):
        env.step(torch.zeros(7, dtype=torch.int64, device=env.device))


@pytest.mark.parametrize("name", ["ref_area_4bs_g100_seed7", "ref_area_5bs_g40_seed8"])
def test_hip_sinr_area_matches_reference(name):
    """LTEChannel.GetSinrInArea (channel.py:411-433), injected draws, against the map the real reference produced."""
    import os

    torch = _torch()
    from conftest import GOLDEN_DIR

    with np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False) as z:
        seed, G, B, bs, want = int(z["seed"]), int(z["grid"]), int(z["n_bs"]), z["bs_xy"], z["sinr_area"]
    N = 3
    env = _make(N, nBS=B, nUE=8, grid_n=G, groups=[2, 2, 2, 2], bs_init=[tuple(r) for r in bs], construct=False)
    env.init()                                                       # UAVs on bs_init
    fading = np.random.RandomState(seed).normal(0.0, 2.0, size=((G - 1) * (G - 1), B))
    f = np.broadcast_to(fading, (N,) + fading.shape)
    got64 = env.sinr_area(fading=f, dtype=torch.float64).cpu().numpy()
    got32 = env.sinr_area(fading=f, dtype=torch.float32).cpu().numpy()
    for n in range(N):
        assert (got64[n][0] == 0).all() and (got64[n][:, 0] == 0).all()
        np.testing.assert_allclose(got64[n], want, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(got32[n], want.astype(np.float32), rtol=F32_RTOL, atol=0)


def test_hip_sinr_area_for_any_bs_loc_matches_reference():
    """GetSinrInArea(bsLoc) takes ANY bsLoc (channel.py:411): uavenv_sinr_area_at with the fixture's five UAV cells passed as an
    override on a handle whose own UAVs sit elsewhere reproduces the map the real reference produced for those cells, and leaves
    the handle's state alone."""
    import os

    torch = _torch()
    from conftest import GOLDEN_DIR

    with np.load(os.path.join(GOLDEN_DIR, "ref_area_5bs_g40_seed8.npz"), allow_pickle=False) as z:
        seed, G, B, bs, want = int(z["seed"]), int(z["grid"]), int(z["n_bs"]), z["bs_xy"], z["sinr_area"]
    N = 2
    elsewhere = [(3 + 7 * b, 36 - 7 * b) for b in range(B)]
    assert not any(tuple(r) in elsewhere for r in bs)
    env = _make(N, nBS=B, nUE=8, grid_n=G, groups=[2, 2, 2, 2], bs_init=elsewhere, construct=False)
    env.init()
    before = env.get_state()
    fading = np.random.RandomState(seed).normal(0.0, 2.0, size=((G - 1) * (G - 1), B))
    f = np.broadcast_to(fading, (N,) + fading.shape)
    cells = np.broadcast_to(np.asarray(bs, np.int32)[:, :2], (N, B, 2))
    got = env.sinr_area(fading=f, dtype=torch.float64, bs_xy=cells).cpu().numpy()
    own = env.sinr_area(fading=f, dtype=torch.float64).cpu().numpy()
    for n in range(N):
        np.testing.assert_allclose(got[n], want, rtol=1e-9, atol=1e-9)
    assert np.abs(own[0] - want).max() > 1.0                                 # the handle's own cells give another map
    assert np.array_equal(env.get_state(), before)
    with pytest.raises(ValueError):
        env.sinr_area(bs_xy=cells[:, :4])


def test_hip_sinr_area_matches_oracle_on_philox_streams():
    torch = _torch()
    from oracle import oracle as O

    N, B, U, G = 5, 4, 20, 64
    env = _make(N, nBS=B, nUE=U, grid_n=G, seed=4242, env_id_base=7)
    orc = O.OracleEnv(O.make_config(B, U, G, groups=[5, 5, 5, 5]), N, seed=4242, env_id_base=7)
    orc.construct()
    rs = np.random.RandomState(1)
    for t in range(6):                                               # move the UAVs around first
        a = rs.randint(0, 625, N).astype(np.int64)
        env.step(torch.as_tensor(a, device=env.device))
        orc.step(a)
    got = env.sinr_area(dtype=torch.float64).cpu().numpy()
    want = orc.sinr_area()
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-9)
    assert float(np.abs(want).max()) > 10.0


def test_dense_obs_incremental_update_equals_full_rewrite():
    """uavenv_obs_dense_update moves only the changed cells; after any number of steps / masked resets the buffer must
    equal a full rewrite (and therefore the reference's state tensor, which the full rewrite is pinned to)."""
    torch = _torch()
    N = 300
    env = _make(N, nBS=4, nUE=20, grid_n=100, max_step=7)
    buf = env.dense_obs()                                   # first use: full write, records the cells
    gen = torch.Generator(device="cpu").manual_seed(3)
    for t in range(20):
        a = torch.randint(0, 625, (N,), generator=gen).to(env.device)
        _, _, done, _ = env.step(a)
        if t % 6 == 5:
            env.reset(mask=(torch.arange(N, device=env.device) % 3 == 0))
        env.dense_obs_update(buf)
        assert torch.equal(buf, _full(env)), "step %d" % t
    assert float(buf.min()) >= 0.0 and float(buf[:, 0].sum()) == 4.0 * N and float(buf[:, 1:].sum()) == 20.0 * N


def _full(env):
    # A full rewrite for comparison must not go through `env` itself: the handle remembers the cells of its LAST obs call,
    # so writing another buffer would break the lineage of `buf` (documented in include/uavenv.h).  A clone has its own handle.
    twin = env.clone()
    return twin.dense_obs()


@pytest.mark.parametrize("shape", [(4, 20, 100, 200), (4, 40, 100, 64), (8, 24, 64, 48), (16, 32, 100, 24),
                                   (16, 200, 100, 6), (4, 200, 100, 6), (32, 64, 100, 6, 2), (32, 66, 100, 4, 2)],   # + multi-pass FAST, BT = 32
                         ids=lambda s: "B%dU%dN%d" % (s[0], s[1], s[3]))
@pytest.mark.parametrize("pin", ["1", "0"], ids=["pinned", "unpinned"])
def test_production_fast_variant_matches_oracle(shape, pin, monkeypatch):
    """The kernels a user actually runs (FAST: no injection, the nine standard float32/int outputs, B == BT), in both
    their pinned and unpinned builds, against the oracle on identical Philox streams.  (The other oracle comparisons ask
    for float64 copies and therefore go through the checked variant.)"""
    torch = _torch()
    from oracle import oracle as O

    monkeypatch.setenv("UAVENV_FORCE_PIN", pin)          # read ONCE, in uavenv_create: set before the env below is built
    B, U, G, N = shape[:4]
    n_act = shape[4] if len(shape) > 4 else 5                     # (5^B must fit int64: B = 32 runs with two actions per UAV)
    over = {} if n_act == 5 else {"n_act": n_act}
    groups = [U // 4] * 3 + [U - 3 * (U // 4)]
    side = int(np.ceil(np.sqrt(B)))
    bs_init = None if B == 4 else [(G // (2 * side) + (b // side) * (G // side), G // (2 * side) + (b % side) * (G // side))
                                   for b in range(B)]
    env = _make(N, nBS=B, nUE=U, grid_n=G, groups=groups, bs_init=bs_init, seed=2024, env_id_base=5, **over)   # f64_outputs=False -> FAST
    assert "cur_sinr_f64" not in env.out
    orc = O.OracleEnv(O.make_config(B, U, G, groups=groups, bs_init=bs_init, **over), N, seed=2024, env_id_base=5)
    oo = orc.construct()
    rs = np.random.RandomState(11)
    T = 48
    for t in range(T):
        digits = rs.randint(0, n_act, size=(N, B)).astype(np.int64)
        act = np.zeros(N, np.int64)
        for b in range(B):
            act = act * n_act + digits[:, b]
        if t == T // 2:
            mask = (np.arange(N) % 3 == 1).astype(np.uint8)
            env.reset(mask=mask)
            orc.reset(mask=mask)
        env.step(torch.as_tensor(act, device=env.device))
        oo = orc.step(act)
        torch.cuda.synchronize()
        g = {k: v.cpu().numpy() for k, v in env.out.items()}
        for k in ("ue_xy", "bs_xy", "serving", "step_n", "n_out", "done"):
            np.testing.assert_array_equal(g[k], oo[k], err_msg="step %d %s" % (t, k))
        for k in ("cur_sinr", "mean_sinr", "reward"):
            np.testing.assert_allclose(g[k], oo[k], rtol=F32_RTOL, atol=0, err_msg="step %d %s" % (t, k))
    s = env.state_fields()
    for k in ("ue_x", "ue_y", "g_x", "g_y", "g_fl", "g_v", "g_cos", "g_sin"):
        np.testing.assert_allclose(s[k], orc.s[k], rtol=0, atol=1e-9, err_msg=k)
    for k in ("agg", "deagg", "tick", "bs_xy", "serving", "fifo_depth", "out_bits", "step_n", "ue_xy"):
        np.testing.assert_array_equal(s[k], orc.s[k], err_msg=k)
