"""CPU: the C oracle against the golden vectors captured from the real reference
(tests/golden/make_golden.py).  Integers must be exact, float64 within 1e-9."""
import numpy as np

from oracle import oracle as O
from replay import make_checker, replay_fixture


def _cfg(fx):
    return O.make_config(fx["n_bs"], fx["n_ue"], fx["grid"], groups=list(fx["groups"]), bs_init=fx["bs_init"],
                         max_step=int(fx["max_step"]))


def test_oracle_replays_reference(golden):
    fx = golden
    env = O.OracleEnv(_cfg(fx), 1)
    U, B, Gr = fx["n_ue"], fx["n_bs"], fx["n_groups"]
    base = make_checker(fx, 1, f64_tol=1e-9)
    W = fx["warmup_ticks"]

    def check(e, kind, out, env):
        base(e, kind, out, env)
        s = env.s
        if kind == "ctor":
            ref = fx["mob_after_warmup"]
            np.testing.assert_allclose(s["ue_x"][0], ref[:U], rtol=0, atol=1e-9)
            np.testing.assert_allclose(s["ue_y"][0], ref[U:2 * U], rtol=0, atol=1e-9)
            o = 4 * U
            for i, k in enumerate(("g_x", "g_y", "g_fl", "g_v", "g_cos", "g_sin")):
                np.testing.assert_allclose(s[k][0], ref[o + i * Gr:o + (i + 1) * Gr], rtol=0, atol=1e-9, err_msg=k)
            assert s["agg"][0] == int(ref[o + 6 * Gr]) and s["deagg"][0] == int(ref[o + 6 * Gr + 1])
            np.testing.assert_array_equal(_bits(s["out_bits"][0], U), fx["init_out_mask"])
            return
        if e < fx["tick_pos"].shape[0]:
            np.testing.assert_allclose(np.stack([s["ue_x"][0], s["ue_y"][0]], 1), fx["tick_pos"][e], rtol=0,
                                       atol=1e-9)
        np.testing.assert_allclose(s["g_fl"][0], fx["tick_g_fl"][W + e], rtol=0, atol=1e-9)
        depth = int(fx["fifo_depth"][e])
        assert int(s["fifo_depth"][0]) == depth
        np.testing.assert_array_equal(s["fifo"][0][:depth], fx["fifo"][e][:depth])
        np.testing.assert_array_equal(_bits(s["out_bits"][0], U), fx["out_mask"][e])
        # dense observation: same non-zero cells as the reference's state tensor
        if e % 16 == 0 or e < 4:
            obs = env.obs_dense()[0]
            nz = np.argwhere(obs != 0)
            got = sorted((int(p), int(x), int(y), int(obs[p, x, y])) for p, x, y in nz)
            want = sorted(tuple(int(v) for v in r) for r in fx["state_nz"][e] if r[0] >= 0)
            assert got == want

    replay_fixture(env, fx, 1, check)
    ref = fx["mob_final"]
    np.testing.assert_allclose(env.s["ue_x"][0], ref[:U], rtol=0, atol=1e-8)


def _bits(words, U):
    return np.array([(int(words[u // 64]) >> (u % 64)) & 1 for u in range(U)], bool)


def test_np_pairwise_sum_matches_numpy():
    rs = np.random.RandomState(0)
    for n in (1, 3, 7, 8, 15, 20, 40, 128, 129, 200, 1000):
        for _ in range(50):
            a = rs.randn(n) * 10.0 ** rs.uniform(-3, 3, n)
            assert O.np_pairwise_sum(a) == float(np.sum(a))


def test_oracle_replays_reference_read_trace():
    """BASELINE config 1: MobiEnvironment(4, 40, 100, 'read_trace', ...) + step_test, captured from the reference."""
    import os

    from conftest import GOLDEN_DIR
    from fixture_io import load_fixture
    from replay import replay_trace_fixture

    fx = load_fixture(os.path.join(GOLDEN_DIR, "ref_trace_4x40_g100_seed6.npz"))
    fx["name"] = "ref_trace_4x40_g100_seed6"
    env = O.OracleEnv(_cfg(fx), 1)
    U = fx["n_ue"]
    base = make_checker(fx, 1, f64_tol=1e-9)
    seen = {"n": 0}

    def check(e, kind, out, env):
        base(e, kind, out, env)
        seen["n"] += 1
        if kind == "ctor":
            np.testing.assert_array_equal(_bits(env.s["out_bits"][0], U), fx["init_out_mask"])
            return
        depth = int(fx["fifo_depth"][e])
        assert int(env.s["fifo_depth"][0]) == depth
        np.testing.assert_array_equal(env.s["fifo"][0][:depth], fx["fifo"][e][:depth])
        np.testing.assert_array_equal(_bits(env.s["out_bits"][0], U), fx["out_mask"][e])
        if e % 16 == 0:
            obs = env.obs_dense()[0]
            nz = np.argwhere(obs != 0)
            got = sorted((int(p), int(x), int(y), int(obs[p, x, y])) for p, x, y in nz)
            want = sorted(tuple(int(v) for v in r) for r in fx["state_nz"][e] if r[0] >= 0)
            assert got == want

    replay_trace_fixture(env, fx, 1, check)
    assert seen["n"] == len(fx["ev_kind"]) + 1


import pytest


@pytest.mark.parametrize("name", ["ref_area_4bs_g100_seed7", "ref_area_5bs_g40_seed8"])
def test_oracle_sinr_area_matches_reference(name):
    """GetSinrInArea (channel.py:411-433): nearest-UAV SINR map with fresh fading, captured from the reference."""
    import os

    from conftest import GOLDEN_DIR

    with np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False) as z:
        seed, G, B, bs, want = int(z["seed"]), int(z["grid"]), int(z["n_bs"]), z["bs_xy"], z["sinr_area"]
    cfg = O.make_config(B, 8, G, groups=[2, 2, 2, 2], bs_init=bs)
    env = O.OracleEnv(cfg, 1)
    env.init()                                                  # puts the UAVs on bs_init
    fading = np.random.RandomState(seed).normal(0.0, 2.0, size=((G - 1) * (G - 1), B))
    got = env.sinr_area(fading=fading[None])[0]
    assert (got[0] == 0).all() and (got[:, 0] == 0).all()       # loops start at 1 (channel.py:416-417)
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-9)
