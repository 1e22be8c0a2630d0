"""GPU: the single-env MobiEnvironment shim is a drop-in at the REFERENCE'S OWN API level.

The shim is driven through its public reset()/step()/step_test() exactly as the reference's drivers do
(main.py:190-202, main_test.py:54-75, gradient.py:15-22), with the fixture's recorded draws injected through a
private hook, and every returned object is compared with what the real reference returned (golden fixtures)."""
import copy
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _need_gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _state_from_nz(fx, e):
    B, G = fx["n_bs"], fx["grid"]
    st = np.zeros((B + 1, G, G))
    for p, x, y, c in fx["state_nz"][e]:
        if p >= 0:
            st[p, x, y] = c
    return st


def _load(name):
    from conftest import GOLDEN_DIR
    from fixture_io import load_fixture, regenerate_draws

    fx = load_fixture(os.path.join(GOLDEN_DIR, name + ".npz"))
    fx["name"] = name
    return fx, regenerate_draws


def test_group_mode_matches_reference_api_level():
    _need_gpu()
    from drl_uav_cellularnet_amd import MobiEnvironment

    fx, regen = _load("ref_4x40_g100_seed1")
    d = regen(fx)
    W = fx["warmup_ticks"]
    cur = {"tick": 0, "chan": 0}

    def hook(kind):
        if kind == "init":
            return dict(u_x=d["init_u_x"][None], u_y=d["init_u_y"][None], u_th=d["init_u_th"][None],
                        u_g=d["init_u_g"][None])
        t = cur["tick"]
        cur["tick"] += 1
        out = dict(theta_u=d["tick_u_th"][t][None], group_u=d["tick_u_grp"][t][None])
        if kind != "warmup":
            out["fading"] = d["fading"][cur["chan"]][None]
            cur["chan"] += 1
        return out

    env = MobiEnvironment(4, 40, 100, _draw_hook=hook)
    assert cur["tick"] == W and cur["chan"] == 1
    assert env.action_space_dim == 625 and env.observation_space_dim == 5 * 100 * 100   # mobile_env.py:104-105
    assert env.state.shape == (5, 100, 100) and not env.state.any()                     # zeros until reset (:107)
    np.testing.assert_array_equal(env.ueLoc, fx["init_ue_loc"])
    np.testing.assert_array_equal(env.channel.current_BS, fx["init_serving"])
    np.testing.assert_allclose(env.channel.current_BS_sinr, fx["init_cur_sinr"], rtol=1e-9, atol=1e-9)
    assert env.bsLoc.shape == (4, 3) and (env.bsLoc[:, 2] == 10).all()                   # mobile_env.py:58

    for e in range(120):
        if fx["ev_kind"][e] == 0:
            s = env.reset()
        else:
            a = int(fx["ev_action"][e])
            arg = [a, np.int64(a), np.array([a])][e % 3]                                 # main_test.py:73-75 passes a (1,) array
            s, r, done, info = env.step(arg)
            assert isinstance(r, float) and isinstance(done, bool) and isinstance(info, list) and len(info) == 2
            np.testing.assert_allclose(r, fx["reward"][e], rtol=1e-9, atol=1e-9)
            assert done == bool(fx["done"][e])
            np.testing.assert_allclose(info[0][0], fx["mean_sinr"][e] / 20, rtol=1e-9, atol=1e-9)
            assert info[0][1] == -1.0 * fx["n_out"][e] / 40
            assert info[1] == fx["step_n"][e]
        assert s.dtype == np.float64 and s.shape == (5, 100, 100)
        np.testing.assert_array_equal(s, _state_from_nz(fx, e))
        assert s is not env.state                                                       # fresh copy (:148,194)
        np.testing.assert_array_equal(env.ueLoc, fx["ue_loc"][e])
        np.testing.assert_array_equal(env.bsLoc[:, :2], fx["bs_loc"][e])
        np.testing.assert_array_equal(env.channel.current_BS, fx["serving"][e])
        np.testing.assert_allclose(env.channel.current_BS_sinr, fx["cur_sinr"][e], rtol=1e-9, atol=1e-9)
        assert env.step_n == fx["step_n"][e]


def test_read_trace_step_test_matches_reference_api_level(tmp_path):
    _need_gpu()
    from drl_uav_cellularnet_amd import MobiEnvironment
    from fixture_io import regenerate_trace_fading

    fx, _ = _load("ref_trace_4x40_g100_seed6")
    fading = regenerate_trace_fading(fx)
    cur = {"chan": 0}

    def hook(kind):
        c = cur["chan"]
        cur["chan"] += 1
        return dict(fading=fading[c][None])

    path = os.path.join(tmp_path, "ue_trace.npy")
    np.save(path, fx["trace"])                                                          # main_test.py:51 loads a .npy
    env = MobiEnvironment(4, 40, 100, "read_trace", path, _draw_hook=hook)
    np.testing.assert_array_equal(env.ueLoc, fx["trace"][0])
    np.testing.assert_array_equal(env.channel.current_BS, fx["init_serving"])
    for e in range(len(fx["ev_kind"])):
        if fx["ev_kind"][e] == 0:
            s = env.reset()
        else:
            s, r, done, info = env.step_test(np.array([int(fx["ev_action"][e])]), False)
            np.testing.assert_allclose(r, fx["reward"][e], rtol=1e-9, atol=1e-9)
            assert info._fields == ("r_dissect", "step_n", "ue_loc", "bs_loc", "outage_fraction", "bs_actions")
            assert info.step_n == fx["step_n"][e]
            np.testing.assert_allclose(info.outage_fraction, fx["outage_fraction"][e], rtol=0, atol=1e-12)
            np.testing.assert_array_equal(info.bs_actions, fx["bs_actions"][e])
            np.testing.assert_array_equal(info.ue_loc, fx["ue_loc"][e])
            np.testing.assert_array_equal(info.bs_loc[:, :2], fx["bs_loc"][e])
        np.testing.assert_array_equal(s, _state_from_nz(fx, e))
        np.testing.assert_allclose(env.channel.current_BS_sinr, fx["cur_sinr"][e], rtol=1e-9, atol=1e-9)


def test_deepcopy_lookahead_like_gradient_py():
    """gradient.py:15-17: deepcopy the env, step the copy with 624 ('stay'), the original must be untouched."""
    _need_gpu()
    from drl_uav_cellularnet_amd import MobiEnvironment

    env = MobiEnvironment(4, 40, 100, seed=11)
    env.reset()
    for a in (3, 100, 624, 17):
        env.step(a)
    before = (env.ueLoc.copy(), env.bsLoc.copy(), env.channel.current_BS.copy(), env.step_n)
    virtual = copy.deepcopy(env)
    s_v, r_v, _, _ = virtual.step_test(624, False)
    np.testing.assert_array_equal(virtual.bsLoc, before[1])                              # 624 moves no UAV
    np.testing.assert_array_equal(env.ueLoc, before[0])
    assert env.step_n == before[3] and virtual.step_n == before[3] + 1
    s_e, r_e, _, _ = env.step_test(624, False)                                           # same stream => same future
    np.testing.assert_array_equal(s_e, s_v)
    assert r_e == r_v


def test_unsupported_models_behave_like_the_reference():
    _need_gpu()
    from drl_uav_cellularnet_amd import MobiEnvironment

    with pytest.raises(SystemExit):                                                      # mobile_env.py:90-91
        MobiEnvironment(4, 40, 100, "no_such_model")
    with pytest.raises(NotImplementedError):
        MobiEnvironment(4, 40, 100, "in_coverage")
    with pytest.raises(AssertionError):                                                  # mobile_env.py:84
        MobiEnvironment(4, 40, 100, "read_trace", "")


def test_gradient_heuristic_runs_on_the_shim_and_leaves_the_env_untouched():
    """gradient.py:14-37,56-86: look-ahead on a deepcopy + step_test; actions use digits 0..3 only."""
    _need_gpu()
    from drl_uav_cellularnet_amd import MobiEnvironment
    from drl_uav_cellularnet_amd.heuristics import choose_act_gradient, run_gradient_policy, side_means

    env = MobiEnvironment(4, 40, 100, seed=21)
    env.reset()
    snap = (env.ueLoc.copy(), env.bsLoc.copy(), env.channel.current_BS_sinr.copy(), env.step_n)
    a = choose_act_gradient(env)
    np.testing.assert_array_equal(env.ueLoc, snap[0])
    np.testing.assert_array_equal(env.bsLoc, snap[1])
    np.testing.assert_array_equal(env.channel.current_BS_sinr, snap[2])
    assert env.step_n == snap[3]
    digits = [(a // 5 ** k) % 5 for k in (3, 2, 1, 0)]
    assert all(0 <= d <= 3 for d in digits)
    # independent evaluation of the rule on the same look-ahead state
    import copy

    v = copy.deepcopy(env)
    v.step_test(624, False)
    want = [int(np.nanargmin(side_means(v.channel.current_BS_sinr, v.ueLoc, v.bsLoc[i]))) for i in range(4)]
    assert digits == want
    rewards, actions = run_gradient_policy(MobiEnvironment(4, 40, 100, seed=21), 40)
    assert rewards.shape == (40,) and np.isfinite(rewards).all() and (rewards >= -1).all()
    r2, a2 = run_gradient_policy(MobiEnvironment(4, 40, 100, seed=21), 40)
    np.testing.assert_array_equal(actions, a2)                          # deterministic given the seed
    np.testing.assert_array_equal(rewards, r2)


def test_shim_get_sinr_in_area_shape_and_determinism():
    """main_test.py:89: test_env.channel.GetSinrInArea(info.bs_loc) -> (G, G) float64 map."""
    _need_gpu()
    from drl_uav_cellularnet_amd import MobiEnvironment

    env = MobiEnvironment(4, 40, 100, seed=3)
    env.reset()
    _, _, _, info = env.step_test(17)
    m = env.channel.GetSinrInArea(info.bs_loc)
    assert m.shape == (100, 100) and m.dtype == np.float64
    assert (m[0] == 0).all() and (m[:, 0] == 0).all() and np.isfinite(m).all()
    bx, by = info.bs_loc[0][:2]
    assert m[bx, by] > 60.0                                           # a cell under a UAV: d = 0 -> loss 0
    moved = info.bs_loc + np.array([3, -2, 0])                        # ANY bsLoc, as channel.py:411 accepts
    m2 = env.channel.GetSinrInArea(moved)
    assert m2.shape == (100, 100) and m2[moved[0][0], moved[0][1]] > 60.0 and not np.array_equal(m2, m)
    assert np.array_equal(env.channel.GetSinrInArea(info.bs_loc), m)  # same tick, same cells: the same Philox draws
    assert np.array_equal(env.bsLoc, info.bs_loc)                     # the env's own UAVs did not move
    with pytest.raises(ValueError):
        env.channel.GetSinrInArea(info.bs_loc[:3])


def test_eval_harness_like_main_test_py(tmp_path):
    """BASELINE config 1 plumbing: synthesise a trace, replay it with step_test under a greedy actor, save what
    main_test.py saves (:46-113)."""
    _need_gpu()
    import importlib.util
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("run_eval", os.path.join(root, "tools", "run_eval.py"))
    run_eval = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(run_eval)
    trace = run_eval.make_trace(n_rows=64)
    assert trace.shape == (64, 40, 2) and trace.dtype == np.int16 and trace.min() >= 0 and trace.max() <= 99
    assert (np.abs(np.diff(trace.astype(int), axis=0)) <= 3).all()          # <= 2.8 cells per tick (1 + 1 + 0.8)
    out = os.path.join(tmp_path, "eval")
    res = run_eval.run_test(trace, out, max_step=50, area_every=25)
    assert len(res["reward"]) == 51                                         # `while step <= MAX_STEP` (:69)
    assert res["sinr"].shape == (51, 40) and res["ue_location"].shape == (51, 40, 2) and res["bs_location"].shape == (51, 4, 3)
    assert res["sinr_area"].shape == (3, 100, 100)                           # steps 0, 25, 50
    np.testing.assert_array_equal(res["ue_location"], trace[:51])            # step_test replays trace[step_n] (:202-203)
    for name in ("reward", "decomposed_reward", "sinr", "time", "outage_fraction", "ue_location", "bs_location",
                 "action", "sinr_area"):
        assert os.path.isfile(os.path.join(out, name + ".npy"))
    res2 = run_eval.run_test(trace, os.path.join(tmp_path, "eval2"), max_step=50, area_every=25)
    np.testing.assert_array_equal(res["reward"], res2["reward"])             # greedy + seeded env: reproducible


def test_deepcopy_of_a_read_trace_env_like_gradient_py():
    """gradient.py:42,61 builds a read_trace env and Choose_Act_Gradient deep-copies it every step (:15)."""
    _need_gpu()
    from drl_uav_cellularnet_amd import MobiEnvironment
    from drl_uav_cellularnet_amd.heuristics import choose_act_gradient

    fx, _ = _load("ref_trace_4x40_g100_seed6")
    env = MobiEnvironment(4, 40, 100, "read_trace", fx["trace"], seed=9)     # in-memory trace
    env.reset()
    for _ in range(3):
        env.step_test(choose_act_gradient(env), False)                        # deep-copies the env inside
    twin = copy.deepcopy(env)
    assert twin.mobility_model == "read_trace" and twin.step_n == env.step_n == 3
    s1, r1, _, i1 = env.step_test(100, False)
    s2, r2, _, i2 = twin.step_test(100, False)
    np.testing.assert_array_equal(s1, s2)
    assert r1 == r2
    np.testing.assert_array_equal(i1.ue_loc, fx["trace"][3])
