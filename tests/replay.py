"""Replays a golden fixture through any backend that offers the OracleEnv call surface
(init / warmup / reset / step with injected draws) and checks every recorded output.

TEST INFRASTRUCTURE shared by the oracle tests (CPU) and the HIP parity tests (GPU)."""
import numpy as np


def tile(a, n):
    a = np.asarray(a)
    return np.broadcast_to(a, (n,) + a.shape).copy()


def replay_fixture(env, fx, n_envs=1, check=None, max_events=None, float_out=True):
    """Drive ``env`` (N identical copies of the fixture's env) through ctor + all events.

    ``check(e, kind, out_dict, env)`` is called after every event with numpy outputs."""
    d = fx["draws"]
    W = fx["warmup_ticks"]
    N = n_envs
    env.init(u_x=tile(d["init_u_x"], N), u_y=tile(d["init_u_y"], N), u_th=tile(d["init_u_th"], N),
             u_g=tile(d["init_u_g"], N))
    for t in range(W - 1):
        env.warmup(theta_u=tile(d["tick_u_th"][t], N), group_u=tile(d["tick_u_grp"][t], N))
    # last constructor tick + LTEChannel.__init__ == reset semantics (mobile_env.py:94-98, channel.py:92-93,110)
    out = env.reset(theta_u=tile(d["tick_u_th"][W - 1], N), group_u=tile(d["tick_u_grp"][W - 1], N),
                    fading=tile(d["fading"][0], N))
    if check:
        check(-1, "ctor", out, env)
    E = len(fx["ev_kind"]) if max_events is None else min(max_events, len(fx["ev_kind"]))
    for e in range(E):
        inj = dict(theta_u=tile(d["tick_u_th"][W + e], N), group_u=tile(d["tick_u_grp"][W + e], N),
                   fading=tile(d["fading"][e + 1], N))
        if fx["ev_kind"][e] == 0:
            out = env.reset(**inj)
            kind = "reset"
        else:
            out = env.step(np.full(N, int(fx["ev_action"][e]), np.int64), **inj)
            kind = "step"
        if check:
            check(e, kind, out, env)
    return env


def make_checker(fx, n_envs, f64_tol=1e-9, f32_rtol=1e-5, has_f64=True, stats=None):
    """Standard per-event assertions: ints exact, float64 outputs <= f64_tol (abs+rel),
    float32 outputs within f32_rtol relative (BASELINE.json north_star: 1e-5)."""
    U, B = fx["n_ue"], fx["n_bs"]

    def every(a):
        a = np.asarray(a)
        assert (a == a[0:1]).all(), "identical envs diverged"
        return a[0]

    def check(e, kind, out, env):
        tag = "%s event %d (%s)" % (fx["name"], e, kind)
        if kind == "ctor":
            np.testing.assert_array_equal(every(out["ue_xy"]), fx["init_ue_loc"], err_msg=tag)
            np.testing.assert_array_equal(every(out["serving"]), fx["init_serving"], err_msg=tag)
            ref_s = fx["init_cur_sinr"]
        else:
            np.testing.assert_array_equal(every(out["ue_xy"]), fx["ue_loc"][e], err_msg=tag)
            np.testing.assert_array_equal(every(out["bs_xy"]), fx["bs_loc"][e], err_msg=tag)
            np.testing.assert_array_equal(every(out["serving"]), fx["serving"][e], err_msg=tag)
            np.testing.assert_array_equal(every(out["step_n"]), fx["step_n"][e], err_msg=tag)
            ref_s = fx["cur_sinr"][e]
        if has_f64:
            np.testing.assert_allclose(every(out["cur_sinr_f64"]), ref_s, rtol=f64_tol, atol=f64_tol, err_msg=tag)
        np.testing.assert_allclose(every(out["cur_sinr"]), ref_s.astype(np.float32), rtol=f32_rtol, atol=0,
                                   err_msg=tag)
        if kind == "step":
            assert int(every(out["n_out"])) == int(fx["n_out"][e]), tag
            assert bool(every(out["done"])) == bool(fx["done"][e]), tag
            if has_f64:
                np.testing.assert_allclose(every(out["mean_sinr_f64"]), fx["mean_sinr"][e], rtol=f64_tol,
                                           atol=f64_tol, err_msg=tag)
                np.testing.assert_allclose(every(out["reward_f64"]), fx["reward"][e], rtol=f64_tol, atol=f64_tol,
                                           err_msg=tag)
            np.testing.assert_allclose(every(out["mean_sinr"]), np.float32(fx["mean_sinr"][e]), rtol=f32_rtol,
                                       err_msg=tag)
            np.testing.assert_allclose(every(out["reward"]), np.float32(fx["reward"][e]), rtol=f32_rtol, err_msg=tag)
        if stats is not None:
            stats["events"] = stats.get("events", 0) + 1

    return check


def replay_trace_fixture(env, fx, n_envs=1, check=None):
    """read_trace fixtures (BASELINE config 1): constructor = reset_trace(trace[0]); 'reset' events the same;
    'step' events = step_test with UE cells trace[row] (mobile_env.py:85-89,128-131,202-203)."""
    from fixture_io import regenerate_trace_fading

    N = n_envs
    fading = regenerate_trace_fading(fx)
    trace = fx["trace"]
    env.init()  # mobility state is unused in read_trace mode but must be defined
    out = env.reset_trace(tile(trace[0], N), fading=tile(fading[0], N))
    if check:
        check(-1, "ctor", out, env)
    for e in range(len(fx["ev_kind"])):
        f = tile(fading[e + 1], N)
        if fx["ev_kind"][e] == 0:
            out = env.reset_trace(tile(trace[0], N), fading=f)
            kind = "reset"
        else:
            row = int(fx["ev_trace_row"][e])
            out = env.step_trace(np.full(N, int(fx["ev_action"][e]), np.int64), tile(trace[row], N), fading=f)
            kind = "step"
        if check:
            check(e, kind, out, env)
    return env
