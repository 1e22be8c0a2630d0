#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the real reference.

TEST INFRASTRUCTURE.  Runs ONLY in the build container (needs /root/reference;
see ref_loader.py).  Usage:  python tests/golden/make_golden.py [name ...]

Each scenario seeds the legacy global NumPy stream, drives the unmodified
reference classes, records every random draw through two thin recording
proxies (``ue_mobility.rand`` bound at ue_mobility.py:6, ``numpy.random.normal``
looked up at channel.py:240), and stores seed + arrival masks + actions + all
outputs.  It then asserts that ``fixture_io.regenerate_draws`` reproduces the
recorded draws bit for bit, so the fixtures need not carry the draws.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_loader import load_reference  # noqa: E402
from fixture_io import regenerate_draws  # noqa: E402

MOB_KEYS = ("x", "y", "costheta", "sintheta", "g_x", "g_y", "g_fl", "g_velocity",
            "g_costheta", "g_sintheta", "aggregating", "deaggregating")


class Recorder:
    """Recording proxies around the reference's two draw sites."""

    def __init__(self, mods):
        self.um = mods["ue_mobility"]
        self.rand_log = []
        self.normal_log = []
        self._orig_rand = self.um.rand
        self._orig_normal = np.random.normal

    def __enter__(self):
        def rand(*shape):
            v = self._orig_rand(*shape)
            self.rand_log.append(np.array(v, dtype=np.float64).ravel().copy())
            return v

        def normal(*a, **k):
            v = self._orig_normal(*a, **k)
            self.normal_log.append(float(v))
            return v

        self.um.rand = rand
        np.random.normal = normal
        return self

    def __exit__(self, *exc):
        self.um.rand = self._orig_rand
        np.random.normal = self._orig_normal


class MobilityProbe:
    """Wraps reference_point_group so the generator's locals can be snapshotted per tick."""

    def __init__(self, mods, groups):
        self.real_rpg = mods["ue_mobility"].reference_point_group
        self.groups = list(groups)
        self.gen = None
        self.snaps = []

    def factory(self, nr_nodes, dimensions, velocity=(0.1, 1.0), aggregation=0.1):
        # mobile_env.py:76 passes [10,10,10,10]; the group list is overridden here for
        # the 20-UE / 200-UE shapes (SURVEY.md section 8 notes N1, N2)
        self.gen = self.real_rpg(self.groups, dimensions=dimensions, velocity=velocity,
                                 aggregation=aggregation)

        def proxy():
            while True:
                pos = next(self.gen)
                loc = self.gen.gi_frame.f_locals
                self.snaps.append({k: np.array(loc[k], dtype=np.float64).copy() for k in MOB_KEYS})
                yield pos

        return proxy()


def state_nonzeros(state, n_rows):
    nz = np.argwhere(state != 0)
    rows = np.full((n_rows, 4), -1, dtype=np.int16)
    assert nz.shape[0] <= n_rows
    for i, (p, x, y) in enumerate(nz):
        rows[i] = (p, x, y, int(state[p, x, y]))
    return rows


def channel_snapshot(ch, U):
    buf = np.asarray(ch.bestBS_buf)
    if buf.ndim == 1:
        buf = buf[None, :]
    depth = buf.shape[0]
    fifo = np.full((3, U), -1, dtype=np.int8)
    fifo[:depth] = buf
    out_mask = np.zeros(U, dtype=bool)
    out_mask[np.asarray(ch.ue_out).ravel().astype(int)] = True
    return (np.array(ch.current_BS, dtype=np.int8), np.array(ch.current_BS_sinr, dtype=np.float64),
            fifo, depth, out_mask)


def finish_fixture(name, meta, rec, probe, ev, extra):
    """Parse the draw logs, cross-check against regenerate_draws, save."""
    U, B, Gr, W = meta["n_walkers"], meta["n_bs"], meta["n_groups"], meta["warmup_ticks"]
    log = rec.rand_log
    assert [a.size for a in log[:8]] == [U, U, U, Gr, Gr, Gr, Gr, Gr], [a.size for a in log[:8]]
    T = len(probe.snaps)
    rec_th = np.zeros((T, U))
    rec_grp = np.zeros((T, Gr, 3))
    arrived = np.zeros((T, Gr), dtype=bool)
    pos = 8
    prev_v = log[6].copy()  # g_velocity = U(0,1) draw (ue_mobility.py:445)
    for t in range(T):
        rec_th[t] = log[pos]
        pos += 1
        v = probe.snaps[t]["g_velocity"]
        idx = np.flatnonzero(v != prev_v)
        if idx.size:
            for k in range(3):
                assert log[pos].size == idx.size, (t, log[pos].size, idx)
                rec_grp[t, idx, k] = log[pos]
                pos += 1
            arrived[t, idx] = True
        prev_v = v.copy()
    assert pos == len(log), (pos, len(log))
    E = T - W
    rows = meta["n_ue_channel_rows"]
    normals = np.array(rec.normal_log)
    assert normals.size == (E + 1) * rows * B, (normals.size, E, rows, B)
    rec_fading = normals.reshape(E + 1, rows, B)

    fx = dict(meta)
    fx["tick_arrived"] = arrived
    fx.update(ev)
    fx.update(extra)
    fx["mob_after_warmup"] = np.concatenate([np.ravel(probe.snaps[W - 1][k]) for k in MOB_KEYS])
    fx["mob_final"] = np.concatenate([np.ravel(probe.snaps[-1][k]) for k in MOB_KEYS])
    # float64 walker positions after each post-constructor tick (first 256 events only: size)
    fx["tick_pos"] = np.stack([np.stack([s["x"], s["y"]], axis=1) for s in probe.snaps[W:W + 256]])
    fx["tick_g_fl"] = np.stack([s["g_fl"] for s in probe.snaps])

    regen = regenerate_draws(fx)
    assert np.array_equal(regen["init_u_x"], log[0]) and np.array_equal(regen["init_u_y"], log[1])
    assert np.array_equal(regen["init_u_th"], log[2])
    assert np.array_equal(regen["init_u_g"], np.stack(log[3:8]))
    assert np.array_equal(regen["tick_u_th"], rec_th)
    assert np.array_equal(regen["tick_u_grp"], rec_grp)
    assert np.array_equal(regen["fading"], rec_fading), "vector normal draw != scalar draws"

    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in fx.items()})
    d0 = int(extra.get("n_zero_distance", 0))
    print("%-34s T=%d E=%d arrivals=%d handovers=%d zero-dist pairs=%d  %.1f KB" % (
        name, T, E, int(arrived.sum()), int(extra.get("n_handovers", 0)), d0,
        os.path.getsize(path) / 1024.0))


def run_env_scenario(name, seed, n_ue, grid, groups, script):
    """Drive the reference's own MobiEnvironment.  ``script`` = list of ('reset',) / ('step', a)."""
    mods = load_reference()
    me = mods["mobile_env"]
    probe = MobilityProbe(mods, groups)
    orig = me.reference_point_group
    me.reference_point_group = probe.factory
    B, U = 4, n_ue
    assert sum(groups) == U
    np.random.seed(seed)
    try:
        with Recorder(mods) as rec:
            env = me.MobiEnvironment(B, U, grid)
            W = len(probe.snaps)
            c0 = channel_snapshot(env.channel, U)
            init_ue = np.array(env.ueLoc[:, :2], dtype=np.int16)
            E = len(script)
            ev = {
                "ev_kind": np.zeros(E, np.int8), "ev_action": np.zeros(E, np.int64),
                "ue_loc": np.zeros((E, U, 2), np.int16), "bs_loc": np.zeros((E, B, 2), np.int16),
                "serving": np.zeros((E, U), np.int8), "cur_sinr": np.zeros((E, U)),
                "fifo": np.zeros((E, 3, U), np.int8), "fifo_depth": np.zeros(E, np.int8),
                "out_mask": np.zeros((E, U), bool), "mean_sinr": np.full(E, np.nan),
                "n_out": np.zeros(E, np.int32), "reward": np.full(E, np.nan),
                "done": np.zeros(E, bool), "step_n": np.zeros(E, np.int32),
                "state_nz": np.zeros((E, U + B, 4), np.int16),
            }
            n_ho = 0
            n_d0 = 0
            for e, item in enumerate(script):
                prev_serv = np.array(env.channel.current_BS)
                if item[0] == "reset":
                    state = env.reset()
                    ev["ev_kind"][e] = 0
                else:
                    a = int(item[1])
                    state, reward, done, info = env.step(a)
                    ev["ev_kind"][e] = 1
                    ev["ev_action"][e] = a
                    ev["mean_sinr"][e] = info[0][0] * 20.0  # kept only as a cross-check
                    ev["n_out"][e] = int(round(-info[0][1] * U))
                    ev["reward"][e] = reward
                    ev["done"][e] = done
                    n_ho += int(np.sum(prev_serv != env.channel.current_BS))
                serv, sinr, fifo, depth, omask = channel_snapshot(env.channel, U)
                ev["ue_loc"][e] = env.ueLoc[:, :2]
                ev["bs_loc"][e] = env.bsLoc[:, :2]
                ev["serving"][e], ev["cur_sinr"][e] = serv, sinr
                ev["fifo"][e], ev["fifo_depth"][e], ev["out_mask"][e] = fifo, depth, omask
                ev["step_n"][e] = env.step_n
                ev["state_nz"][e] = state_nonzeros(state, U + B)
                if item[0] == "step":
                    # exact mean as the reference computed it (channel.py:216)
                    ev["mean_sinr"][e] = float(np.mean(env.channel.current_BS_sinr))
                d = env.ueLoc[:, None, :2] - env.bsLoc[None, :, :2]
                n_d0 += int(np.sum((d ** 2).sum(-1) == 0))
    finally:
        me.reference_point_group = orig
    meta = {"seed": seed, "n_walkers": U, "n_ue_channel_rows": U, "n_bs": B, "n_ue": U,
            "n_groups": len(groups), "groups": np.array(groups, np.int32), "grid": grid,
            "warmup_ticks": W, "max_step": int(me.MAXSTEP),
            "bs_init": np.array(env.initBsLoc[:, :2], np.int16)}
    extra = {"init_ue_loc": init_ue, "init_serving": c0[0], "init_cur_sinr": c0[1],
             "init_out_mask": c0[4], "n_handovers": n_ho, "n_zero_distance": n_d0}
    finish_fixture(name, meta, rec, probe, ev, extra)


def run_parts_scenario(name, seed, n_bs, groups, grid, bs_init, actions):
    """Shapes the reference ctor cannot build: drive LTEChannel, reference_point_group
    and BS_move directly, in the order MobiEnvironment.__init__/step use them
    (mobile_env.py:76-98,150-189).  The glue below is this script's, the arithmetic the reference's."""
    mods = load_reference()
    um, chm = mods["ue_mobility"], mods["channel"]
    U, B = sum(groups), n_bs
    probe = MobilityProbe(mods, groups)
    np.random.seed(seed)
    with Recorder(mods) as rec:
        mm = probe.factory(groups, dimensions=(grid, grid), velocity=(0, 1), aggregation=0.8)
        for _ in range(200):
            next(mm)
        ue = next(mm).astype(int)
        W = len(probe.snaps)
        bs = np.concatenate([np.array(bs_init, dtype=int), np.full((B, 1), 10, dtype=int)], axis=1)
        ch = chm.LTEChannel(U, B, [1, grid, 1, grid], ue, bs)
        c0 = channel_snapshot(ch, U)
        E = len(actions)
        ev = {
            "ev_kind": np.ones(E, np.int8), "ev_action": np.array(actions, np.int64),
            "ue_loc": np.zeros((E, U, 2), np.int16), "bs_loc": np.zeros((E, B, 2), np.int16),
            "serving": np.zeros((E, U), np.int8), "cur_sinr": np.zeros((E, U)),
            "fifo": np.zeros((E, 3, U), np.int8), "fifo_depth": np.zeros(E, np.int8),
            "out_mask": np.zeros((E, U), bool), "mean_sinr": np.full(E, np.nan),
            "n_out": np.zeros(E, np.int32), "reward": np.full(E, np.nan),
            "done": np.zeros(E, bool), "step_n": np.zeros(E, np.int32),
            "state_nz": np.full((E, U + B, 4), -1, np.int16),
        }
        n_ho = 0
        n_d0 = 0
        for e, a in enumerate(actions):
            prev_serv = np.array(ch.current_BS)
            ue = next(mm).astype(int)
            bs, _ = um.BS_move(bs, [1, grid, 1, grid], int(a), 2, 4, 5)
            assoc, mean_sinr, n_out = ch.UpdateDroneNet(ue, bs, False, e)
            # state planes as mobile_env.py:160,169-170 builds them
            state = np.concatenate([um.GetGridMap(grid, grid, bs)[None], assoc], axis=0)
            ev["state_nz"][e] = state_nonzeros(state, U + B)
            serv, sinr, fifo, depth, omask = channel_snapshot(ch, U)
            ev["ue_loc"][e], ev["bs_loc"][e] = ue[:, :2], bs[:, :2]
            ev["serving"][e], ev["cur_sinr"][e] = serv, sinr
            ev["fifo"][e], ev["fifo_depth"][e], ev["out_mask"][e] = fifo, depth, omask
            ev["mean_sinr"][e], ev["n_out"][e] = mean_sinr, n_out
            ev["reward"][e] = max(mean_sinr / 20 + -1.0 * n_out / U, -1)
            ev["step_n"][e] = e + 1
            n_ho += int(np.sum(prev_serv != ch.current_BS))
            d = ue[:, None, :2] - bs[None, :, :2]
            n_d0 += int(np.sum((d ** 2).sum(-1) == 0))
    meta = {"seed": seed, "n_walkers": U, "n_ue_channel_rows": U, "n_bs": B, "n_ue": U,
            "n_groups": len(groups), "groups": np.array(groups, np.int32), "grid": grid,
            "warmup_ticks": W, "max_step": 2000, "bs_init": np.array(bs_init, np.int16)}
    extra = {"init_ue_loc": np.array(probe_last_ue(probe, W), np.int16), "init_serving": c0[0],
             "init_cur_sinr": c0[1], "init_out_mask": c0[4], "n_handovers": n_ho,
             "n_zero_distance": n_d0}
    finish_fixture(name, meta, rec, probe, ev, extra)


def run_trace_scenario(name, seed, n_steps, trace_len):
    """BASELINE config 1: MobiEnvironment(4, 40, 100, "read_trace", file) driven by step_test
    (main_test.py:51-75).  ue_trace_10k.npy is absent from the mount (.MISSING_LARGE_BLOBS), so a trace of the
    same format (T, 40, 2) int is produced the way README.md:32 / main_test.py:114 describe: the group model's
    integer positions."""
    import tempfile

    mods = load_reference()
    me, um = mods["mobile_env"], mods["ue_mobility"]
    B, U, G = 4, 40, 100
    np.random.seed(seed + 1000)
    mm = um.reference_point_group([10, 10, 10, 10], dimensions=(G, G), velocity=(0, 1), aggregation=0.8)
    for _ in range(200):
        next(mm)
    trace = np.stack([next(mm).astype(int) for _ in range(trace_len)]).astype(np.int16)
    rs = np.random.RandomState(777)
    actions = rs.randint(0, 625, n_steps)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "trace.npy")
        np.save(path, trace)
        np.random.seed(seed)
        with Recorder(mods) as rec:
            env = me.MobiEnvironment(B, U, G, "read_trace", path)
            c0 = channel_snapshot(env.channel, U)
            script = [("reset",)] + [("step", int(a)) for a in actions[:n_steps // 2]] + [("reset",)] + \
                     [("step", int(a)) for a in actions[n_steps // 2:]]
            E = len(script)
            ev = {
                "ev_kind": np.zeros(E, np.int8), "ev_action": np.zeros(E, np.int64),
                "ev_trace_row": np.zeros(E, np.int32),
                "ue_loc": np.zeros((E, U, 2), np.int16), "bs_loc": np.zeros((E, B, 2), np.int16),
                "serving": np.zeros((E, U), np.int8), "cur_sinr": np.zeros((E, U)),
                "fifo": np.zeros((E, 3, U), np.int8), "fifo_depth": np.zeros(E, np.int8),
                "out_mask": np.zeros((E, U), bool), "mean_sinr": np.full(E, np.nan),
                "n_out": np.zeros(E, np.int32), "reward": np.full(E, np.nan),
                "done": np.zeros(E, bool), "step_n": np.zeros(E, np.int32),
                "state_nz": np.zeros((E, U + B, 4), np.int16), "bs_actions": np.zeros((E, B), np.int8),
                "outage_fraction": np.zeros(E),
            }
            for e, item in enumerate(script):
                if item[0] == "reset":
                    state = env.reset()
                    ev["ev_kind"][e] = 0
                    ev["ev_trace_row"][e] = 0
                else:
                    ev["ev_trace_row"][e] = env.step_n          # mobile_env.py:203 ueLoc_trace[self.step_n]
                    state, reward, done, info = env.step_test(item[1])
                    ev["ev_kind"][e] = 1
                    ev["ev_action"][e] = item[1]
                    ev["reward"][e] = reward
                    ev["done"][e] = done
                    ev["mean_sinr"][e] = float(np.mean(env.channel.current_BS_sinr))
                    ev["n_out"][e] = int(round(info.outage_fraction * U))
                    ev["outage_fraction"][e] = info.outage_fraction
                    ev["bs_actions"][e] = info.bs_actions
                serv, sinr, fifo, depth, omask = channel_snapshot(env.channel, U)
                ev["ue_loc"][e] = np.asarray(env.ueLoc)[:, :2]
                ev["bs_loc"][e] = env.bsLoc[:, :2]
                ev["serving"][e], ev["cur_sinr"][e] = serv, sinr
                ev["fifo"][e], ev["fifo_depth"][e], ev["out_mask"][e] = fifo, depth, omask
                ev["step_n"][e] = env.step_n
                ev["state_nz"][e] = state_nonzeros(state, U + B)
    assert len(rec.rand_log) == 0, "read_trace mode must not draw uniforms"
    normals = np.array(rec.normal_log)
    assert normals.size == (E + 1) * U * B
    fx = {"seed": seed, "n_walkers": U, "n_ue_channel_rows": U, "n_bs": B, "n_ue": U, "n_groups": 4,
          "groups": np.array([10, 10, 10, 10], np.int32), "grid": G, "warmup_ticks": 0, "max_step": int(me.MAXSTEP),
          "bs_init": np.array(env.initBsLoc[:, :2], np.int16), "trace": trace,
          "init_ue_loc": trace[0], "init_serving": c0[0], "init_cur_sinr": c0[1], "init_out_mask": c0[4]}
    fx.update(ev)
    from fixture_io import regenerate_trace_fading
    assert np.array_equal(regenerate_trace_fading(fx), normals.reshape(E + 1, U, B))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in fx.items()})
    print("%-34s E=%d trace rows=%d handovers=n/a  %.1f KB" % (name, E, trace_len, os.path.getsize(path) / 1024.0))


def run_area_scenario(name, seed, grid, bs_cells):
    """LTEChannel.GetSinrInArea (channel.py:411-433) for given UAV cells; the draws are (G-1)^2 * B normals in the
    reference's call order (interferers ascending, then the nearest UAV) and are regenerated from the seed."""
    mods = load_reference()
    chm = mods["channel"]
    B, U = len(bs_cells), 8
    bs = np.concatenate([np.array(bs_cells, dtype=int), np.full((B, 1), 10, dtype=int)], axis=1)
    ue = np.zeros((U, 3), dtype=int) + 5
    np.random.seed(seed + 5000)
    ch = chm.LTEChannel(U, B, [1, grid, 1, grid], ue, bs)      # its own draws come from another seed
    np.random.seed(seed)
    with Recorder(mods) as rec:
        sinr = ch.GetSinrInArea(bs)
    normals = np.array(rec.normal_log)
    W = grid - 1
    assert normals.size == W * W * B and sinr.shape == (grid, grid)
    rs = np.random.RandomState(seed)
    assert np.array_equal(rs.normal(0.0, 2.0, size=(W * W, B)).ravel(), normals)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, seed=seed, grid=grid, n_bs=B, bs_xy=np.array(bs_cells, np.int32), sinr_area=sinr)
    print("%-34s cells=%d B=%d  min %.1f dB max %.1f dB  %.1f KB" % (name, W * W, B, sinr[1:, 1:].min(), sinr.max(),
                                                                  os.path.getsize(path) / 1024.0))


def probe_last_ue(probe, W):
    s = probe.snaps[W - 1]
    return np.stack([s["x"], s["y"]], axis=1).astype(int)


def digits_to_action(d):
    a = 0
    for v in d:  # most-significant digit -> UAV 0 (ue_mobility.py:310-336)
        a = a * 5 + int(v)
    return a


def scenario_list():
    sc = {}

    def s1():
        rng = np.random.RandomState(12345)  # action lists only
        script = [("reset",)] + [("step", a) for a in rng.randint(0, 625, 150)]
        script += [("reset",)] + [("step", a) for a in rng.randint(0, 625, 30)]
        run_env_scenario("ref_4x40_g100_seed1", 1, 40, 100, [10, 10, 10, 10], script)

    def s2():
        rng = np.random.RandomState(12346)  # action lists only
        script = [("reset",)] + [("step", a) for a in rng.randint(0, 625, 60)]
        run_env_scenario("ref_4x40_g200_seed2", 2, 40, 200, [10, 10, 10, 10], script)

    def s3():
        rng = np.random.RandomState(12347)  # action lists only
        # full episode to `done`, then reset and a few more steps; mostly "stay"-biased actions so
        # that the UAVs keep moving for a while before the collision freeze
        acts = []
        for _ in range(2000):
            d = [rng.randint(0, 5) if rng.rand() < 0.5 else 4 for _ in range(4)]
            acts.append(digits_to_action(d))
        script = [("reset",)] + [("step", a) for a in acts]
        script += [("reset",)] + [("step", a) for a in rng.randint(0, 625, 10)]
        run_env_scenario("ref_4x20_g100_seed3_episode", 3, 20, 100, [5, 5, 5, 5], script)

    def s4():
        rng = np.random.RandomState(12348)  # action lists only
        grid = 100
        lat = [grid // 8 + k * (grid // 4) for k in range(4)]
        bs_init = [(x, y) for x in lat for y in lat]
        acts = [int(rng.randint(0, 5 ** 8)) * (5 ** 8) + int(rng.randint(0, 5 ** 8)) for _ in range(40)]
        run_parts_scenario("ref_16x200_g100_seed4_parts", 4, 16, [50, 50, 50, 50], grid, bs_init, acts)

    def s5():
        rng = np.random.RandomState(12349)  # action lists only
        # scripted walls + collision freeze (ue_mobility.py:221-235,256-268)
        script = [("reset",)]
        script += [("step", digits_to_action([1, 2, 3, 0]))] * 40   # every UAV into a wall
        script += [("step", digits_to_action([0, 3, 2, 1]))] * 30   # back towards the centre
        script += [("step", digits_to_action([2, 4, 4, 4]))] * 30   # UAV0 up towards UAV1
        script += [("step", digits_to_action([4, 3, 4, 4]))] * 30   # UAV1 down onto UAV0 -> freeze
        script += [("step", a) for a in rng.randint(0, 625, 30)]
        run_env_scenario("ref_4x40_g100_seed5_walls", 5, 40, 100, [10, 10, 10, 10], script)

    def s6():
        run_trace_scenario("ref_trace_4x40_g100_seed6", 6, 240, 260)

    def s7():
        run_area_scenario("ref_area_4bs_g100_seed7", 7, 100, [(25, 25), (25, 75), (75, 25), (75, 75)])
        run_area_scenario("ref_area_5bs_g40_seed8", 8, 40, [(3, 3), (9, 31), (20, 20), (20, 21), (37, 12)])

    sc["s1"], sc["s2"], sc["s3"], sc["s4"], sc["s5"], sc["s6"], sc["s7"] = s1, s2, s3, s4, s5, s6, s7
    return sc


if __name__ == "__main__":
    import contextlib
    import io

    todo = sys.argv[1:]
    for key, fn in scenario_list().items():
        if todo and key not in todo:
            continue
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):  # the reference prints a banner / "COLLIDED"
            try:
                fn()
            except Exception:
                sys.stderr.write(buf.getvalue()[-2000:])
                raise
        lines = [l for l in buf.getvalue().splitlines() if l.startswith("ref_")]
        print("\n".join(lines))
