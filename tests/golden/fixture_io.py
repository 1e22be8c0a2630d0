"""Golden-fixture I/O shared by the generator and the parity tests.

TEST INFRASTRUCTURE (pure numpy; travels to the GPU box).

A fixture holds *data only*: the seed of the legacy global NumPy stream the
reference was run with, the per-tick "group arrived" masks (which make the
reference's data-dependent draw order reproducible without the reference), the
action list, and every output the reference produced.  ``regenerate_draws``
rebuilds the exact uniform / normal draws the reference consumed, in the
layout the oracle and the C-ABI take as *injected randomness*:

  init_u_x, init_u_y, init_u_th : [U]      uniforms in [0,1)   (ue_mobility.py:434-437)
  init_u_g                      : [5, Gr]  g_x,g_y,g_fl,g_v,g_theta (ue_mobility.py:442-446)
  tick_u_th                     : [T, U]   heading uniforms    (ue_mobility.py:508)
  tick_u_grp                    : [T, Gr, 3] (theta, fl, v) uniforms of the groups
                                  that arrived this tick, 0 elsewhere (ue_mobility.py:517-521)
  fading                        : [C, U_ch, B] N(0,2) draws, UE-major (channel.py:240,254-256)

Tick t = t-th ``next()`` of the mobility generator since construction
(``warmup_ticks`` of them happen inside the constructor); channel update c =
c-th GetChannelGainAll call (0 = LTEChannel.__init__).
"""
import numpy as np


def regenerate_draws(fx):
    """Replay the reference's draw ORDER on RandomState(seed).

    Order (mobile_env.py:76-98, ue_mobility.py:434-448,508-521, channel.py:92,254-256):
    init uniforms; ``warmup_ticks`` ticks; channel-init normals; then per event
    one tick followed by one channel update.  ``read_trace`` events have no tick.
    """
    rs = np.random.RandomState(int(fx["seed"]))
    U = int(fx["n_walkers"])          # walkers in the mobility model
    U_ch = int(fx["n_ue_channel_rows"])  # rows GetChannelGainAll iterates over
    B = int(fx["n_bs"])
    Gr = int(fx["n_groups"])
    arrived = np.asarray(fx["tick_arrived"]).astype(bool)
    T = arrived.shape[0]
    W = int(fx["warmup_ticks"])
    E = T - W
    out = {}
    out["init_u_x"] = rs.rand(U)
    out["init_u_y"] = rs.rand(U)
    out["init_u_th"] = rs.rand(U)
    g = np.zeros((5, Gr))
    for k in range(5):
        g[k] = rs.rand(Gr)
    # ue_mobility.py:442-446 draw order: g_x, g_y, g_fl, g_velocity, g_theta
    out["init_u_g"] = g
    th = np.zeros((T, U))
    grp = np.zeros((T, Gr, 3))
    fading = np.zeros((E + 1, U_ch, B))

    def tick(t):
        th[t] = rs.rand(U)
        idx = np.flatnonzero(arrived[t])
        if idx.size:
            grp[t, idx, 0] = rs.rand(idx.size)  # g_theta  (:517)
            grp[t, idx, 1] = rs.rand(idx.size)  # g_fl     (:520)
            grp[t, idx, 2] = rs.rand(idx.size)  # g_velocity (:521)

    def chan(c):
        # scalar normal(0,2) calls in UE-major/BS-minor order == one C-order vector draw
        fading[c] = rs.normal(0.0, 2.0, size=(U_ch, B))

    for t in range(W):
        tick(t)
    chan(0)
    for e in range(E):
        tick(W + e)
        chan(e + 1)
    out["tick_u_th"] = th
    out["tick_u_grp"] = grp
    out["fading"] = fading
    return out


def regenerate_trace_fading(fx):
    """read_trace fixtures: the only draws are the U*B normals of each channel update (constructor, then one
    per reset / step_test), consumed from RandomState(seed) in order."""
    rs = np.random.RandomState(int(fx["seed"]))
    E = len(fx["ev_kind"])
    U, B = int(fx["n_ue"]), int(fx["n_bs"])
    return np.stack([rs.normal(0.0, 2.0, size=(U, B)) for _ in range(E + 1)])


def load_fixture(path):
    with np.load(path, allow_pickle=False) as z:
        fx = {k: z[k] for k in z.files}
    for k in ("seed", "n_walkers", "n_ue_channel_rows", "n_bs", "n_ue", "n_groups", "grid", "warmup_ticks"):
        fx[k] = int(fx[k])
    return fx
