"""CPU: the rotation schedule of uavenv_step_many (csrc/uavenv_capi.hip: build_schedule) as pure host arithmetic, through
the test hook uavenv_debug_schedule -- no device needed.  For many (env-wavefronts W, slots S, steps T): every step of every job exactly
once and in order; at most three pieces per slot, in the column order publish / whole / wait; a job is one whole piece or two pieces on
different slots; the publishing piece is the FIRST piece of its slot (so it never waits: every hand-off wait ends, whatever order the
hardware starts wavefronts in); per-slot work within the makespan ceil(W T / S); the padding rows of the last workgroup are empty.
What the schedule stands for: T consecutive MobiEnvironment.step calls of every env (mobile_env.py:150-194), each env's steps in order."""
import ctypes as C

import numpy as np
import pytest

from drl_uav_cellularnet_amd import _capi

WAIT, PUBLISH = 1, 2


def _schedule(W, S, T):
    lib = _capi.load()
    rows, mk = C.c_int64(0), C.c_int64(0)
    rc = lib.uavenv_debug_schedule(W, S, T, None, 0, C.byref(rows), C.byref(mk))
    if rc != 0:
        return None
    tab = np.zeros((rows.value, 3, 4), np.int32)
    assert lib.uavenv_debug_schedule(W, S, T, tab.ctypes.data, rows.value, C.byref(rows), C.byref(mk)) == 0
    return tab, int(mk.value)


SHAPES = [(1366, 1024, 100), (1366, 1024, 20), (1366, 1024, 2), (2731, 2048, 100), (2731, 2048, 20), (3000, 2048, 50), (34, 24, 7), (34, 26, 50),
          (101, 80, 33), (11, 8, 6), (7, 5, 5), (10, 7, 6), (22, 16, 100), (34, 26, 21), (101, 87, 40), (22, 20, 12), (1025, 1024, 64),
          (2047, 1024, 2000), (1500, 1024, 3), (5, 4, 2), (3, 2, 4)]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "W%d_S%d_T%d" % s)
def test_schedule_invariants(shape):
    W, S, T = shape
    got = _schedule(W, S, T)
    M = -(-W * T // S)
    if M - T < 1 or M >= 2 * T:
        assert got is None                                          # no room to split (or more than three pieces per slot): the plain launch runs
        return
    assert got is not None
    tab, mk = got
    assert mk == M and tab.shape[0] == -(-S // 4) * 4 and not tab[S:].any()       # rows padded to whole workgroups of 4 wavefronts, padding empty
    next_step = np.zeros(W, np.int64)
    pieces_of = [[] for _ in range(W)]
    for slot in range(S):
        work = 0
        for col in range(3):
            ew, t0, nt, bits = (int(v) for v in tab[slot, col])
            if nt == 0:
                assert (ew, t0, bits) == (0, 0, 0)
                continue
            assert 0 <= ew < W and 0 <= t0 and t0 + nt <= T
            work += nt
            pieces_of[ew].append((t0, nt, slot, col, bits))
            # the column says the kind, the bits say the same
            assert (col == 0) == (t0 + nt < T) and (col == 2) == (t0 > 0)
            assert bool(bits & PUBLISH) == (col == 0) and bool(bits & WAIT) == (col == 2)
        assert work <= M                                            # a slot's work fits the makespan
    for ew, ps in enumerate(pieces_of):
        ps.sort()
        assert 1 <= len(ps) <= 2
        if len(ps) == 1:
            assert ps[0][:2] == (0, T) and ps[0][3] == 1            # a whole job sits in the middle column
        else:
            (t0a, na, slot_a, col_a, _), (t0b, nb, slot_b, col_b, _) = ps
            assert t0a == 0 and t0b == na and na + nb == T and slot_a != slot_b and (col_a, col_b) == (0, 2)
    # the publishing piece is the first piece of its slot by construction of the columns (column 0 runs first); the waiting piece the last


def test_no_schedule_where_none_can_exist():
    for W, S, T in ((1024, 1024, 100), (1000, 1024, 100), (2048, 1024, 100), (3000, 1024, 100), (1366, 1024, 1), (0, 4, 10), (10, 0, 10), (10, 4, 0)):
        assert _schedule(W, S, T) is None, (W, S, T)
    lib = _capi.load()
    rows, mk = C.c_int64(0), C.c_int64(0)
    small = np.zeros((4, 3, 4), np.int32)
    assert lib.uavenv_debug_schedule(1366, 1024, 100, small.ctypes.data, 4, C.byref(rows), C.byref(mk)) != 0      # table too small: refused, nothing written
    assert not small.any()


def test_total_work_is_balanced_at_the_baseline_shape():
    """BASELINE configs[1]: 4096 envs x 20 UEs = 1366 env-wavefronts on 1024 SIMDs: all but the last few slots carry 134 of the 136 600
    wavefront-steps (the plain launch: 342 SIMDs carry 200, 682 carry 100)."""
    tab, M = _schedule(1366, 1024, 100)
    work = tab[:1024, :, 2].sum(axis=1)
    assert M == 134 and int(work.sum()) == 136600 and int(work.max()) == 134
    assert int((work == 134).sum()) >= 1019                         # 134 x 1024 - 136 600 = 616 step-times of slack: the last 5 slots take it
    split = int((tab[:1024, 0, 2] > 0).sum())
    assert split == int((tab[:1024, 2, 2] > 0).sum())               # as many publishing pieces as waiting ones: one hand-off per split job
