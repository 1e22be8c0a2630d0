"""CPU: the PyTorch actor-critic / A2C pieces against NumPy restatements of the reference's TF1 formulas
(main.py:64-74,143-156,300-301; a2c_single_thread.py:176-183).  TensorFlow is not installable here, so this row
is PARITY UNPINNED against real TF outputs; what is pinned is internal consistency with the formulas as written."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_obs(N, U, B, G, seed=0):
    rs = np.random.RandomState(seed)
    ue = rs.randint(0, G, size=(N, U, 2))
    ue[:, 1] = ue[:, 0]                                    # force co-located UEs (counts > 1)
    obs = {"ue_xy": torch.tensor(ue, dtype=torch.int16), "bs_xy": torch.tensor(rs.randint(2, G - 2, size=(N, B, 2)), dtype=torch.int32),
           "serving": torch.tensor(rs.randint(0, B, size=(N, U)), dtype=torch.int8)}
    obs["serving"][:, 1] = obs["serving"][:, 0]            # ... served by the same UAV -> same cell of the same plane
    return obs


def _dense_state(obs, G, B):
    N, U = obs["serving"].shape
    st = np.zeros((N, B + 1, G, G), np.float32)
    for n in range(N):
        for b in range(B):
            x, y = obs["bs_xy"][n, b].tolist()
            st[n, 0, x, y] += 1                            # GetGridMap, ue_mobility.py:184-186
        for u in range(U):
            x, y = obs["ue_xy"][n, u].tolist()
            st[n, 1 + int(obs["serving"][n, u]), x, y] += 1   # GetCurrentAssociationMap, channel.py:404-406
    return st


def test_sparse_first_layer_equals_dense_matmul_on_the_count_map():
    from drl_uav_cellularnet_amd.agent import ACNet, obs_to_indices

    N, U, B, G = 6, 20, 4, 12
    obs = _fake_obs(N, U, B, G)
    net = ACNet(G * G * (B + 1), 5 ** B, seed=6).double()
    idx = obs_to_indices(obs, G, B)
    assert idx.shape == (N, B + U) and idx.dtype == torch.int64
    s = torch.tensor(_dense_state(obs, G, B).reshape(N, -1), dtype=torch.float64)      # np.ravel(state), main.py:190
    assert float(s.max()) >= 2.0                                                       # duplicates really occur
    p_s, v_s = net(idx)
    p_d, v_d = net.forward_dense(s)
    torch.testing.assert_close(p_s, p_d, rtol=1e-12, atol=1e-14)
    torch.testing.assert_close(v_s, v_d, rtol=1e-12, atol=1e-12)
    torch.testing.assert_close(p_s.sum(dim=1), torch.ones(N, dtype=torch.float64))
    torch.testing.assert_close(net.actor_only(idx), p_s)
    torch.testing.assert_close(net.critic_only(idx), v_s)


def test_parameter_count_matches_the_survey():
    from drl_uav_cellularnet_amd.agent import ACNet, expected_param_count, grad_allreduce_bytes

    assert expected_param_count(50000, 625) == 20206626
    net = ACNet(500, 625)
    assert sum(p.numel() for p in net.parameters()) == expected_param_count(500, 625)
    assert grad_allreduce_bytes(net) == 4 * expected_param_count(500, 625)
    assert float(net.a_b1.detach().abs().max()) == 0.0 and 0.08 < float(net.a_w1.detach().std()) < 0.12   # N(0, 0.1) / zero bias


def test_losses_match_numpy_restatement():
    from drl_uav_cellularnet_amd.agent import a2c_losses

    rs = np.random.RandomState(1)
    M, A = 64, 25
    logits = rs.randn(M, A)
    p = np.exp(logits) / np.exp(logits).sum(1, keepdims=True)
    v, vt = rs.randn(M, 1), rs.randn(M, 1)
    a = rs.randint(0, A, M)
    td = vt - v                                                                        # main.py:64
    c_ref = np.mean(td ** 2)                                                           # :66
    logp = np.log(p[np.arange(M), a][:, None] + 1e-5)                                  # :69
    ent = -(p * np.log(p + 1e-5)).sum(1, keepdims=True)                                # :71
    a_ref = np.mean(-(0.001 * ent + logp * td))                                        # :70-74
    pt = torch.tensor(p, requires_grad=True)
    vtens = torch.tensor(v, requires_grad=True)
    a_loss, c_loss = a2c_losses(pt, vtens, torch.tensor(a), torch.tensor(vt))
    np.testing.assert_allclose(float(a_loss), a_ref, rtol=1e-12)
    np.testing.assert_allclose(float(c_loss), c_ref, rtol=1e-12)
    a_loss.backward()
    assert vtens.grad is None                                                          # stop_gradient(td), main.py:70


def test_tf1_rmsprop_semantics():
    from drl_uav_cellularnet_amd.agent import TFRMSProp

    rs = np.random.RandomState(2)
    w0 = rs.randn(7, 3)
    w = torch.nn.Parameter(torch.tensor(w0))
    opt = TFRMSProp([w], lr=1e-4)
    ref_w, ms = w0.copy(), np.ones_like(w0)                                            # accumulator starts at ONE
    for _ in range(4):
        g = rs.randn(7, 3)
        w.grad = torch.tensor(g)
        opt.step()
        ms = 0.9 * ms + 0.1 * g * g
        ref_w = ref_w - 1e-4 * g / np.sqrt(ms + 1e-10)                                 # epsilon inside the sqrt
    np.testing.assert_allclose(w.detach().numpy(), ref_w, rtol=1e-13)


def test_nstep_returns_match_the_reference_loop():
    from drl_uav_cellularnet_amd.agent import nstep_returns

    rs = np.random.RandomState(3)
    T, N = 50, 5
    r, boot = rs.randn(T, N), rs.randn(N)
    boot[2] = 0.0                                                                      # a finished episode
    want = np.zeros((T, N))
    for n in range(N):
        ve, buf = boot[n], []
        for x in r[::-1, n]:                                                           # a2c_single_thread.py:178-183
            ve = x + 0.9 * ve
            buf.append(ve)
        buf.reverse()
        want[:, n] = buf
    got = nstep_returns(torch.tensor(r), torch.tensor(boot)).numpy()
    np.testing.assert_allclose(got, want, rtol=1e-13)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _grads(net, idx, act, vt):
    from drl_uav_cellularnet_amd.agent import a2c_losses

    for p in net.parameters():
        p.grad = None
    a_prob, v = net(idx)
    a_loss, c_loss = a2c_losses(a_prob, v, act, vt)
    (a_loss + c_loss).backward()
    return [p.grad.clone() for p in net.parameters()]


def _ddp_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from drl_uav_cellularnet_amd.agent import ACNet, allreduce_mean_grads, obs_to_indices

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        N, U, B, G = 8, 20, 4, 10
        obs = _fake_obs(N, U, B, G, seed=5)
        idx = obs_to_indices(obs, G, B)
        rs = np.random.RandomState(9)
        act = torch.tensor(rs.randint(0, 625, N))
        vt = torch.tensor(rs.randn(N, 1))
        net = ACNet(G * G * (B + 1), 625, seed=6).double()
        half = slice(rank * N // world, (rank + 1) * N // world)
        _grads(net, idx[half], act[half], vt[half])                                    # this rank's shard of the batch
        n = allreduce_mean_grads(list(net.parameters()))
        mine = [p.grad.clone() for p in net.parameters()]
        if rank == 0:
            full = _grads(net, idx, act, vt)                                           # single-process, whole batch
            err = max(float((a - b).abs().max()) for a, b in zip(mine, full))
            q.put(("ok", err, n))
        dist.barrier()
    except Exception as exc:
        q.put(("err", repr(exc), 0))
        raise
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_allreduce_equals_full_batch_gradient():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    status, err, n = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert status == "ok", err
    assert err < 1e-12, "mean of the shard gradients differs from the full-batch gradient by %g" % err
    assert n == 2 * (500 * 200 + 200 + 200 * 200 + 200) + 200 * 625 + 625 + 200 + 1


def test_actor_checkpoint_roundtrip_without_pickle(tmp_path):
    from drl_uav_cellularnet_amd.agent import ACNet, load_actor_npz, save_actor_npz

    a, b = ACNet(300, 25, seed=1), ACNet(300, 25, seed=2)
    path = os.path.join(tmp_path, "Global_A_PARA.npz")
    save_actor_npz(a, path)
    with np.load(path, allow_pickle=False) as z:                       # loadable with pickling disabled
        assert sorted(z.files) == sorted(("a_w1", "a_b1", "a_w2", "a_b2", "a_w3", "a_b3"))
    load_actor_npz(b, path)
    for pa, pb in zip(a.actor_params(), b.actor_params()):
        assert torch.equal(pa, pb)
    assert not torch.equal(a.c_w1, b.c_w1)                             # the critic is never saved (main.py:265-269)
    with pytest.raises(ValueError):
        load_actor_npz(ACNet(301, 25), path)
    np.savez(os.path.join(tmp_path, "other.npz"), arr_0=np.zeros(3))
    with pytest.raises(ValueError):
        load_actor_npz(b, os.path.join(tmp_path, "other.npz"))


def test_gradient_heuristic_side_means():
    from drl_uav_cellularnet_amd.heuristics import side_means

    sinr = np.array([10.0, 20.0, -5.0, 1.0])
    ue = np.array([[5, 5], [9, 1], [2, 8], [5, 9]])
    m = side_means(sinr, ue, np.array([5, 5, 10]))
    np.testing.assert_allclose(m, [20.0, np.mean([10.0, -5.0, 1.0]), np.mean([-5.0, 1.0]), np.mean([10.0, 20.0])])
    m = side_means(sinr, ue, np.array([9, 0, 10]))                     # nobody has x > 9 or y <= 0 ... y<=0: none
    assert np.isnan(m[0]) and not np.isnan(m[1]) and int(np.nanargmin(m)) in (1, 2, 3)


def test_sample_actions_is_an_inverse_cdf_draw():
    """agent.sample_actions against np.random.choice's algorithm (main.py:167-168): zero-probability actions are never drawn,
    frequencies follow p, the draw is a pure function of (prob, generator state)."""
    from drl_uav_cellularnet_amd.agent import sample_actions

    p = torch.tensor([[0.0, 0.5, 0.0, 0.25, 0.25, 0.0],        # zeros at both ends and inside
                      [1.0, 0.0, 0.0, 0.0, 0.0, 0.0],
                      [0.0, 0.0, 0.0, 0.0, 0.0, 1.0],
                      [0.1, 0.1, 0.2, 0.3, 0.2, 0.1]])
    n = 40000
    g = torch.Generator().manual_seed(5)
    draws = sample_actions(p.repeat(n, 1), g).reshape(n, 4)
    assert draws.dtype == torch.int64 and int(draws.min()) >= 0 and int(draws.max()) <= 5
    freq = torch.stack([(draws == a).float().mean(dim=0) for a in range(6)], dim=1)     # [4 rows of p, 6 actions]
    assert torch.all(freq[p == 0] == 0)
    assert torch.allclose(freq, p, atol=0.01)
    g2 = torch.Generator().manual_seed(5)
    assert torch.equal(sample_actions(p.repeat(n, 1), g2).reshape(n, 4), draws)
    # the same uniforms through numpy's own steps
    g3 = torch.Generator().manual_seed(11)
    q = torch.softmax(torch.randn(64, 625, generator=torch.Generator().manual_seed(1)), dim=1)
    mine = sample_actions(q, g3)
    u = torch.rand((64, 1), generator=torch.Generator().manual_seed(11))
    cdf = np.cumsum(q.numpy(), axis=1)
    want = [int(np.searchsorted(cdf[i], float(u[i, 0]) * cdf[i, -1], side="right")) for i in range(64)]
    assert mine.tolist() == [min(w, 624) for w in want]
