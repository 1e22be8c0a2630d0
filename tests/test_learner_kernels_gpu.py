"""GPU: the learner-side kernels of libuavagent.so (include/uavagent.h, ABI 3) against their plain PyTorch forms, and the fused
A2C update / graph-captured rollout against the autograd / eager paths they replace.  The formulas are the reference's
(main.py:64-74 loss, :143-156 network, :165-169 action choice, :300-301 RMSProp); TensorFlow is not installable here and the
reference holds no fixtures for its learner, so this parity is against restatements: "parity unpinned" (DESIGN.md section 9)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _torch():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch


def test_obs_indices_kernel_equals_obs_to_indices():
    torch = _torch()
    from drl_uav_cellularnet_amd import BatchedMobiEnv, _agent_capi as A
    from drl_uav_cellularnet_amd.agent import obs_to_indices

    env = BatchedMobiEnv(777, nBS=4, nUE=20, grid_n=100)
    for t in range(3):
        env.step(torch.randint(0, 625, (777,), device=env.device))
    obs = {k: v.clone() for k, v in env.observation().items()}
    obs["ue_xy"][5, 3, 0] = 100            # a walker exactly on x == G: no cell -> -1 in both
    obs["ue_xy"][9, 0, 1] = -1
    want = obs_to_indices(obs, 100, 4)
    got = A.obs_indices(obs, 100, 4)
    assert torch.equal(got, want)
    assert int(got[5, 4 + 3]) == -1 and int(got[9, 4]) == -1 and int((got < 0).sum()) == 2


@pytest.mark.parametrize("n_envs,n_ue,groups,two", [(777, 20, [5, 5, 5, 5], True), (130, 40, None, True), (65, 20, [5, 5, 5, 5], False),
                                                     (33, 12, [3, 3, 3, 3], True)])
def test_first_layer_from_obs_equals_obs_indices_then_first_layer(n_envs, n_ue, groups, two):
    """uavagent_first_layer_from_obs_f32 (ABI 4) = uavagent_obs_indices + uavagent_first_layer_f32, bit for bit: the stored index list and
    both trunks' activations, for the K = 24 / 44 instantiations and the run-time-K one, with off-grid walkers (-1 = no row)."""
    torch = _torch()
    from drl_uav_cellularnet_amd import BatchedMobiEnv, _agent_capi as A

    kw = {} if groups is None else {"groups": groups}
    env = BatchedMobiEnv(n_envs, nBS=4, nUE=n_ue, grid_n=100, **kw)
    for t in range(3):
        env.step(torch.randint(0, 625, (n_envs,), device=env.device))
    obs = {k: v.clone() for k, v in env.observation().items()}
    obs["ue_xy"][5, 3, 0] = 100
    obs["ue_xy"][9, 0, 1] = -1
    obs["ue_xy"][11] = -5                                          # a whole env's walkers off the grid: only its UAV rows count
    g = torch.Generator(device="cuda").manual_seed(11)
    S, H = 5 * 100 * 100, 200
    wa, wc = torch.randn(S, H, device="cuda", generator=g), torch.randn(S, H, device="cuda", generator=g)
    ba, bc = torch.randn(H, device="cuda", generator=g), torch.randn(H, device="cuda", generator=g)
    idx = A.obs_indices(obs, 100, 4)
    for relu6 in (True, False):
        want = A.sparse_rows_sum(idx, wa, ba, wc if two else None, bc if two else None, relu6=relu6)
        oa, oc = torch.full((n_envs, H), 7.0, device="cuda"), (torch.full((n_envs, H), 7.0, device="cuda") if two else None)
        idx_out = torch.full_like(idx, 12345)
        A.first_layer_from_obs(obs, 100, wa, ba, wc if two else None, bc if two else None, oa, oc, idx_out=idx_out, relu6=relu6)
        assert torch.equal(idx_out, idx)
        if two:
            assert torch.equal(oa, want[0]) and torch.equal(oc, want[1])
        else:
            assert torch.equal(oa, want)
        oa2 = torch.empty_like(oa)
        A.first_layer_from_obs(obs, 100, wa, ba, None, None, oa2, None, idx_out=None, relu6=relu6)       # no index list asked for
        assert torch.equal(oa2, want[0] if two else want)
    with pytest.raises(A.UavAgentError):                          # a table too small for (n_bs + 1) * G^2 rows
        A.first_layer_from_obs(obs, 100, wa[:40000], ba, None, None, oa, None)


def test_rollout_with_index_lists_built_in_the_gather_is_the_same_rollout():
    """A2CRunner(fused_obs=True), the default, against the separate obs_indices launch per step: the same samples, actions, rewards,
    activations and, after the update, parameters -- bit for bit, through the captured graph."""
    torch = _torch()
    from drl_uav_cellularnet_amd import BatchedMobiEnv
    from drl_uav_cellularnet_amd.agent import A2CRunner

    runs = []
    for fused in (True, False):
        env = BatchedMobiEnv(512, nBS=4, nUE=20, grid_n=100, groups=[5, 5, 5, 5])
        r = A2CRunner(env, rollout=7, fused_obs=fused)
        assert r.fused_obs == fused
        for _ in range(3):
            r.train_rollout()
        idx, act, rew, boot = r.collect()
        runs.append((idx.clone(), act.clone(), rew.clone(), boot.clone(), r._fwd["h1c"].clone(), r.idx_buf.clone(), r.flat.w.clone()))
    for a, b in zip(*runs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("kw", [{"overlap_dw": True}, {"early_sort": False}, {"overlap_dw": True, "early_sort": False}], ids=lambda k: "+".join(sorted(k)))
def test_update_stream_variants_leave_the_same_parameters(kw):
    """The update's optional side-stream forms (dW GEMMs beside the table gradient; the table gradient's sort before the forward pass)
    run the same kernels on the same operands in another order of issue: parameters and RMSProp accumulators bit for bit."""
    torch = _torch()
    from drl_uav_cellularnet_amd import BatchedMobiEnv
    from drl_uav_cellularnet_amd.agent import A2CRunner

    outs = []
    for k in ({}, kw):
        env = BatchedMobiEnv(1024, nBS=4, nUE=20, grid_n=100, groups=[5, 5, 5, 5])
        r = A2CRunner(env, rollout=40, **k)                            # 40 960 samples: the update's large-M kernels
        for _ in range(2):
            r.train_rollout()
        assert r.stats["dw_on_side_stream"] == bool(k.get("overlap_dw", False))
        outs.append((r.flat.w.clone(), r.flat.ms.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_sample_actions_kernel_is_the_inverse_cdf_draw():
    torch = _torch()
    from drl_uav_cellularnet_amd import _agent_capi as A

    g = torch.Generator(device="cuda").manual_seed(3)
    N, NA = 8192, 625
    logits = torch.randn(N, NA, device="cuda", generator=g) * 3.0
    logits[:, 17] = -1e30                                          # probability exactly 0: never drawn
    u = torch.rand(N, device="cuda", generator=g)
    u[0], u[1] = 0.0, 1.0 - 2 ** -24
    prob = torch.empty_like(logits)
    a = A.sample_actions(logits, u, prob_out=prob)
    torch.testing.assert_close(prob, torch.softmax(logits, dim=1), rtol=1e-5, atol=1e-9)
    p64 = torch.softmax(logits.double(), dim=1)
    cdf = p64.cumsum(dim=1)
    want = torch.searchsorted(cdf, (u.double() * cdf[:, -1]).unsqueeze(1), right=True).squeeze(1).clamp_(max=NA - 1)
    diff = (a != want)
    assert int(diff.sum()) <= 2                                    # float32 vs float64 CDF: only a draw ON a boundary may differ
    if bool(diff.any()):
        rows = diff.nonzero().squeeze(1)
        lo = torch.minimum(a[rows], want[rows])
        assert bool(((a[rows] - want[rows]).abs() <= 2).all())
        assert bool(((cdf[rows, lo] - u[rows].double() * cdf[rows, -1]).abs() < 1e-6).all())
    assert int((a == 17).sum()) == 0 and int(a.min()) >= 0 and int(a.max()) <= NA - 1
    # small action counts take other per-lane widths
    for na in (5, 64, 100, 1000):
        lg = torch.randn(300, na, device="cuda", generator=g)
        uu = torch.rand(300, device="cuda", generator=g)
        got = A.sample_actions(lg, uu)
        c = torch.softmax(lg.double(), dim=1).cumsum(dim=1)
        ref = torch.searchsorted(c, (uu.double() * c[:, -1]).unsqueeze(1), right=True).squeeze(1).clamp_(max=na - 1)
        assert int((got != ref).sum()) <= 1


@pytest.mark.parametrize("shape", [(4096, 625, 640), (4096, 625, 625), (1000, 5, 5), (300, 100, 100), (513, 1000, 1000), (700, 256, 260), (257, 513, 516)],
                         ids=lambda s: "M%d_A%d_ld%d" % s)
def test_loss_grad_kernel_matches_autograd(shape):
    """uavagent_a2c_loss_grad against autograd of agent.a2c_losses (main.py:64-74).  Rows of ld floats: a multiple of 4 takes the float4
    kernel with the hardware's exp2 / log2 / rcp (the learner's own logits: 625 columns in rows of 640 with a zero tail, which must stay
    zero), anything else the dword kernel with the library's expf / logf."""
    torch = _torch()
    from drl_uav_cellularnet_amd import _agent_capi as A
    from drl_uav_cellularnet_amd.agent import a2c_losses

    M, NA, LD = shape
    g = torch.Generator(device="cuda").manual_seed(11)
    logits = (torch.randn(M, NA, device="cuda", generator=g) * 2).requires_grad_()
    v = torch.randn(M, 1, device="cuda", generator=g).requires_grad_()
    target = torch.randn(M, 1, device="cuda", generator=g)
    act = torch.randint(0, NA, (M,), device="cuda", generator=g)
    a_loss, c_loss = a2c_losses(torch.softmax(logits, dim=1), v, act, target, beta=0.001)
    (a_loss + c_loss).backward()
    pad = torch.zeros((M, LD), device="cuda")
    pad[:, :NA] = logits.detach()
    work = pad[:, :NA]                                  # rows contiguous, row stride LD
    dv = torch.empty(M, device="cuda")
    db = torch.empty(NA, device="cuda")
    loss = torch.zeros(3, dtype=torch.float64, device="cuda")
    A.a2c_loss_grad(work, v.detach().reshape(M).contiguous(), target.reshape(M).contiguous(), act, 0.001, dv, db, loss,
                    A.loss_grad_workspace(NA, "cuda"))
    scale = float(logits.grad.abs().max())
    torch.testing.assert_close(work, logits.grad, rtol=1e-4, atol=1e-5 * scale)
    torch.testing.assert_close(dv, v.grad.reshape(M), rtol=1e-5, atol=1e-9)
    torch.testing.assert_close(db, logits.grad.sum(dim=0), rtol=1e-4, atol=1e-5 * float(logits.grad.sum(dim=0).abs().max()) + 1e-9)
    np.testing.assert_allclose(loss.cpu().numpy()[:2], [float(a_loss), float(c_loss)], rtol=1e-5)
    np.testing.assert_allclose(float(loss[2]), float(v.grad.sum()), rtol=1e-4, atol=1e-7)
    assert float(pad[:, NA:].abs().max()) == 0.0 if LD > NA else True          # the zero tail the update's GEMMs read stays zero


def test_relu6_bwd_and_value_head_kernels():
    torch = _torch()
    from drl_uav_cellularnet_amd import _agent_capi as A

    M, H = 5000, 200
    g = torch.Generator(device="cuda").manual_seed(5)
    y = (torch.randn(M, H, device="cuda", generator=g) * 4).clamp_(0, 6)      # a relu6 output: zeros, interior, sixes
    dy = torch.randn(M, H, device="cuda", generator=g)
    ws = A.relu6_bwd_workspace(H, "cuda")
    cat = torch.full((M, 2 * H), 7.0, device="cuda")
    db = torch.empty(H, device="cuda")
    A.relu6_bwd(dy, y, cat[:, H:], 2 * H, db, ws)
    want = dy * ((y > 0) & (y < 6))
    assert torch.equal(cat[:, H:], want) and bool((cat[:, :H] == 7.0).all())
    torch.testing.assert_close(db, want.sum(dim=0), rtol=1e-4, atol=1e-4)
    # value head: v = y @ w3 + b3
    w3 = torch.randn(H, 1, device="cuda", generator=g) * 0.1
    b3 = torch.randn(1, device="cuda", generator=g)
    v = torch.empty(M, device="cuda")
    A.rowdot(y, w3, b3, v)
    torch.testing.assert_close(v, (y @ w3 + b3).reshape(M), rtol=1e-5, atol=1e-5)
    dv = torch.randn(M, device="cuda", generator=g)
    dx = torch.empty(M, H, device="cuda")
    dw3 = torch.empty(H, 1, device="cuda")
    A.relu6_bwd(None, y, dx, H, db, ws, dv=dv, w3=w3, dw3_out=dw3)
    want = (dv.unsqueeze(1) * w3.t()) * ((y > 0) & (y < 6))
    torch.testing.assert_close(dx, want, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(db, want.sum(dim=0), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(dw3, (y * dv.unsqueeze(1)).sum(dim=0).unsqueeze(1), rtol=1e-4, atol=1e-4)


def _rows_grad_case(torch, A, idx, H, S, two):
    M, K = idx.shape
    g = torch.randn(M, (2 if two else 1) * H, device="cuda", generator=torch.Generator(device="cuda").manual_seed(M + H))
    dw0 = torch.full((S, H), 9.0, device="cuda")
    dw1 = torch.full((S, H), 9.0, device="cuda") if two else None
    ws = A.rows_grad_workspace(M, K, g.shape[1], S, "cuda")
    A.rows_grad(idx, g, H, S, dw0, dw1, ws)
    ref0 = A._table_grad_aten(g[:, :H], idx, S)
    tol = dict(rtol=1e-5, atol=1e-5 * max(1.0, float(ref0.abs().max())))
    torch.testing.assert_close(dw0, ref0, **tol)
    if two:
        ref1 = A._table_grad_aten(g[:, H:], idx, S)
        torch.testing.assert_close(dw1, ref1, rtol=1e-5, atol=1e-5 * max(1.0, float(ref1.abs().max())))
    # float64 ground truth on a sample of rows, and bit-reproducibility
    g64 = g[:, :H].double()
    for r in torch.unique(idx[idx >= 0])[:5].tolist():
        cnt = (idx == r).sum(dim=1).double()
        want = (cnt.unsqueeze(1) * g64).sum(dim=0)
        # float32 accumulation of up to 120 000 terms: the error bound grows with the number of terms, not with the result
        tol64 = 2e-6 * float(cnt.sum()) ** 0.5 * float(g64.abs().max()) + 1e-5
        torch.testing.assert_close(dw0[r].double(), want, rtol=1e-4, atol=tol64)
    again = torch.empty_like(dw0)
    A.rows_grad(idx, g, H, S, again, torch.empty_like(dw0) if two else None, ws)
    assert torch.equal(again, dw0)


def test_rows_grad_matches_the_embedding_bag_backward():
    """d loss / d W1 by sort + segmented sums against ATen's embedding_bag backward: uniform rows, a few HOT rows whose runs
    cross many 512-pair chunks (every env's UAV on the same cell), 'no row' entries, one / two tables, small tables."""
    torch = _torch()
    from drl_uav_cellularnet_amd import _agent_capi as A

    gen = torch.Generator(device="cuda").manual_seed(1)
    S, H = 50000, 200
    idx = torch.randint(0, S, (4096, 24), device="cuda", generator=gen)
    _rows_grad_case(torch, A, idx, H, S, True)
    idx = torch.randint(0, S, (6000, 24), device="cuda", generator=gen)
    idx[:, 0] = 2525                                   # 6000 pairs of one row: 12 chunks
    idx[:, 1] = 49999
    idx[::2, 2] = 0
    idx[::7, 5] = -1
    idx[:3000, 6] = 31
    _rows_grad_case(torch, A, idx, H, S, True)
    idx = torch.full((40000, 3), 7, dtype=torch.int64, device="cuda")     # ONE row, 120000 pairs: 235 chunks (the gallop path)
    idx[-1, 2] = 8
    _rows_grad_case(torch, A, idx, 64, 100, False)
    _rows_grad_case(torch, A, torch.randint(0, 37, (1000, 5), device="cuda", generator=gen), 8, 37, True)
    _rows_grad_case(torch, A, torch.full((10, 4), -1, dtype=torch.int64, device="cuda"), 200, 50, True)   # nothing to add: zeros


def test_rows_grad_sort_once_sums_per_trunk_equals_the_single_call():
    """ABI 4: uavagent_rows_grad_sort + uavagent_rows_grad_sums_f32 against uavagent_rows_grad_f32, bit for bit -- both tables in one sum,
    and ONE sort (issued for the 2 x H layout, on another stream) serving a sum per trunk (what a rank does that exchanges the trunks'
    gradients separately)."""
    torch = _torch()
    from drl_uav_cellularnet_amd import _agent_capi as A

    gen = torch.Generator(device="cuda").manual_seed(21)
    S, H, M, K = 50000, 200, 5003, 24
    idx = torch.randint(0, S, (M, K), device="cuda", generator=gen)
    idx[:, 0] = 777
    idx[::5, 3] = -1
    g = torch.randn(M, 2 * H, device="cuda", generator=gen)
    ws = A.rows_grad_workspace(M, K, 2 * H, S, "cuda")
    want_a, want_c = torch.empty(S, H, device="cuda"), torch.empty(S, H, device="cuda")
    A.rows_grad(idx, g, H, S, want_a, want_c, ws)
    ws.zero_()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        A.rows_grad_sort(idx, 2 * H, S, ws)
    torch.cuda.current_stream().wait_stream(side)
    got_a, got_c = torch.full_like(want_a, 3.0), torch.full_like(want_c, 3.0)
    A.rows_grad_sums((M, K), g, H, S, got_a, got_c, ws)
    assert torch.equal(got_a, want_a) and torch.equal(got_c, want_c)
    ga, gc = g[:, :H].contiguous(), g[:, H:].contiguous()
    one_a, one_c = torch.full_like(want_a, 3.0), torch.full_like(want_c, 3.0)
    A.rows_grad_sums((M, K), gc, H, S, one_c, None, ws)             # the same sorted pairs, one table at a time
    A.rows_grad_sums((M, K), ga, H, S, one_a, None, ws)
    assert torch.equal(one_a, want_a) and torch.equal(one_c, want_c)
    with pytest.raises(A.UavAgentError):
        A.rows_grad_sort(idx, 2 * H, S, ws[:1024])                  # a workspace that cannot hold the pairs


def test_rmsprop_kernel_has_tf1_semantics():
    torch = _torch()
    from drl_uav_cellularnet_amd import _agent_capi as A

    n = 100003
    g = torch.Generator(device="cuda").manual_seed(2)
    w = torch.randn(n + 1, device="cuda", generator=g)[:n]
    w0 = w.clone()
    ms = torch.ones(n, device="cuda")
    grad = torch.randn(n, device="cuda", generator=g)
    ms_ref, w_ref = ms.clone(), w.clone()
    for it in range(3):
        A.rmsprop_tf1(w, ms, grad, 1e-4, g_scale=0.5)
        gs = grad * 0.5
        ms_ref = 0.9 * ms_ref + 0.1 * gs * gs                     # ms initialised to ones; epsilon inside the sqrt
        w_ref = w_ref - 1e-4 * gs / torch.sqrt(ms_ref + 1e-10)
    torch.testing.assert_close(ms, ms_ref, rtol=1e-6, atol=0)
    torch.testing.assert_close(w, w_ref, rtol=1e-6, atol=1e-7)
    assert not torch.equal(w, w0)


def test_nstep_returns_kernel():
    torch = _torch()
    from drl_uav_cellularnet_amd import _agent_capi as A
    from drl_uav_cellularnet_amd.agent import nstep_returns

    g = torch.Generator(device="cuda").manual_seed(4)
    rew = torch.randn(50, 777, device="cuda", generator=g)
    boot = torch.randn(777, device="cuda", generator=g)
    boot[::5] = 0.0                                                # episodes that ended: value_estimate = 0
    got = A.nstep_returns(rew, boot, 0.9)
    assert torch.equal(got, nstep_returns(rew, boot, 0.9))         # same float32 operations in the same order


def _twin_runners(torch, n_envs, T, **kw):
    from drl_uav_cellularnet_amd import BatchedMobiEnv
    from drl_uav_cellularnet_amd.agent import A2CRunner

    env1 = BatchedMobiEnv(n_envs, nBS=4, nUE=20, grid_n=100, max_step=kw.pop("max_step", 2000))
    env2 = env1.clone()
    r1 = A2CRunner(env1, rollout=T, **kw.pop("first", {}))
    r2 = A2CRunner(env2, rollout=T, **kw.pop("second", {}))
    return r1, r2


def _rel(got, want64):
    return float((got.double() - want64).abs().max() / want64.abs().max().clamp_min(1e-30))


# (M, K, N): the learner's own shapes + tile tails; aligned ones take the float4 kernel, the others the dword kernel
ROWS_SHAPES = [(1, 200, 200), (129, 200, 200), (4133, 200, 200), (1000, 640, 200), (777, 625, 200), (300, 37, 50), (513, 20, 208), (260, 204, 8),
               (40001, 200, 200), (33000, 640, 200), (35000, 40, 64)]       # > 32768 rows: the update's 128-row kernel


@pytest.mark.parametrize("shape", ROWS_SHAPES, ids=lambda s: "M%d_K%d_N%d" % s)
def test_gemm_rows_matches_torch_mm(shape):
    """uavagent_gemm_rows_f32 (float32 MFMA): x @ W, dy @ W^T, + bias / relu6, relu6-backward mask, column sums -- against float64
    products; tolerance 1e-5 of the largest output (O(1) data; torch.mm's own error on these shapes is 1e-6)."""
    torch = _torch()
    from drl_uav_cellularnet_amd import _agent_capi as A

    M, K, N = shape
    g = torch.Generator(device="cuda").manual_seed(M + K + N)
    rnd = lambda *s: torch.rand(s, device="cuda", generator=g) * 2.0 - 1.0
    x, w, wt, bias = rnd(M, K), rnd(K, N), rnd(N, K), rnd(N)
    h = (rnd(M, N) * 4.0 + 2.0).clamp_(0.0, 6.0)
    h[0, 0], h[M - 1, N - 1] = 0.0, 6.0                                 # exactly on the clamp: gradient 0 (strict inequalities)
    out = torch.full((M, N), float("nan"), device="cuda")
    A.gemm_rows(x, w, out)
    assert _rel(out, x.double() @ w.double()) < 1e-5
    A.gemm_rows(x, w, out, bias=bias, relu6=True)
    assert _rel(out, (x.double() @ w.double() + bias.double()).clamp(0.0, 6.0)) < 1e-5
    A.gemm_rows(x, wt, out, w_transposed=True)
    assert _rel(out, x.double() @ wt.double().t()) < 1e-5
    want = (x.double() @ wt.double().t()) * ((h > 0) & (h < 6)).double()
    A.gemm_rows(x, wt, out, w_transposed=True, relu6_mask_h=h)
    assert _rel(out, want) < 1e-5
    assert float(out[0, 0]) == 0.0 and float(out[M - 1, N - 1]) == 0.0
    aligned = (K % 4 == 0) and (N % 4 == 0)
    if aligned:                                                         # column sums (bias gradient) ride on the aligned kernels only
        cs, ws = torch.empty(N, device="cuda"), A.gemm_rows_workspace(M, "cuda")
        wide = torch.full((M, 2 * N + 8), float("nan"), device="cuda")     # C as a column slice of a wider buffer (gcat)
        A.gemm_rows(x, wt, wide[:, N + 8:], w_transposed=True, relu6_mask_h=h, colsum_out=cs, workspace=ws)
        assert torch.equal(wide[:, N + 8:], out) and bool(torch.isnan(wide[:, :N + 8]).all())
        assert float((cs.double() - want.sum(dim=0)).abs().max()) < 1e-5 * max(1.0, float(want.abs().sum(dim=0).max()))
        cs2 = torch.empty_like(cs)
        A.gemm_rows(x, wt, wide[:, N + 8:], w_transposed=True, relu6_mask_h=h, colsum_out=cs2, workspace=ws)
        assert torch.equal(cs, cs2)                                      # fixed summation order
    else:
        with pytest.raises(A.UavAgentError):
            A.gemm_rows(x, wt, out, w_transposed=True, colsum_out=torch.empty(N, device="cuda"), workspace=A.gemm_rows_workspace(M, "cuda"))
    if aligned:                                                         # x @ W through the transposed form: the rollout's layers
        A.gemm_rows(x, w.t().contiguous(), out, w_transposed=True, bias=bias, relu6=True)
        assert _rel(out, (x.double() @ w.double() + bias.double()).clamp(0.0, 6.0)) < 1e-5
    with pytest.raises(A.UavAgentError):
        A.gemm_rows(x, wt, out, w_transposed=True, bias=bias, relu6_mask_h=h)
    with pytest.raises(A.UavAgentError):
        A.gemm_rows(x, rnd(N + 1, K), out, w_transposed=True)


def test_gemm_rows_wide_output_in_slices():
    """The policy head forwards: [N, 200] x [200, 625] with the logits in rows of 640 -- the caller hands over W^T and the bias padded
    to 640 rows (zeros), N is cut into slices across workgroups, and the tail of every logits row comes out exactly zero (the update's
    GEMMs read it)."""
    torch = _torch()
    from drl_uav_cellularnet_amd import _agent_capi as A

    g = torch.Generator(device="cuda").manual_seed(3)
    rnd = lambda *s: torch.rand(s, device="cuda", generator=g) * 2.0 - 1.0
    for M in (8192, 100, 40000):
        x, w3t, b3 = rnd(M, 200), torch.zeros((640, 200), device="cuda"), torch.zeros(640, device="cuda")
        w3t[:625], b3[:625] = rnd(625, 200), rnd(625)
        out = torch.full((M, 640), float("nan"), device="cuda")
        A.gemm_rows(x, w3t, out, w_transposed=True, bias=b3)
        assert _rel(out[:, :625], x.double() @ w3t[:625].double().t() + b3[:625].double()) < 1e-5
        assert float(out[:, 625:].abs().max()) == 0.0
    with pytest.raises(A.UavAgentError):
        A.gemm_rows(x, rnd(200, 640), out)                                 # x @ W with W [K, N]: N > 208 is not built


@pytest.mark.parametrize("shape", [(300, 40, 320), (300, 36, 320), (33000, 40, 320), (33000, 36, 320), (33001, 204, 200), (700, 44, 96),
                                   (33001, 200, 200), (40000, 200, 132), (33010, 200, 208), (32800, 200, 116), (36000, 200, 112)],
                         ids=lambda s: "M%d_K%d_N%d" % s)
def test_gemm_rows_every_tile_shape_and_epilogue(shape):
    """The w_transposed kernels come in tile shapes chosen by M (64- or 128-row workgroups), N (7-, 10- or 13-block column slices) and
    K (LDS-DMA ring when K % 40 == 0, register-staged otherwise; K = 200 with more than 32768 rows and 112 < N <= 208: W^T resident in LDS,
    persistent workgroups, columns split 112 + the rest): every one of them with every epilogue against float64."""
    torch = _torch()
    from drl_uav_cellularnet_amd import _agent_capi as A

    M, K, N = shape
    g = torch.Generator(device="cuda").manual_seed(M * 7 + K + N)
    rnd = lambda *s: torch.rand(s, device="cuda", generator=g) * 2.0 - 1.0
    x, wt, bias = rnd(M, K), rnd(N, K), rnd(N)
    h = (rnd(M, N) * 4.0 + 2.0).clamp_(0.0, 6.0)
    want = x.double() @ wt.double().t()
    out = torch.full((M, N), float("nan"), device="cuda")
    A.gemm_rows(x, wt, out, w_transposed=True)
    assert _rel(out, want) < 1e-5
    A.gemm_rows(x, wt, out, w_transposed=True, bias=bias, relu6=True)
    assert _rel(out, (want + bias.double()).clamp(0.0, 6.0)) < 1e-5
    A.gemm_rows(x, wt, out, w_transposed=True, bias=bias)
    assert _rel(out, want + bias.double()) < 1e-5
    A.gemm_rows(x, wt, out, w_transposed=True, relu6_mask_h=h)
    assert _rel(out, want * ((h > 0) & (h < 6)).double()) < 1e-5
    if N <= 208:
        cs, ws = torch.empty(N, device="cuda"), A.gemm_rows_workspace(M, "cuda")
        A.gemm_rows(x, wt, out, w_transposed=True, relu6_mask_h=h, colsum_out=cs, workspace=ws)
        wm = want * ((h > 0) & (h < 6)).double()
        assert float((cs.double() - wm.sum(dim=0)).abs().max()) < 1e-5 * max(1.0, float(wm.abs().sum(dim=0).max()))


@pytest.mark.parametrize("tile", ["auto", "1", "2"], ids=["tile_auto", "16_row_tiles", "32_row_tiles"])
@pytest.mark.parametrize("n_rows", [8192, 33, 1000, 4128])
def test_actor_head_kernel_equals_the_three_launches(n_rows, tile, monkeypatch):
    """uavagent_actor_head_f32 (layer 2 + policy head + inverse-CDF draw in one kernel, a workgroup per 32 rows) against
    uavagent_gemm_rows_f32 twice + uavagent_sample_actions: h2, logits (incl. the zero tail) and actions bit for bit, and against float64."""
    torch = _torch()
    from drl_uav_cellularnet_amd import _agent_capi as A

    if tile != "auto":                                  # both workgroup shapes at every size (auto: 16-row tiles up to 24 rows per CU)
        monkeypatch.setenv("UAVAGENT_HEAD_RB", tile)
    g = torch.Generator(device="cuda").manual_seed(n_rows)
    rnd = lambda *s: torch.rand(s, device="cuda", generator=g) * 2.0 - 1.0
    H, NA = 200, 625
    h1 = (rnd(n_rows, H) * 4.0 + 2.0).clamp_(0.0, 6.0)
    w2, b2, w3, b3 = rnd(H, H) * 0.2, rnd(H), rnd(H, NA) * 0.3, rnd(NA)
    w2t = w2.t().contiguous()
    w3t, b3p = torch.zeros((640, H), device="cuda"), torch.zeros(640, device="cuda")
    w3t[:NA], b3p[:NA] = w3.t(), b3
    u = torch.rand(n_rows, device="cuda", generator=g)
    u[0] = 0.0
    h2a, lga, acta = torch.full((n_rows, H), float("nan"), device="cuda"), torch.full((n_rows, 640), float("nan"), device="cuda"), torch.full((n_rows,), -1, dtype=torch.int64, device="cuda")
    h2b, lgb = torch.empty_like(h2a), torch.empty_like(lga)
    A.actor_head(h1, w2t, b2, w3t, b3p, u, NA, h2a, lga, acta)
    A.gemm_rows(h1, w2t, h2b, w_transposed=True, bias=b2, relu6=True)
    A.gemm_rows(h2b, w3t, lgb, w_transposed=True, bias=b3p)
    actb = A.sample_actions(lgb[:, :NA], u)
    assert torch.equal(h2a, h2b) and torch.equal(lga, lgb) and torch.equal(acta, actb)
    assert float(lga[:, NA:].abs().max()) == 0.0
    want_h2 = (h1.double() @ w2.double() + b2.double()).clamp(0.0, 6.0)
    assert _rel(h2a, want_h2) < 1e-5 and _rel(lga[:, :NA], want_h2 @ w3.double() + b3.double()) < 1e-5
    with pytest.raises(A.UavAgentError):
        A.actor_head(h1, w2t, b2, w3t, b3p, u, 100, h2a, lga, acta)          # built for the reference's 625 actions (577..640)


# (M, I, J, ldb): J <= 208 runs plan 13, wider plan 20; ldb > J = a column slice of a padded buffer (the learner's logits: 625 of 640)
TN_SHAPES = [(1, 200, 200, 200), (31, 200, 200, 200), (5000, 200, 200, 200), (40037, 200, 200, 200), (3000, 200, 625, 640), (2999, 200, 625, 625),
             (700, 200, 640, 640), (1500, 64, 100, 100), (900, 200, 321, 324), (1200, 8, 5, 5)]


@pytest.mark.parametrize("shape", TN_SHAPES, ids=lambda s: "M%d_I%d_J%d_ld%d" % s)
def test_gemm_tn_matches_torch_mm_and_is_bit_reproducible(shape):
    """uavagent_gemm_tn_f32: x^T dy with the bias gradient (column sums of dy) from the ones column, split over the CUs and reduced
    in a fixed order."""
    torch = _torch()
    from drl_uav_cellularnet_amd import _agent_capi as A

    M, I, J, ldb = shape
    g = torch.Generator(device="cuda").manual_seed(M + I + J)
    x = torch.rand((M, I), device="cuda", generator=g) * 2.0 - 1.0
    full = torch.zeros((M, ldb), device="cuda")
    full[:, :J] = torch.rand((M, J), device="cuda", generator=g) * 2.0 - 1.0
    y = full[:, :J]
    ws = A.gemm_tn_workspace(M, J, "cuda")
    out, db = torch.full((I, J), float("nan"), device="cuda"), torch.full((J,), float("nan"), device="cuda")
    A.gemm_tn(x, y, out, ws, dbias_out=db)
    want = x.double().t() @ y.double()
    scale = max(1.0, float(want.abs().max()))
    assert float((out.double() - want).abs().max()) < 1e-5 * scale
    assert float((db.double() - y.double().sum(dim=0)).abs().max()) < 1e-5 * max(1.0, float(y.double().abs().sum(dim=0).max()))
    out2, db2 = torch.empty_like(out), torch.empty_like(db)
    A.gemm_tn(x, y, out2, ws, dbias_out=db2)
    assert torch.equal(out, out2) and torch.equal(db, db2)
    A.gemm_tn(x, y, out2, ws)                                            # without the bias output: same product
    assert torch.equal(out, out2)
    with pytest.raises(A.UavAgentError):
        A.gemm_tn(x, y, out, torch.empty(16, dtype=torch.uint8, device="cuda"))
    with pytest.raises(A.UavAgentError):
        A.gemm_tn(x, y, torch.empty((I, J + 1), device="cuda"), ws)


def test_update_with_hip_gemms_matches_the_update_with_torch_gemms():
    """The same fused update with the dense layers through libuavagent's MFMA kernels (relu6 masks and bias gradients fused) and
    through torch.mm + separate relu6-backward passes: every gradient within float32 summation-order noise, and the HIP form is
    bit-reproducible."""
    torch = _torch()
    r1, r2 = _twin_runners(torch, 700, 5, first=dict(collect_launch="eager"), second=dict(collect_launch="eager"))
    assert r1.hip_gemms and r2.hip_gemms               # (the twins collect with the same kernels: identical samples; only r2's UPDATE
    for it in range(2):                                #  runs on torch GEMMs)
        b1, b2 = r1.collect(), r2.collect()
        for x, y in zip(b1, b2):
            assert torch.equal(x, y)
        if it == 1:
            r1._fwd_valid = r2._fwd_valid = False      # round 2: both recompute the forward pass (the HIP form with its own GEMM)
        r2.hip_gemms = False
        s1, s2 = r1.update(*b1), r2.update(*b2)
        r2.hip_gemms = True
        assert s1["hip_gemms"] and not s2["hip_gemms"]
        np.testing.assert_allclose([s1["a_loss"], s1["c_loss"]], [s2["a_loss"], s2["c_loss"]], rtol=1e-5)
        for k in r1.flat.gv:
            g1, g2 = r1.flat.gv[k], r2.flat.gv[k]
            torch.testing.assert_close(g1, g2, rtol=1e-4, atol=1e-5 * float(g2.abs().max()) + 1e-12, msg=lambda m: "%s: %s" % (k, m))
        r2.flat.w.copy_(r1.flat.w)                     # keep the twins in lockstep for round 2
        r2.flat.ms.copy_(r1.flat.ms)
    # bit-reproducibility proper: two fresh twins, both HIP
    r3, r4 = _twin_runners(torch, 300, 4, first=dict(hip_gemms=True), second=dict(hip_gemms=True))
    for _ in range(2):
        r3.train_rollout(), r4.train_rollout()
        assert torch.equal(r3.flat.g, r4.flat.g) and torch.equal(r3.flat.w, r4.flat.w)


def test_fused_update_matches_the_autograd_update():
    torch = _torch()
    r1, r2 = _twin_runners(torch, 600, 7, first=dict(fused_update=True, collect_launch="eager"),
                           second=dict(fused_update=False, collect_launch="eager", update_chunk=1500))
    assert torch.equal(r1.flat.w, r2.flat.w)
    for it in range(2):
        b1, b2 = r1.collect(), r2.collect()
        for x, y in zip(b1, b2):
            assert torch.equal(x, y)
        if it == 1:
            r1._fwd_valid = False          # round 2: the update recomputes its forward pass instead of reusing the rollout's
        s1, s2 = r1.update(*b1), r2.update(*b2)
        assert s1["forward_reused"] == (it == 0)
        np.testing.assert_allclose([s1["a_loss"], s1["c_loss"]], [s2["a_loss"], s2["c_loss"]], rtol=1e-5)
        for k in r1.flat.gv:
            g1, g2 = r1.flat.gv[k], r2.flat.gv[k]
            torch.testing.assert_close(g1, g2, rtol=1e-4, atol=1e-5 * float(g2.abs().max()) + 1e-12, msg=lambda m: "%s: %s" % (k, m))
        assert s1["grad_elems"] == s2["grad_elems"] == 20206626
        for k in r1.flat.gv:              # (views only: the fused step also decays the alignment padding's accumulator, harmlessly)
            torch.testing.assert_close(r1._ms_view(k), r2._ms_view(k), rtol=1e-4, atol=1e-9)
        torch.testing.assert_close(r1.flat.w, r2.flat.w, rtol=0, atol=2e-6)        # lr = 1e-4: steps of <= 1e-4 / sqrt(0.9)
        r2.flat.w.copy_(r1.flat.w)                                                  # keep the twins in lockstep for round 2
        r2.flat.ms.copy_(r1.flat.ms)


@pytest.mark.parametrize("first,second", [(dict(collect_launch="graph", pipeline_halves=False), dict(collect_launch="eager", pipeline_halves=False)),
                                          (dict(collect_launch="graph", pipeline_halves="force"), dict(collect_launch="eager", pipeline_halves=False)),
                                          (dict(collect_launch="eager", pipeline_halves="force"), dict(collect_launch="graph", pipeline_halves="force")),
                                          (dict(collect_launch="graph", pipeline_halves=False, persistent_rollout=True), dict(collect_launch="eager", pipeline_halves=False)),
                                          (dict(collect_launch="eager", pipeline_halves=False, persistent_rollout=True), dict(collect_launch="graph", pipeline_halves="force"))],
                         ids=["graph_vs_eager", "pipelined_graph_vs_plain_eager", "pipelined_eager_vs_pipelined_graph", "persistent_graph_vs_plain_eager",
                              "persistent_eager_vs_pipelined_graph"])
def test_graph_captured_rollout_equals_the_eager_rollout(first, second):
    """The rollout as a replayed hipGraph against the eager loop; and with the batch cut in two halves that ping-pong on two streams
    (A2CRunner(pipeline_halves=...): gather + env step of one half beside the actor head of the other) against the unsplit loop:
    one uniform per env and step, the same kernels on the same rows -- indices, actions, rewards, env state and, after the update,
    parameters bit for bit."""
    torch = _torch()
    first.setdefault("persistent_rollout", False); second.setdefault("persistent_rollout", False)
    r1, r2 = _twin_runners(torch, 512, 6, max_step=15, first=dict(first), second=dict(second))
    assert (r1._halves is not None) == (first["pipeline_halves"] == "force") and (r2._halves is not None) == (second["pipeline_halves"] == "force")
    assert r1._persistent == (first["persistent_rollout"] is True) and not r2._persistent       # (the two persistent launches instead of 3 T: see A2CRunner)
    if r1._halves is not None:
        assert r1._halves == ((0, 256), (256, 512))                # cut on a multiple of the head's 16-row tiles (inside an env wavefront: 256 = 85 x 3 + 1)
    for it in range(4):                                            # crosses done + masked reset (MAXSTEP 15 inside rollout 3)
        b1, b2 = r1.collect(), r2.collect()
        assert r1._persistent == (first["persistent_rollout"] is True)         # (it did not fall back)
        for name, x, y in zip(("idx", "act", "rew", "boot"), b1, b2):
            assert torch.equal(x, y), "%s differs in rollout %d" % (name, it)
        assert np.array_equal(r1.env.get_state(), r2.env.get_state())
        assert torch.equal(r1.idx, r2.idx)
        r1.update(*b1)
        r2.update(*b2)
        assert torch.equal(r1.flat.w, r2.flat.w)                   # deterministic kernels: the twins stay bit-identical
    assert r1.running_r is not None and r1.running_r == r2.running_r


@pytest.mark.parametrize("n_envs", [1000, 4100], ids=lambda n: "%d_envs" % n)
def test_persistent_rollout_with_a_ragged_last_block_equals_the_per_step_rollout(n_envs):
    """Batches that are no multiple of the 16-env blocks (1000 = 62 blocks + 8 envs: the last pair has ONE, ragged block; 4100 = 256 + 1 pairs: more
    pairs than CUs, the last one claimed by whichever workgroup finishes first): persistent rollout against the per-step launches, bit for bit."""
    torch = _torch()
    r1, r2 = _twin_runners(torch, n_envs, 5, max_step=9, first=dict(collect_launch="graph", persistent_rollout=True, pipeline_halves=False),
                           second=dict(collect_launch="eager", persistent_rollout=False, pipeline_halves=False))
    assert r1._persistent and not r2._persistent
    for it in range(3):
        b1, b2 = r1.collect(), r2.collect()
        assert r1._persistent
        for name, x, y in zip(("idx", "act", "rew", "boot"), b1, b2):
            assert torch.equal(x, y), "%s differs in rollout %d" % (name, it)
        assert np.array_equal(r1.env.get_state(), r2.env.get_state())
        r1.update(*b1); r2.update(*b2)
        assert torch.equal(r1.flat.w, r2.flat.w)


@pytest.mark.parametrize("launch", ["eager", "graph"])
def test_persistent_rollout_falls_back_when_its_kernels_cannot_run_side_by_side(launch, monkeypatch):
    """The two persistent rollout kernels wait for each other, so they must run at the same time.  With both in ONE stream (test hook) the
    first waits in vain: every wait is bounded, the trial on a clone of the env state sees the error words, and the runner goes on with the
    per-step launches -- same results as a runner that never tried, nothing hangs, the env handle is usable."""
    torch = _torch()
    import warnings

    monkeypatch.setenv("UAVAGENT_PERSIST_SAME_STREAM", "1")
    monkeypatch.setenv("UAVAGENT_GATE_SPIN_US", "30000")           # 30 ms
    monkeypatch.setenv("UAVENV_HANDOFF_SPIN_US", "30000")          # (read in uavenv_create)
    r1, r2 = _twin_runners(torch, 256, 5, first=dict(collect_launch=launch, persistent_rollout=True, pipeline_halves=False),
                           second=dict(collect_launch="eager", persistent_rollout=False, pipeline_halves=False))
    assert r1._persistent
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        b1 = r1.collect()
    assert not r1._persistent and any("side by side" in str(x.message) for x in w)
    b2 = r2.collect()
    for name, x, y in zip(("idx", "act", "rew", "boot"), b1, b2):
        assert torch.equal(x, y), name
    assert np.array_equal(r1.env.get_state(), r2.env.get_state()) and r1.env.device_error() == 0
    r1.update(*b1); r2.update(*b2)
    b1, b2 = r1.collect(), r2.collect()
    assert torch.equal(b1[1], b2[1]) and torch.equal(r1.flat.w, r2.flat.w)


def test_persistent_rollout_recovers_when_a_kernel_gives_up_in_the_middle_of_a_run(monkeypatch):
    """After the trial has shown the two kernels side by side, a later rollout's wait still times out (here: the test hook moves both kernels
    into one stream).  collect() restores the state snapshot the rollout started from, switches to the per-step launches for good and collects the
    rollout again: the caller sees a warning and the results of an undisturbed run."""
    torch = _torch()
    import warnings

    monkeypatch.setenv("UAVENV_HANDOFF_SPIN_US", "30000")          # (read in uavenv_create)
    r1, r2 = _twin_runners(torch, 256, 5, max_step=12, first=dict(collect_launch="eager", persistent_rollout=True, pipeline_halves=False),
                           second=dict(collect_launch="eager", persistent_rollout=False, pipeline_halves=False))
    for it in range(4):
        if it == 2:
            r1._persist_same_stream, r1._gate_spin_us = True, 30000
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            b1 = r1.collect()
        b2 = r2.collect()
        assert r1._persistent == (it < 2) and (it != 2 or any("gave up" in str(x.message) for x in w))
        for name, x, y in zip(("idx", "act", "rew", "boot"), b1, b2):
            assert torch.equal(x, y), "%s differs in rollout %d" % (name, it)
        assert np.array_equal(r1.env.get_state(), r2.env.get_state()) and r1.env.device_error() == 0
        r1.update(*b1); r2.update(*b2)
        assert torch.equal(r1.flat.w, r2.flat.w)
    assert r1.running_r == r2.running_r


def test_fused_update_is_bit_identical_with_and_without_forward_reuse():
    """update_fused computes the critic's layer 2 with ONE arithmetic (W2^T through the k-contiguous kernels) whether it reuses the
    rollout's forward pass or recomputes it (first update after load_state_dict, external buffers): same parameters bit for bit
    (the two branches used kernels with different k orders)."""
    torch = _torch()
    r1, r2 = _twin_runners(torch, 700, 5, first=dict(collect_launch="eager"), second=dict(collect_launch="eager"))
    for it in range(2):
        b1, b2 = r1.collect(), r2.collect()
        r2._fwd_valid = False                          # r2 recomputes the forward pass with the same weights
        s1, s2 = r1.update(*b1), r2.update(*b2)
        assert s1["forward_reused"] and not s2["forward_reused"]
        assert torch.equal(r1.flat.g, r2.flat.g) and torch.equal(r1.flat.w, r2.flat.w) and torch.equal(r1.flat.ms, r2.flat.ms)


def test_gemm_wrappers_validate_their_operands():
    """gemm_rows / gemm_tn / actor_head refuse short or mistyped side outputs instead of letting a kernel overrun them."""
    torch = _torch()
    from drl_uav_cellularnet_amd import _agent_capi as A

    a = torch.randn(256, 200, device="cuda")
    w = torch.randn(200, 200, device="cuda")
    out = torch.empty(256, 200, device="cuda")
    ws = A.gemm_rows_workspace(256, "cuda")
    with pytest.raises(A.UavAgentError):               # column sums shorter than N
        A.gemm_rows(a, w, out, w_transposed=True, colsum_out=torch.empty(100, device="cuda"), workspace=ws)
    with pytest.raises(A.UavAgentError):               # wrong dtype
        A.gemm_rows(a, w, out, w_transposed=True, colsum_out=torch.empty(200, device="cuda", dtype=torch.float64), workspace=ws)
    with pytest.raises(A.UavAgentError):               # bias of the wrong length
        A.gemm_rows(a, w, out, w_transposed=True, bias=torch.empty(199, device="cuda"))
    with pytest.raises(A.UavAgentError):               # an operand on the host
        A.gemm_rows(a, w.cpu(), out, w_transposed=True)
    with pytest.raises(A.UavAgentError):               # mask of another shape
        A.gemm_rows(a, w, out, w_transposed=True, relu6_mask_h=torch.empty(255, 200, device="cuda"))
    cs = torch.empty(200, device="cuda")
    A.gemm_rows(a, w, out, w_transposed=True, colsum_out=cs, workspace=ws)       # the valid call still runs
    torch.testing.assert_close(cs, out.sum(dim=0), rtol=1e-4, atol=1e-3)
    with pytest.raises(A.UavAgentError):
        A.gemm_tn(a, out, torch.empty(200, 200, device="cuda"), None)


def test_gemm_tuning_is_frozen_after_the_first_rollout_and_update():
    """A2CRunner(tune_gemms=True): TunableOp may tune while the first rollout + update run eagerly, never afterwards."""
    torch = _torch()
    from drl_uav_cellularnet_amd import BatchedMobiEnv, agent

    env = BatchedMobiEnv(256, nBS=4, nUE=20, grid_n=100, groups=[5, 5, 5, 5])
    r = agent.A2CRunner(env, rollout=3, tune_gemms=True)
    if not r.gemm_tuning:
        pytest.skip("torch.cuda.tunable unavailable")
    import torch.cuda.tunable as tun

    try:
        r.train_rollout()
        assert not agent.gemm_tuning_is_active()
        assert tun.is_enabled()                        # the picks stay in use
        r.train_rollout()
        assert not agent.gemm_tuning_is_active()
    finally:
        tun.enable(False)                              # (the other tests of this process run on the library defaults)
