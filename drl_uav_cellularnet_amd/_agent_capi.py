"""ctypes binding of libuavagent.so (include/uavagent.h, ABI 4) and the autograd wrapper agent.py uses for CUDA tensors.

The plain PyTorch forms stay in agent.py as the reference implementations and the CPU path.  Here: thin launch wrappers (no
allocation beyond outputs, current torch stream) for the first layer, index construction, action sampling and the pieces of
the fused update.  A CUDA tensor with no library is an error, not a silent PyTorch fallback.
"""
import ctypes as C
import os

import torch

from . import build as _build

EXPORTS = ("uavagent_abi_version", "uavagent_last_error", "uavagent_sparse_rows_sum_f32", "uavagent_first_layer_f32",
           "uavagent_first_layer_from_obs_f32",
           "uavagent_obs_indices", "uavagent_sample_actions", "uavagent_loss_grad_workspace_bytes", "uavagent_a2c_loss_grad",
           "uavagent_relu6_bwd_workspace_bytes", "uavagent_relu6_bwd", "uavagent_rowdot_f32",
           "uavagent_rows_grad_workspace_bytes", "uavagent_rows_grad_f32", "uavagent_rows_grad_sort", "uavagent_rows_grad_sums_f32", "uavagent_nstep_returns_f32", "uavagent_rmsprop_tf1",
           "uavagent_gemm_rows_f32", "uavagent_gemm_rows_workspace_bytes", "uavagent_gemm_tn_workspace_bytes", "uavagent_gemm_tn_f32",
           "uavagent_debug_tn_plan_check", "uavagent_actor_head_f32", "uavagent_actor_head_gated_f32", "uavagent_gate_prepare",
           "uavagent_device_error", "uavagent_device_error_clear")
ABI_VERSION = 5

_lib = None
_P, _I64, _I32, _F = C.c_void_p, C.c_int64, C.c_int32, C.c_float


class UavAgentError(RuntimeError):
    pass


def lib_path():
    return os.environ.get("UAVAGENT_LIB") or _build.AGENT_LIB


# ---- per-launch timing (bench.py: the learner's roofline figures) -----------------------------------------------------------------
# profile_begin() makes load() hand out a proxy that brackets every launch entry point with HIP events on the current stream;
# profile_end() synchronises and returns {key: [milliseconds per call]}.  Meant for EAGER passes (a replayed hipGraph makes no calls);
# the kernels of one stream run one after the other, so an event pair measures the launch it brackets.  Off (the default): load()
# returns the plain CDLL, nothing is added to any call.
_prof = None
_PROF_KEYS = {
    # name -> indices of the integer arguments that tell two uses of one entry point apart
    "uavagent_gemm_rows_f32": lambda a: "M=%d,K=%d,N=%d%s" % (a[5], a[6], a[7], ",relu6_mask" if a[10] else (",bias" if a[8] else "")),
    "uavagent_gemm_tn_f32": lambda a: "M=%d,I=%d,J=%d" % (a[2], a[3], a[4]),
    "uavagent_actor_head_f32": lambda a: "rows=%d" % a[6],
    "uavagent_first_layer_from_obs_f32": lambda a: "rows=%d" % a[9],
    "uavagent_sparse_rows_sum_f32": lambda a: "rows=%d" % a[7],
    "uavagent_rows_grad_sums_f32": lambda a: "M=%d,K=%d" % (a[1], a[2]),
    "uavagent_rows_grad_f32": lambda a: "M=%d,K=%d" % (a[2], a[3]),
    "uavagent_a2c_loss_grad": lambda a: "M=%d,A=%d" % (a[5], a[6]),
}


class _ProfiledLib:
    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.endswith(("_f32", "_grad", "_bwd", "_tf1", "_sort", "_indices", "_actions")) or name.endswith("_bytes"):
            return fn

        def timed(*args):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            keyf = _PROF_KEYS.get(name)
            key = name + ("[%s]" % keyf(args) if keyf else "")
            if _prof is not None:
                _prof.setdefault(key, []).append((e0, e1))
            return rc
        return timed


def profile_begin():
    global _prof
    load()
    _prof = {}


def profile_end():
    """-> {"entry point[shape]": [ms, ...]} for every launch since profile_begin() (synchronises the device)."""
    global _prof
    torch.cuda.synchronize()
    out = {k: [a.elapsed_time(b) for a, b in v] for k, v in (_prof or {}).items()}
    _prof = None
    return out


def load():
    global _lib
    if _lib is not None:
        return _ProfiledLib(_lib) if _prof is not None else _lib
    path = lib_path()
    if not os.path.isfile(path):
        raise UavAgentError("%s not found: run `python -m drl_uav_cellularnet_amd.build` (there is no fallback for CUDA "
                            "tensors)" % path)
    lib = C.CDLL(path)
    lib.uavagent_abi_version.restype = C.c_int
    lib.uavagent_last_error.restype = C.c_char_p
    sig = {
        "uavagent_sparse_rows_sum_f32": [_P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _I64, _P],
        "uavagent_first_layer_f32": [_P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _I64, _I32, _P],
        "uavagent_obs_indices": [_P, _P, _P, _I64, _I32, _I32, _I32, _P, _P],
        "uavagent_first_layer_from_obs_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I64, _I32, _P, _P],
        "uavagent_sample_actions": [_P, _I64, _P, _I64, _I32, _P, _P, _P],
        "uavagent_a2c_loss_grad": [_P, _I64, _P, _P, _P, _I64, _I32, _F, _P, _P, _P, _P, _P],
        "uavagent_relu6_bwd": [_P, _P, _P, _P, _I64, _I32, _P, _I64, _P, _P, _P, _P],
        "uavagent_rowdot_f32": [_P, _P, _P, _I64, _I32, _P, _P],
        "uavagent_rows_grad_f32": [_P, _P, _I64, _I32, _I32, _I32, _I64, _P, _P, _P, C.c_size_t, _P],
        "uavagent_rows_grad_sort": [_P, _I64, _I32, _I32, _I64, _P, C.c_size_t, _P],
        "uavagent_rows_grad_sums_f32": [_P, _I64, _I32, _I32, _I32, _I64, _P, _P, _P, C.c_size_t, _P],
        "uavagent_nstep_returns_f32": [_P, _P, _I64, _I32, _F, _P, _P],
        "uavagent_rmsprop_tf1": [_P, _P, _P, _I64, _F, _F, _F, _F, _P],
        "uavagent_gemm_rows_f32": [_P, _I64, _P, _I64, _I32, _I64, _I32, _I32, _P, _I32, _P, _I64, _P, _I64, _P, _P, C.c_size_t, _P],
        "uavagent_gemm_tn_f32": [_P, _P, _I64, _I32, _I32, _I64, _P, _I64, _P, _P, C.c_size_t, _P],
        "uavagent_actor_head_f32": [_P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _P, _P, _I64, _P, _P],
        "uavagent_actor_head_gated_f32": [_P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _P, _P, _I64, _P, _P, _P, _P, C.c_uint32, _P],
        "uavagent_gate_prepare": [],
        "uavagent_device_error": [C.POINTER(C.c_uint32)],
        "uavagent_device_error_clear": [],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = args
    lib.uavagent_loss_grad_workspace_bytes.restype = C.c_size_t
    lib.uavagent_loss_grad_workspace_bytes.argtypes = [_I32]
    lib.uavagent_relu6_bwd_workspace_bytes.restype = C.c_size_t
    lib.uavagent_relu6_bwd_workspace_bytes.argtypes = [_I32]
    lib.uavagent_rows_grad_workspace_bytes.restype = C.c_size_t
    lib.uavagent_rows_grad_workspace_bytes.argtypes = [_I64, _I32, _I32, _I64]
    lib.uavagent_gemm_tn_workspace_bytes.restype = C.c_size_t
    lib.uavagent_gemm_tn_workspace_bytes.argtypes = [_I64, _I32]
    lib.uavagent_debug_tn_plan_check.restype = C.c_int
    lib.uavagent_debug_tn_plan_check.argtypes = [_I32]
    lib.uavagent_gemm_rows_workspace_bytes.restype = C.c_size_t
    lib.uavagent_gemm_rows_workspace_bytes.argtypes = [_I64]
    if lib.uavagent_abi_version() != ABI_VERSION:
        raise UavAgentError("libuavagent.so ABI version mismatch")
    _lib = lib
    return lib


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _check(rc, what):
    if rc != 0:
        raise UavAgentError("%s: %s" % (what, load().uavagent_last_error().decode()))


def _same_device(what, *tensors):
    """Every operand of one launch lives on one device (the launch goes to that device's current stream)."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise UavAgentError("%s: every operand must be a CUDA tensor" % what)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise UavAgentError("%s: operands on different devices (%s and %s)" % (what, dev, t.device))


def _f32c(t, what):
    if t is not None and (t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous()):
        raise UavAgentError("%s must be a contiguous float32 CUDA tensor" % what)


def sparse_rows_sum(idx, w_a, b_a, w_c=None, b_c=None, relu6=False, out_a=None, out_c=None):
    """Raw launch, no autograd: returns out_a or (out_a, out_c).  idx int64 [M, K]; w_* f32 [S, H]; b_* f32 [H] or None."""
    lib = load()
    if not (idx.is_cuda and w_a.is_cuda):
        raise UavAgentError("sparse_rows_sum needs CUDA tensors")
    if idx.dtype != torch.int64 or idx.dim() != 2:
        raise UavAgentError("idx must be int64 [M, K]")
    tensors = [t for t in (w_a, b_a, w_c, b_c) if t is not None]
    if any(t.dtype != torch.float32 or t.device != idx.device for t in tensors):
        raise UavAgentError("tables and biases must be float32 on the device of idx")
    if w_c is not None and w_c.shape != w_a.shape:
        raise UavAgentError("actor and critic tables must have the same shape")
    idx = idx.contiguous()
    w_a = w_a.contiguous()
    w_c = None if w_c is None else w_c.contiguous()
    b_a = None if b_a is None else b_a.contiguous()
    b_c = None if b_c is None else b_c.contiguous()
    (M, K), (S, H) = idx.shape, w_a.shape
    if out_a is None:
        out_a = torch.empty((M, H), dtype=torch.float32, device=idx.device)
    if w_c is not None and out_c is None:
        out_c = torch.empty((M, H), dtype=torch.float32, device=idx.device)
    for o in (out_a, out_c):
        if o is not None and (o.shape != (M, H) or o.dtype != torch.float32 or not o.is_contiguous()):
            raise UavAgentError("outputs must be contiguous float32 [M, H]")
    with torch.cuda.device(idx.device):
        rc = lib.uavagent_first_layer_f32(_ptr(w_a), _ptr(b_a), _ptr(out_a), _ptr(w_c), _ptr(b_c), _ptr(out_c), _ptr(idx),
                                          M, K, H, S, 1 if relu6 else 0, _stream(idx.device))
    _check(rc, "uavagent_first_layer_f32")
    return out_a if w_c is None else (out_a, out_c)


def first_layer_from_obs(obs, grid_n, w_a, b_a, w_c, b_c, out_a, out_c, idx_out=None, relu6=True):
    """obs_indices + sparse_rows_sum in one launch (the rollout loop): obs = env.observation(); the index list of every env is built in
    the kernel, stored to idx_out [N, B + U] (or not, when None) and summed.  Same bits as the two separate launches."""
    ue, bs, srv = obs["ue_xy"], obs["bs_xy"], obs["serving"]
    if ue.dtype != torch.int16 or bs.dtype != torch.int32 or srv.dtype != torch.int8:
        raise UavAgentError("first_layer_from_obs needs the env's compact observation dtypes (int16 / int32 / int8)")
    if not (ue.is_contiguous() and bs.is_contiguous() and srv.is_contiguous()):
        raise UavAgentError("first_layer_from_obs needs contiguous observation arrays")
    N, U, B = ue.shape[0], ue.shape[1], bs.shape[1]
    S, H = w_a.shape
    for t in (w_a, b_a, w_c, b_c, out_a, out_c):
        if t is not None and (t.dtype != torch.float32 or t.device != ue.device or not t.is_contiguous()):
            raise UavAgentError("tables, biases and outputs must be contiguous float32 on the device of the observation")
    if (w_c is None) != (out_c is None) or out_a.shape != (N, H) or (out_c is not None and out_c.shape != (N, H)):
        raise UavAgentError("outputs must be [N, H]; w_c and out_c go together")
    if idx_out is not None and (idx_out.shape != (N, B + U) or idx_out.dtype != torch.int64 or not idx_out.is_contiguous()):
        raise UavAgentError("idx_out must be contiguous int64 [N, B + U]")
    with torch.cuda.device(ue.device):
        rc = load().uavagent_first_layer_from_obs_f32(_ptr(w_a), _ptr(b_a), _ptr(out_a), _ptr(w_c), _ptr(b_c), _ptr(out_c), _ptr(ue), _ptr(bs),
                                                      _ptr(srv), N, U, B, int(grid_n), H, S, 1 if relu6 else 0, _ptr(idx_out),
                                                      _stream(ue.device))
    _check(rc, "uavagent_first_layer_from_obs_f32")
    return out_a if w_c is None else (out_a, out_c)


def obs_indices(obs, grid_n, n_bs, out=None):
    """agent.obs_to_indices on the device in one launch: obs = env.observation() (ue_xy i16, bs_xy i32, serving i8)."""
    ue, bs, srv = obs["ue_xy"], obs["bs_xy"], obs["serving"]
    if ue.dtype != torch.int16 or bs.dtype != torch.int32 or srv.dtype != torch.int8:
        raise UavAgentError("obs_indices needs the env's compact observation dtypes (int16 / int32 / int8)")
    N, U, B = ue.shape[0], ue.shape[1], bs.shape[1]
    if B != n_bs:
        raise UavAgentError("n_bs does not match bs_xy")
    if out is None:
        out = torch.empty((N, B + U), dtype=torch.int64, device=ue.device)
    elif out.shape != (N, B + U) or out.dtype != torch.int64 or not out.is_contiguous():
        raise UavAgentError("out must be contiguous int64 [N, B + U]")
    with torch.cuda.device(ue.device):
        rc = load().uavagent_obs_indices(_ptr(ue.contiguous()), _ptr(bs.contiguous()), _ptr(srv.contiguous()), N, U, B, int(grid_n),
                                         _ptr(out), _stream(ue.device))
    _check(rc, "uavagent_obs_indices")
    return out


def sample_actions(logits, uniforms, out=None, prob_out=None):
    """softmax + inverse-CDF draw per row (main.py:165-169) with caller-supplied uniforms [N]; logits may be a column slice of a
    wider buffer (rows contiguous)."""
    ld = _row_stride(logits, "logits")
    _f32c(uniforms, "uniforms")
    N, A = logits.shape
    if uniforms.numel() != N:
        raise UavAgentError("one uniform per row")
    if out is None:
        out = torch.empty((N,), dtype=torch.int64, device=logits.device)
    with torch.cuda.device(logits.device):
        rc = load().uavagent_sample_actions(_ptr(logits), ld, _ptr(uniforms), N, A, _ptr(out), _ptr(prob_out), _stream(logits.device))
    _check(rc, "uavagent_sample_actions")
    return out


def rowdot(y, w, b, out):
    _f32c(y, "y")
    with torch.cuda.device(y.device):
        rc = load().uavagent_rowdot_f32(_ptr(y), _ptr(w), _ptr(b), y.shape[0], y.shape[1], _ptr(out), _stream(y.device))
    _check(rc, "uavagent_rowdot_f32")
    return out


def loss_grad_workspace(n_actions, device):
    return torch.empty(load().uavagent_loss_grad_workspace_bytes(int(n_actions)), dtype=torch.uint8, device=device)


def a2c_loss_grad(logits, v, target, actions, beta, dv_out, dbias_out, loss_out, ws):
    """In place: logits <- d a_loss / d logits.  loss_out: float64 [3] = (a_loss, c_loss, sum dv).  logits may be a column slice of a
    wider buffer (rows contiguous)."""
    ld = _row_stride(logits, "logits")
    M, A = logits.shape
    with torch.cuda.device(logits.device):
        rc = load().uavagent_a2c_loss_grad(_ptr(logits), ld, _ptr(v), _ptr(target), _ptr(actions), M, A, float(beta), _ptr(dv_out),
                                           _ptr(dbias_out), _ptr(loss_out), _ptr(ws), _stream(logits.device))
    _check(rc, "uavagent_a2c_loss_grad")


def relu6_bwd_workspace(n_cols, device):
    return torch.empty(load().uavagent_relu6_bwd_workspace_bytes(int(n_cols)), dtype=torch.uint8, device=device)


def relu6_bwd(dy, y, dx_out, ldx, dbias_out, ws, dv=None, w3=None, dw3_out=None):
    """dx = dy * (0 < y < 6) + column sums; dy=None: the critic head form (dy = dv x w3)."""
    _f32c(y, "y")
    M, Cn = y.shape
    with torch.cuda.device(y.device):
        rc = load().uavagent_relu6_bwd(_ptr(dy), _ptr(y), _ptr(dv), _ptr(w3), M, Cn, _ptr(dx_out), int(ldx), _ptr(dbias_out),
                                       _ptr(dw3_out), _ptr(ws), _stream(y.device))
    _check(rc, "uavagent_relu6_bwd")


def rows_grad_workspace(m_rows, k, n_cols_total, n_rows, device):
    n = load().uavagent_rows_grad_workspace_bytes(int(m_rows), int(k), int(n_cols_total), int(n_rows))
    if n == 0:
        raise UavAgentError("uavagent_rows_grad_workspace_bytes: %s" % load().uavagent_last_error().decode())
    return torch.empty(n + 256, dtype=torch.uint8, device=device)


def rows_grad(idx, g, h, n_rows, dw0, dw1, ws):
    """dw[r] = sum of g[m] over pairs idx[m, k] == r (deterministic sort + segmented sum).  g: [M, h] or [M, 2h]."""
    _f32c(g, "g")
    M, K = idx.shape
    n_tables = g.shape[1] // h
    off = (-ws.data_ptr()) % 256
    with torch.cuda.device(g.device):
        rc = load().uavagent_rows_grad_f32(_ptr(idx), _ptr(g), M, K, int(h), n_tables, int(n_rows), _ptr(dw0), _ptr(dw1),
                                           C.c_void_p(ws.data_ptr() + off), ws.numel() - off, _stream(g.device))
    _check(rc, "uavagent_rows_grad_f32")


def rows_grad_sort(idx, n_cols_total, n_rows, ws):
    """First half of rows_grad: the stable sort of the (row, sample) pairs into ws (needs only idx; current stream)."""
    if idx.dtype != torch.int64 or idx.dim() != 2 or not idx.is_contiguous():
        raise UavAgentError("idx must be contiguous int64 [M, K]")
    M, K = idx.shape
    off = (-ws.data_ptr()) % 256
    with torch.cuda.device(idx.device):
        rc = load().uavagent_rows_grad_sort(_ptr(idx), M, K, int(n_cols_total), int(n_rows), C.c_void_p(ws.data_ptr() + off), ws.numel() - off,
                                            _stream(idx.device))
    _check(rc, "uavagent_rows_grad_sort")


def rows_grad_sums(idx_shape, g, h, n_rows, dw0, dw1, ws):
    """Second half: the segmented sums over the pairs rows_grad_sort left in ws (same idx; ordered behind the sort by the caller)."""
    _f32c(g, "g")
    M, K = idx_shape
    n_tables = g.shape[1] // h
    off = (-ws.data_ptr()) % 256
    with torch.cuda.device(g.device):
        rc = load().uavagent_rows_grad_sums_f32(_ptr(g), M, K, int(h), n_tables, int(n_rows), _ptr(dw0), _ptr(dw1),
                                                C.c_void_p(ws.data_ptr() + off), ws.numel() - off, _stream(g.device))
    _check(rc, "uavagent_rows_grad_sums_f32")


def nstep_returns(rewards, bootstrap, gamma, out=None):
    """[T, N] value targets: out[t] = r[t] + gamma * out[t + 1], out[T] = bootstrap (agent.nstep_returns in one launch)."""
    _f32c(rewards, "rewards")
    _f32c(bootstrap, "bootstrap")
    T, N = rewards.shape
    if out is None:
        out = torch.empty_like(rewards)
    with torch.cuda.device(rewards.device):
        rc = load().uavagent_nstep_returns_f32(_ptr(rewards), _ptr(bootstrap), N, T, float(gamma), _ptr(out), _stream(rewards.device))
    _check(rc, "uavagent_nstep_returns_f32")
    return out


def rmsprop_tf1(w, ms, g, lr, decay=0.9, eps=1e-10, g_scale=1.0):
    with torch.cuda.device(w.device):
        rc = load().uavagent_rmsprop_tf1(_ptr(w), _ptr(ms), _ptr(g), w.numel(), float(lr), float(decay), float(eps), float(g_scale),
                                         _stream(w.device))
    _check(rc, "uavagent_rmsprop_tf1")


def _row_stride(t, what):
    """Row stride (floats) of a 2-D float32 CUDA tensor whose rows are contiguous (a column slice of a wider buffer qualifies)."""
    if t.dtype != torch.float32 or not t.is_cuda or t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise UavAgentError("%s must be a float32 CUDA matrix with contiguous rows" % what)
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def gemm_rows_workspace(m_rows, device):
    return torch.empty(max(int(load().uavagent_gemm_rows_workspace_bytes(int(m_rows))), 16), dtype=torch.uint8, device=device)


def gemm_rows(a, w, out, w_transposed=False, bias=None, relu6=False, relu6_mask_h=None, colsum_out=None, workspace=None):
    """out[m, :] = epilogue(a[m, :] @ w) (w [K, N]) or epilogue(a[m, :] @ w.T) (w [N, K], w_transposed): uavagent_gemm_rows_f32,
    N <= 208.  Epilogue: + bias, relu6; or the relu6-backward mask by the layer's forward output relu6_mask_h.  colsum_out [N]:
    column sums of out as stored (needs workspace = gemm_rows_workspace(M, device)).  Returns out."""
    lib = load()
    M, K = a.shape
    N = w.shape[0] if w_transposed else w.shape[1]
    if (w.shape[1] if w_transposed else w.shape[0]) != K or tuple(out.shape) != (M, N):
        raise UavAgentError("gemm_rows: shapes do not agree: a %s, w %s (transposed=%s), out %s" % (tuple(a.shape), tuple(w.shape), w_transposed, tuple(out.shape)))
    if bias is not None:
        _f32c(bias, "bias")
        if bias.numel() != N:
            raise UavAgentError("gemm_rows: bias must have %d elements" % N)
    if colsum_out is not None:
        _f32c(colsum_out, "colsum_out")
        if colsum_out.numel() != N:
            raise UavAgentError("gemm_rows: colsum_out must have %d elements" % N)
    h = relu6_mask_h
    if h is not None and tuple(h.shape) != (M, N):
        raise UavAgentError("gemm_rows: relu6_mask_h must be [%d, %d]" % (M, N))
    if workspace is not None and (workspace.dtype != torch.uint8 or not workspace.is_cuda or not workspace.is_contiguous()):
        raise UavAgentError("gemm_rows: workspace must be a contiguous uint8 CUDA tensor (gemm_rows_workspace)")
    _same_device("gemm_rows", a, w, out, bias, h, colsum_out, workspace)
    with torch.cuda.device(a.device):
        rc = lib.uavagent_gemm_rows_f32(_ptr(a), _row_stride(a, "a"), _ptr(w), _row_stride(w, "w"), 1 if w_transposed else 0, M, K, N,
                                        _ptr(bias), 1 if relu6 else 0, _ptr(h), 0 if h is None else _row_stride(h, "relu6_mask_h"),
                                        _ptr(out), _row_stride(out, "out"), _ptr(colsum_out), _ptr(workspace),
                                        0 if workspace is None else workspace.numel(), _stream(a.device))
    _check(rc, "uavagent_gemm_rows_f32")
    return out


def actor_head(h1, w2t, b2, w3t_padded, b3_padded, uniforms, n_actions, h2_out, logits_pad_out, actions_out):
    """h2 = relu6(h1 @ W2 + b2); logits = h2 @ W3 + b3; one action per row by the inverse-CDF draw: uavagent_actor_head_f32 (one launch;
    bit-identical to gemm_rows x 2 + sample_actions).  logits_pad_out: [N, >= 640] rows, all 640 columns written (zero tail)."""
    for t, what in ((h1, "h1"), (w2t, "w2t"), (b2, "b2"), (w3t_padded, "w3t_padded"), (b3_padded, "b3_padded"), (uniforms, "uniforms"), (h2_out, "h2_out")):
        _f32c(t, what)
    N, H = h1.shape
    if tuple(w2t.shape) != (H, H) or w3t_padded.shape[1] != H or w3t_padded.shape[0] != b3_padded.numel() or tuple(h2_out.shape) != (N, H):
        raise UavAgentError("actor_head: shapes do not agree")
    if actions_out.dtype != torch.int64 or actions_out.numel() != N or uniforms.numel() != N or logits_pad_out.shape[0] != N:
        raise UavAgentError("actor_head: one uniform, one logits row and one int64 action per row of h1")
    if b2.numel() != H:
        raise UavAgentError("actor_head: b2 must have %d elements" % H)
    _same_device("actor_head", h1, w2t, b2, w3t_padded, b3_padded, uniforms, h2_out, logits_pad_out, actions_out)
    with torch.cuda.device(h1.device):
        rc = load().uavagent_actor_head_f32(_ptr(h1), _ptr(w2t), _ptr(b2), _ptr(w3t_padded), _ptr(b3_padded), _ptr(uniforms), N, H, int(n_actions),
                                            _ptr(h2_out), _ptr(logits_pad_out), _row_stride(logits_pad_out, "logits_pad_out"), _ptr(actions_out),
                                            _stream(h1.device))
    _check(rc, "uavagent_actor_head_f32")
    return actions_out


def gate_prepare():
    """Allocates the host-mapped error word of the gated head once per process (never inside a stream capture)."""
    _check(load().uavagent_gate_prepare(), "uavagent_gate_prepare")


def device_error():
    """0, or the word a gated launch left when one of its waits timed out (0x47415445 "GATE")."""
    code = C.c_uint32(0)
    _check(load().uavagent_device_error(C.byref(code)), "uavagent_device_error")
    return int(code.value)


def device_error_clear():
    _check(load().uavagent_device_error_clear(), "uavagent_device_error_clear")


def actor_head_gated(h1, w2t, b2, w3t_padded, b3_padded, uniforms, n_actions, h2_out, logits_pad_out, actions_out, gate_obs, gate_actions,
                     claim, spin_us=0):
    """uavagent_actor_head_gated_f32: the head of all T steps of a rollout in one persistent launch beside uavenv_rollout_gated
    (BatchedMobiEnv.rollout_gated).  h1, h2_out [T, N, 200]; uniforms [T, N]; logits_pad_out [T, N, >= 640]; actions_out int64 [T, N];
    gate_obs / gate_actions int32 [ceil(N / 16)]; claim int32 [1], zero before the launch."""
    for t, what in ((h1, "h1"), (w2t, "w2t"), (b2, "b2"), (w3t_padded, "w3t_padded"), (b3_padded, "b3_padded"), (uniforms, "uniforms"), (h2_out, "h2_out"),
                    (logits_pad_out, "logits_pad_out")):
        _f32c(t, what)
    if h1.dim() != 3:
        raise UavAgentError("actor_head_gated: h1 must be [T, N, H]")
    T, N, H = h1.shape
    nb = (N + 15) // 16
    if tuple(w2t.shape) != (H, H) or w3t_padded.shape[1] != H or w3t_padded.shape[0] != b3_padded.numel() or tuple(h2_out.shape) != (T, N, H) or b2.numel() != H:
        raise UavAgentError("actor_head_gated: shapes do not agree")
    if actions_out.dtype != torch.int64 or tuple(actions_out.shape) != (T, N) or not actions_out.is_contiguous() or tuple(uniforms.shape) != (T, N):
        raise UavAgentError("actor_head_gated: uniforms float32 [T, N] and actions_out contiguous int64 [T, N]")
    if logits_pad_out.dim() != 3 or tuple(logits_pad_out.shape[:2]) != (T, N):
        raise UavAgentError("actor_head_gated: logits_pad_out must be [T, N, >= 640]")
    for g, what in ((gate_obs, "gate_obs"), (gate_actions, "gate_actions")):
        if g.dtype != torch.int32 or tuple(g.shape) != (nb,) or not g.is_contiguous():
            raise UavAgentError("actor_head_gated: %s must be a contiguous int32 [%d] tensor" % (what, nb))
    if claim.dtype != torch.int32 or claim.numel() != 1:
        raise UavAgentError("actor_head_gated: claim must be an int32 [1] tensor")
    _same_device("actor_head_gated", h1, w2t, b2, w3t_padded, b3_padded, uniforms, h2_out, logits_pad_out, actions_out, gate_obs, gate_actions, claim)
    with torch.cuda.device(h1.device):
        rc = load().uavagent_actor_head_gated_f32(_ptr(h1), _ptr(w2t), _ptr(b2), _ptr(w3t_padded), _ptr(b3_padded), _ptr(uniforms), N, T, H,
                                                  int(n_actions), _ptr(h2_out), _ptr(logits_pad_out), int(logits_pad_out.shape[2]),
                                                  _ptr(actions_out), _ptr(gate_obs), _ptr(gate_actions), _ptr(claim), int(spin_us), _stream(h1.device))
    _check(rc, "uavagent_actor_head_gated_f32")
    return actions_out


def gemm_tn_workspace(m_rows, n_j, device):
    n = load().uavagent_gemm_tn_workspace_bytes(int(m_rows), int(n_j))
    return torch.empty(max(int(n), 16), dtype=torch.uint8, device=device)


def gemm_tn(a, b, out, workspace, dbias_out=None):
    """out[i, j] = sum_m a[m, i] * b[m, j] (the weight gradient x^T dy) and dbias_out[j] = sum_m b[m, j]: uavagent_gemm_tn_f32.
    a [M, I <= 200] contiguous, b [M, J <= 640] with contiguous rows; deterministic (ordered second pass)."""
    lib = load()
    _f32c(a, "a")
    M, I = a.shape
    J = b.shape[1]
    if b.shape[0] != M or tuple(out.shape) != (I, J):
        raise UavAgentError("gemm_tn: shapes do not agree: a %s, b %s, out %s" % (tuple(a.shape), tuple(b.shape), tuple(out.shape)))
    if dbias_out is not None:
        _f32c(dbias_out, "dbias_out")
        if dbias_out.numel() != J:
            raise UavAgentError("gemm_tn: dbias_out must have %d elements" % J)
    if workspace is None or workspace.dtype != torch.uint8 or not workspace.is_cuda or not workspace.is_contiguous():
        raise UavAgentError("gemm_tn: workspace must be a contiguous uint8 CUDA tensor (gemm_tn_workspace)")
    _same_device("gemm_tn", a, b, out, dbias_out, workspace)
    with torch.cuda.device(a.device):
        rc = lib.uavagent_gemm_tn_f32(_ptr(a), _ptr(b), M, I, J, _row_stride(b, "b"), _ptr(out), _row_stride(out, "out"), _ptr(dbias_out),
                                      _ptr(workspace), workspace.numel(), _stream(a.device))
    _check(rc, "uavagent_gemm_tn_f32")
    return out


_bag_cache = {}


def _bags(M, K, device):
    """offset2bag / bag_size / maximum_indices of M bags of K entries, as _embedding_bag returns them for mode="sum"."""
    key = (M, K, str(device))
    if key not in _bag_cache:
        if len(_bag_cache) > 8:
            _bag_cache.clear()
        _bag_cache[key] = (torch.arange(M, device=device).repeat_interleave(K),
                           torch.full((M,), K, dtype=torch.int64, device=device),
                           torch.empty(0, dtype=torch.int64, device=device))
    return _bag_cache[key]


def _table_grad_aten(g, idx, n_rows):
    """ATen's own embedding_bag backward (what autograd runs for the reference form): the comparison target of rows_grad."""
    o2b, size, max_idx = _bags(idx.shape[0], idx.shape[1], idx.device)
    flat = idx.reshape(-1)
    keep = (flat >= 0).to(g.dtype)                # -1 = "no row" (agent.obs_to_indices): no gradient, as in first_layer_reference
    return torch.ops.aten._embedding_bag_dense_backward(g.contiguous(), flat.clamp(min=0), o2b, size, max_idx, n_rows, False, 0,
                                                        keep, -1)


class _SparseRowsSum(torch.autograd.Function):
    """Differentiable first layer for the autograd (reference) path of the learner: forward = the HIP gather kernel, backward =
    uavagent_rows_grad_f32 for both tables in one sorted pass + plain column sums for the biases."""

    @staticmethod
    def forward(ctx, idx, w_a, b_a, w_c, b_c):
        ctx.save_for_backward(idx)
        ctx.n_rows = w_a.shape[0]
        ctx.h = w_a.shape[1]
        ctx.two = w_c is not None
        out = sparse_rows_sum(idx, w_a, b_a, w_c, b_c)
        if ctx.two:
            return out
        return out, None

    @staticmethod
    def backward(ctx, g_a, g_c):
        (idx,) = ctx.saved_tensors
        need = ctx.needs_input_grad            # (idx, w_a, b_a, w_c, b_c)
        two = ctx.two and g_c is not None
        gw_a = gw_c = None
        if need[1] or (two and need[3]):
            g = torch.cat([g_a, g_c], dim=1).contiguous() if two else g_a.contiguous()
            gw_a = torch.empty((ctx.n_rows, ctx.h), dtype=torch.float32, device=g.device)
            gw_c = torch.empty_like(gw_a) if two else None
            ws = rows_grad_workspace(idx.shape[0], idx.shape[1], g.shape[1], ctx.n_rows, g.device)
            rows_grad(idx.contiguous(), g, ctx.h, ctx.n_rows, gw_a, gw_c, ws)
        gb_a = g_a.sum(dim=0) if need[2] else None
        gb_c = g_c.sum(dim=0) if (two and need[4]) else None
        return None, gw_a, gb_a, gw_c, gb_c


def sparse_first_layer_cuda(idx, w_a, b_a, w_c=None, b_c=None):
    """Differentiable: (h_a, h_c) with h = sum_k W[idx[:, k]] + b; h_c is None when no critic table is given."""
    return _SparseRowsSum.apply(idx, w_a, b_a, w_c, b_c)
