"""ctypes binding of libuavagent.so (include/uavagent.h) and the autograd wrapper agent.py uses for CUDA tensors.

The plain PyTorch form, ``F.embedding_bag(idx, W, mode="sum") + b``, stays in agent.py as the reference implementation and the
CPU path.  Here: forward = the HIP kernel (one launch for the actor and the critic table, which share their indices);
backward = ATen's own ``_embedding_bag_dense_backward`` -- the routine autograd runs for embedding_bag -- so gradients are
the reference's by construction.  A CUDA tensor with no library is an error, not a silent PyTorch fallback.
"""
import ctypes as C
import os

import torch

from . import build as _build

EXPORTS = ("uavagent_abi_version", "uavagent_last_error", "uavagent_sparse_rows_sum_f32")

_lib = None


class UavAgentError(RuntimeError):
    pass


def lib_path():
    return os.environ.get("UAVAGENT_LIB") or _build.AGENT_LIB


def load():
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.isfile(path):
        raise UavAgentError("%s not found: run `python -m drl_uav_cellularnet_amd.build` (there is no fallback for CUDA "
                            "tensors)" % path)
    lib = C.CDLL(path)
    lib.uavagent_abi_version.restype = C.c_int
    lib.uavagent_last_error.restype = C.c_char_p
    lib.uavagent_sparse_rows_sum_f32.restype = C.c_int
    lib.uavagent_sparse_rows_sum_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_void_p]
    if lib.uavagent_abi_version() != 1:
        raise UavAgentError("libuavagent.so ABI version mismatch")
    _lib = lib
    return lib


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def sparse_rows_sum(idx, w_a, b_a, w_c=None, b_c=None):
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
    out_a = torch.empty((M, H), dtype=torch.float32, device=idx.device)
    out_c = None if w_c is None else torch.empty((M, H), dtype=torch.float32, device=idx.device)
    with torch.cuda.device(idx.device):
        stream = C.c_void_p(torch.cuda.current_stream(idx.device).cuda_stream)
        rc = lib.uavagent_sparse_rows_sum_f32(_ptr(w_a), _ptr(b_a), _ptr(out_a), _ptr(w_c), _ptr(b_c), _ptr(out_c), _ptr(idx),
                                              M, K, H, S, stream)
    if rc != 0:
        raise UavAgentError("uavagent_sparse_rows_sum_f32: %s" % lib.uavagent_last_error().decode())
    return out_a if w_c is None else (out_a, out_c)


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


def _bias_grad(g):
    """Column sums of g [M, H].  Measured alternatives that did NOT help (tools/profile_a2c.py, 65536 x 200): torch.mv(g.t(), ones)
    runs rocBLAS gemv at 614 us per call, twice as slow; a two-stage slab reduction changes nothing.  (The 288 us column
    reductions that stand out in the update profile are autograd's own bias gradients of the hidden layers, not these.)"""
    return g.sum(dim=0)


def _table_grad(g, idx, n_rows):
    o2b, size, max_idx = _bags(idx.shape[0], idx.shape[1], idx.device)
    flat = idx.reshape(-1)
    keep = (flat >= 0).to(g.dtype)                # -1 = "no row" (agent.obs_to_indices): no gradient, as in first_layer_reference
    return torch.ops.aten._embedding_bag_dense_backward(g.contiguous(), flat.clamp(min=0), o2b, size, max_idx, n_rows, False, 0,
                                                        keep, -1)


class _SparseRowsSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, idx, w_a, b_a, w_c, b_c):
        ctx.save_for_backward(idx)
        ctx.n_rows = w_a.shape[0]
        ctx.two = w_c is not None
        out = sparse_rows_sum(idx, w_a, b_a, w_c, b_c)
        if ctx.two:
            return out
        return out, None

    @staticmethod
    def backward(ctx, g_a, g_c):
        (idx,) = ctx.saved_tensors
        need = ctx.needs_input_grad            # (idx, w_a, b_a, w_c, b_c)
        gw_a = _table_grad(g_a, idx, ctx.n_rows) if need[1] else None
        gb_a = _bias_grad(g_a) if need[2] else None
        gw_c = _table_grad(g_c, idx, ctx.n_rows) if (ctx.two and need[3]) else None
        gb_c = _bias_grad(g_c) if (ctx.two and need[4]) else None
        return None, gw_a, gb_a, gw_c, gb_c


def sparse_first_layer_cuda(idx, w_a, b_a, w_c=None, b_c=None):
    """Differentiable: (h_a, h_c) with h = sum_k W[idx[:, k]] + b; h_c is None when no critic table is given."""
    return _SparseRowsSum.apply(idx, w_a, b_a, w_c, b_c)
