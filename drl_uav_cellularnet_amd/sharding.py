"""Multi-GPU layout of the env path: env instances are independent, so rank r simply owns
the contiguous block of global env ids [r*E, (r+1)*E).  There is no data-path collective; the only
communication is the barrier / max-reduce of the timing in bench.py.  torch.distributed is plumbing here.
"""


def shard_for_rank(rank, world_size, envs_per_rank):
    """-> (env_id_base, global_slice).  ``env_id_base`` goes to BatchedMobiEnv / uavenv_create: it offsets the
    Philox env id, so env e of rank r is bit-identical to env r*E + e of a single batch of world_size*E envs."""
    rank, world_size, envs_per_rank = int(rank), int(world_size), int(envs_per_rank)
    if not (0 <= rank < world_size) or envs_per_rank < 1:
        raise ValueError("need 0 <= rank < world_size and envs_per_rank >= 1")
    if world_size * envs_per_rank > 0xFFFFFFFF:
        raise ValueError("global env ids must fit the 32-bit Philox counter word")
    base = rank * envs_per_rank
    return base, slice(base, base + envs_per_rank)


def max_over_ranks(values, device=None):
    """Element-wise MAX of a list of floats over all ranks (identity when torch.distributed is not initialised).
    bench.py reports the slowest rank's time, as the bench contract requires."""
    import torch

    dist = torch.distributed
    if not (dist.is_available() and dist.is_initialized()):
        return [float(v) for v in values]
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t]


def whole_job_rate(units_per_rank, world_size, seconds):
    """Whole-job throughput: units processed by ALL ranks / the slowest rank's time."""
    return float(units_per_rank) * int(world_size) / float(seconds)


def gather_over_ranks(values, device=None):
    """[[values of rank 0], [values of rank 1], ...] on every rank (one all-gather of a small float64 tensor; [[values]] when
    torch.distributed is not initialised).  bench.py prints every rank's own elapsed time beside the MAX it reports, so that a
    straggler is visible in the line instead of hidden in the maximum."""
    import torch

    dist = torch.distributed
    vals = [float(v) for v in values]
    if not (dist.is_available() and dist.is_initialized()):
        return [vals]
    t = torch.tensor(vals, dtype=torch.float64, device=device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[float(v) for v in o] for o in out]
