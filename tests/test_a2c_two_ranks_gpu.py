"""GPU, 2 processes sharing the box's one GPU (gloo; RCCL refuses two ranks on one device): the fused A2C update with its flat
gradient all-reduce keeps the ranks' parameters BIT-IDENTICAL while each rank steps its own env shard
(a2c_single_thread.py:107-133: one synchronous update over all workers' samples).  The production path is the same call
with backend nccl = RCCL over xGMI, one rank per GPU (bench.py --mode a2c --gpus N)."""
import os
import socket

import pytest

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q, overlap):
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from drl_uav_cellularnet_amd import BatchedMobiEnv
    from drl_uav_cellularnet_amd.agent import A2CRunner
    from drl_uav_cellularnet_amd.sharding import shard_for_rank

    base, _ = shard_for_rank(rank, world, 256)
    env = BatchedMobiEnv(256, nBS=4, nUE=20, grid_n=100, device="cuda:0", env_id_base=base)
    r = A2CRunner(env, rollout=5, overlap_allreduce=overlap)
    w0 = r.flat.w.clone()
    rewards = []
    for _ in range(3):
        st = r.train_rollout()
        rewards.append(st["mean_reward"])
        assert st["grad_elems"] == 20206626
        # two buckets (critic trunk first, on a side stream while the actor trunk's backward pass runs) or one
        assert (st["allreduce_buckets"] is not None) == overlap
        if overlap:
            assert sum(st["allreduce_buckets"]) == 4 * r.flat.n_flat and st["allreduce_overlapped_ms"] is not None
    digest = [float(r.flat.w.double().sum()), float(r.flat.w.double().abs().sum()), float(r.flat.ms.double().sum())]
    q.put((rank, digest, rewards, bool(torch.equal(w0, r.flat.w)), r.flat.w[::100003].cpu().tolist()))
    dist.barrier()
    dist.destroy_process_group()


def _run_pair(overlap):
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, overlap)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def test_two_ranks_stay_in_lockstep():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    res = _run_pair(True)
    (_, d0, rew0, same0, s0), (_, d1, rew1, same1, s1) = res
    assert d0 == d1 and s0 == s1                      # identical parameters and RMSProp accumulators on both ranks
    assert not same0 and not same1                    # ... which did move
    assert rew0 != rew1                               # while the ranks saw different env shards (env_id_base)
    # the bucketed exchange (critic trunk first, overlapped) against ONE all-reduce of the whole flat buffer: the same bits
    (_, e0, rew2, _, t0), (_, e1, _, _, t1) = _run_pair(False)
    assert e0 == e1 == d0 and t0 == t1 == s0 and rew2 == rew0
