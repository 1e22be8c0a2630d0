"""CPU, world_size 2, gloo: the N>1 path.  Env instances shard with no data-path collective, so what must hold is
(1) rank r's shard (env_id_base = r*E) is bit-identical to slice [r*E, (r+1)*E) of one batch of 2E envs, and
(2) the timing reduction bench.py uses takes the slowest rank.  The per-env computation is the CPU oracle here
(the HIP path has no CPU fallback); HIP == oracle per env and shard == slice on the device are GPU tests
(test_hip_parity.py::test_sharding_and_determinism)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E, T, SEED = 24, 12, 0xABCDEF


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_envs(n, base):
    from oracle import oracle as O

    env = O.OracleEnv(O.make_config(4, 20, 100, groups=[5, 5, 5, 5]), n, seed=SEED, env_id_base=base)
    env.construct()
    rs = np.random.RandomState(3)
    acts = rs.randint(0, 625, size=(T, 2 * E)).astype(np.int64)   # one global action table, sliced per shard
    outs = []
    for t in range(T):
        o = env.step(acts[t, base:base + n])
        outs.append(np.concatenate([o["ue_xy"].reshape(n, -1).astype(np.float64), o["serving"].astype(np.float64),
                                    o["cur_sinr_f64"], o["reward_f64"][:, None], o["n_out"][:, None].astype(np.float64)],
                                   axis=1))
    return np.stack(outs)   # [T, n, F]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from drl_uav_cellularnet_amd.sharding import max_over_ranks, shard_for_rank, whole_job_rate

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        base, sl = shard_for_rank(rank, world, E)
        assert (base, sl) == (rank * E, slice(rank * E, (rank + 1) * E))
        mine = torch.from_numpy(_run_envs(E, base))
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)                  # test-only exchange: lets rank 0 compare with one big batch
        fake_times = [1.0 + rank, 10.0 - rank]           # rank 1 slower on the first, rank 0 on the second
        red = max_over_ranks(fake_times)
        dist.barrier()
        if rank == 0:
            whole = _run_envs(world * E, 0)
            sharded = torch.cat(gathered, dim=1).numpy()
            q.put(("ok", bool(np.array_equal(whole, sharded)), red, whole_job_rate(E * T, world, red[0])))
    except Exception as exc:  # surface the failure in the parent
        q.put(("err", repr(exc), None, None))
        raise
    finally:
        dist.destroy_process_group()


def test_two_rank_shards_equal_one_big_batch():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    status, equal, red, rate = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert status == "ok", equal
    assert equal, "shards (env_id_base = rank*E) differ from the slices of one 2E batch"
    assert red == [2.0, 10.0]                             # element-wise MAX over ranks
    assert rate == pytest.approx(2 * E * T / 2.0)


def test_shard_spec_validation():
    from drl_uav_cellularnet_amd.sharding import max_over_ranks, shard_for_rank

    assert shard_for_rank(3, 8, 8192) == (24576, slice(24576, 32768))     # BASELINE config 4: 65536 envs on 8 GPUs
    with pytest.raises(ValueError):
        shard_for_rank(2, 2, 8)
    with pytest.raises(ValueError):
        shard_for_rank(0, 2, 2 ** 32)
    assert max_over_ranks([1.5, 2.5]) == [1.5, 2.5]       # no process group: identity
