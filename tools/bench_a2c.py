#!/usr/bin/env python3
"""Secondary measurement (NOT the contract bench): end-to-end synchronous A2C on the HIP env.
BASELINE configs[2]: 8192 envs + MLP actor-critic, 1 GPU;  configs[3]: 8192 envs per GPU x 8 with one flat
gradient all-reduce per update (launch with torch.distributed.run, one process per GPU, backend nccl = RCCL).

  python tools/bench_a2c.py [--envs 8192] [--rollouts 6] [--rollout-len 50]
Prints one JSON line on rank 0: env-steps/s including policy inference, sampling, env step and the update."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=8192)
    ap.add_argument("--rollouts", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rollout-len", type=int, default=50)
    args = ap.parse_args()
    import torch

    rank, local, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from drl_uav_cellularnet_amd import BatchedMobiEnv
    from drl_uav_cellularnet_amd.agent import A2CRunner, grad_allreduce_bytes
    from drl_uav_cellularnet_amd.sharding import max_over_ranks, shard_for_rank, whole_job_rate

    base, _ = shard_for_rank(rank, world, args.envs)
    env = BatchedMobiEnv(args.envs, nBS=4, nUE=20, grid_n=100, device=dev, env_id_base=base)
    runner = A2CRunner(env, rollout=args.rollout_len)
    t_col = t_upd = 0.0
    for it in range(args.warmup + args.rollouts):
        if it == args.warmup:
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            t_col = t_upd = 0.0
        a = time.perf_counter()
        batch = runner.collect()
        torch.cuda.synchronize()
        b = time.perf_counter()
        stats = runner.update(*batch)
        torch.cuda.synchronize()
        c = time.perf_counter()
        t_col += b - a
        t_upd += c - b
    if world > 1:
        dist.barrier()
    el = max_over_ranks([time.perf_counter() - t0], device=dev)[0]
    if rank == 0:
        n = args.envs * args.rollout_len * args.rollouts
        print(json.dumps({"metric": "A2C end-to-end env steps/sec (policy + env + update)", "value": whole_job_rate(n, world, el),
                          "n_gpus": world, "envs_per_gpu": args.envs, "rollout_len": args.rollout_len, "rollouts": args.rollouts,
                          "collect_s_per_rollout": t_col / args.rollouts, "update_s_per_rollout": t_upd / args.rollouts,
                          "grad_allreduce_bytes": grad_allreduce_bytes(runner.net), "a_loss": stats["a_loss"],
                          "c_loss": stats["c_loss"], "mean_reward": stats["mean_reward"]}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
