#!/usr/bin/env python3
"""The training loop of the reference's a2c_single_thread.py (:107-137, 207-224) on the batched HIP env: `--workers` env instances play the
reference's workers (4 there), all stepped together; episodes of MAXSTEP = 2000 steps in rollouts of `--rollout` steps, one synchronous
update per rollout; at the end the two files the reference writes: Global_return.npy (the running episode return, :169-172) and the
actor parameters (Global_A_PARA: here a named .npz, agent.save_actor_npz -- loadable by tools/run_eval.py without pickle).

  python tools/train_a2c.py --out train/run1 [--workers 8192] [--episodes 2] [--rollout 50] [--first-state zeros]
  python -m torch.distributed.run --nproc-per-node 8 tools/train_a2c.py ...     # one process per GPU, gradients all-reduced (RCCL)"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def save_checkpoint(torch, world, path, payload):
    """Crash-safe and rank-consistent: every rank writes <path>.tmp, all ranks meet at a barrier (so a kill during the save
    leaves EVERY rank's previous checkpoint in place), then each renames its file into place (atomic on POSIX) and they meet
    again before training continues.  A kill between the two barriers can still leave ranks one checkpoint apart; --resume
    detects that (check_ranks_agree) instead of hanging in the first all-reduce."""
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    tmp = path + ".tmp"
    torch.save(payload, tmp)
    if world > 1:
        torch.distributed.barrier()
    os.replace(tmp, path)
    if world > 1:
        torch.distributed.barrier()


def check_ranks_agree(torch, world, dev, episode, runner):
    """After --resume: every rank must continue from the same episode with the same weights (the weights are kept identical by
    the gradient all-reduce, so any difference means the checkpoint files are from different saves).  Exits non-zero otherwise."""
    if world <= 1:
        return
    w = runner.flat.w
    digest = torch.stack([w.double().sum(), w.double().abs().sum(), torch.tensor(float(episode), dtype=torch.float64, device=w.device)])
    lo, hi = digest.clone(), digest.clone()
    torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
    torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
    if not torch.equal(lo, hi):
        print("train_a2c: the ranks' checkpoints disagree (episode min/max %d/%d, weight digests %s vs %s): refusing to resume"
              % (int(lo[2]), int(hi[2]), lo[:2].tolist(), hi[:2].tolist()), file=sys.stderr, flush=True)
        torch.distributed.destroy_process_group()
        sys.exit(4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="train/run")
    ap.add_argument("--workers", type=int, default=8192, help="env instances per GPU")
    ap.add_argument("--episodes", type=int, default=2)
    ap.add_argument("--rollout", type=int, default=50)
    ap.add_argument("--n-ue", type=int, default=40, help="the reference's scripts run 4 UAV x 40 UE on a 100 x 100 grid")
    ap.add_argument("--grid", type=int, default=100)
    ap.add_argument("--first-state", choices=("obs", "zeros"), default="zeros")
    ap.add_argument("--checkpoint-every", type=int, default=0, help="episodes between full checkpoints (<out>/checkpoint_rank<r>.pt); 0 = never")
    ap.add_argument("--resume", action="store_true", help="continue from <out>/checkpoint_rank<r>.pt (bit-identical to an uninterrupted run)")
    ap.add_argument("--rollout-form", choices=("auto", "per-step"), default="auto", help="auto: A2CRunner's default (the two persistent rollout kernels from "
                    "4096 workers on); per-step: the pipelined per-step launches (same results bit for bit)")
    ap.add_argument("--no-gemm-tuning", action="store_true", help="leave PyTorch's TunableOp off (library default; this tool turns the "
                    "shipped per-shape GEMM picks on: a resumed run is bit-identical only if it makes the same choice as the original)")
    a = ap.parse_args()
    import numpy as np
    import torch

    rank, local, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from drl_uav_cellularnet_amd import BatchedMobiEnv
    from drl_uav_cellularnet_amd.agent import A2CRunner, save_actor_npz
    from drl_uav_cellularnet_amd.sharding import shard_for_rank

    base, _ = shard_for_rank(rank, world, a.workers)
    env = BatchedMobiEnv(a.workers, nBS=4, nUE=a.n_ue, grid_n=a.grid, device=dev, env_id_base=base)
    runner = A2CRunner(env, rollout=a.rollout, first_state=a.first_state, tune_gemms=not a.no_gemm_tuning,
                       persistent_rollout="auto" if a.rollout_form == "auto" else False)
    per_episode = int(env.cfg.max_step) // a.rollout                       # a2c_single_thread.py:108
    returns, t0, first_ep = [], time.time(), 0
    ckpt = os.path.join(a.out, "checkpoint_rank%d.pt" % rank)
    if a.resume:
        sd = torch.load(ckpt, weights_only=True)                           # (a file this tool wrote: tensors and plain scalars only)
        runner.load_state_dict(sd["runner"])
        first_ep, returns = int(sd["episode"]) + 1, list(sd["returns"])
        check_ranks_agree(torch, world, dev, first_ep, runner)
    for ep in range(first_ep, a.episodes):
        for r in range(per_episode):
            st = runner.train_rollout()
        returns.append(runner.running_r)                                   # GLOBAL_RUNNING_R, :169-172
        if rank == 0:
            print(json.dumps({"episode": ep, "episode_return": runner.last_episode_return, "running_return": runner.running_r,
                              "a_loss": st["a_loss"], "c_loss": st["c_loss"],
                              "mean_reward": st["mean_reward"], "env_steps": (ep + 1) * per_episode * a.rollout * a.workers * world,
                              "seconds": time.time() - t0}), flush=True)
        if a.checkpoint_every and (ep + 1) % a.checkpoint_every == 0:
            save_checkpoint(torch, world, ckpt, {"runner": runner.state_dict(), "episode": ep, "returns": returns})
    if rank == 0:
        os.makedirs(a.out, exist_ok=True)
        np.save(os.path.join(a.out, "Global_return"), np.array([x for x in returns if x is not None], dtype=np.float64))   # :135
        save_actor_npz(runner.net, os.path.join(a.out, "Global_A_PARA.npz"))                                             # :136
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
