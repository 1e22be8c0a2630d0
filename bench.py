#!/usr/bin/env python3
"""bench.py -- env steps/s of the HIP hot path (BASELINE.json metric), one JSON line on rank 0.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--envs E] [--no-cpu-baseline]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one batched MobiEnvironment.step() over all envs of a rank = ONE kernel launch.
Workload at N=1: BASELINE.json configs[1] -- 4096 envs, 4 UAV x 20 UE (groups 5,5,5,5), G=100, on-device
Philox randomness, uniform random joint actions resident in HBM before the timed region, compact
outputs only (no dense observation).  N>1: the same workload per rank (weak scaling); env instances are
independent, so there is NO data-path collective -- only the barrier/max-reduce of the timing.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_BS, N_UE, GRID, GROUPS = 4, 20, 100, [5, 5, 5, 5]
SEED = 0x5EED
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def algorithmic_bytes_per_env_step(U, B, Gr):
    """compact state read+write + outputs, on-device RNG."""
    return 48 * U + 2 * ((U + 7) // 8) + 96 * Gr + 16 * B + 45


def transcendental_evals_per_env_step(U, B):
    """SURVEY.md section 8(d) asks for the achieved transcendental rate beside the HBM figure (the kernel is issue-bound, not
    bandwidth-bound).  Float64 function evaluations per walker and step in csrc/uavenv_kernels.h, HB = ceil(B/2) Box-Muller pairs:
    sincospi 1 (heading) + HB;  log HB (Box-Muller) + 2 (SINR of the best and of the serving UAV);
    rsqrt 1 (pull towards the group centre) + HB (Box-Muller radius) + B (d^-3);  exp2 B (shadowing)  =  4 + 3*HB + 2*B."""
    hb = (B + 1) // 2
    return U * (4 + 3 * hb + 2 * B)


STEP_KERNEL = "env_kernel_packed<4, 2, true, true, true>"   # rocprofv3 name (template part) of the step kernel of this workload


def measured_traffic(envs):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic_current.json), or None when
    that file describes another kernel / batch size.  bench.py cannot run the profiler on itself."""
    path = os.path.join(ROOT, "profiles", "traffic_current.json")
    try:
        with open(path) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None
    if t.get("kernel") != STEP_KERNEL or t.get("envs") != envs or t.get("n_ue") != N_UE or t.get("n_bs") != N_BS:
        return None
    return int(t["fetch_size_bytes_raw"]) + int(t["write_size_bytes_raw"])


def cpu_baseline(target_seconds=15.0):
    """Times the CPU oracle (oracle/, kind 'port': scalar C restatement of the reference's step())
    on a bounded sample of the same workload, one shard per host thread."""
    import numpy as np

    from oracle import oracle as O

    cores = max(1, min(os.cpu_count() or 1, 16))
    per = 64
    cfg = O.make_config(N_BS, N_UE, GRID, groups=GROUPS)
    envs = [O.OracleEnv(cfg, per, seed=SEED, env_id_base=i * per) for i in range(cores)]
    for e in envs:
        e.construct()
    rs = np.random.RandomState(1234)
    acts = rs.randint(0, 625, size=(256, per)).astype(np.int64)
    # calibrate on one thread, then size the sample for ~target_seconds
    t0 = time.perf_counter()
    for t in range(20):
        envs[0].step(acts[t])
    dt = (time.perf_counter() - t0) / 20
    steps = int(max(50, min(60000, target_seconds / max(dt, 1e-9))))

    def work(env):
        for t in range(steps):
            env.step(acts[t % 256])

    th = [threading.Thread(target=work, args=(e,)) for e in envs]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    el = time.perf_counter() - t0
    return {"value": cores * per * steps / el, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%d threads x %d envs x %d steps of oracle step() (4 UAV x 20 UE, G=100, Philox), %.1f s"
                      % (cores, per, steps, el)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs", type=int, default=4096, help="env instances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to "
                    "rehearse the multi-rank path with several ranks sharing one GPU)")
    ap.add_argument("--force-device", type=int, default=None, help="rehearsal only: every rank uses this GPU")
    ap.add_argument("--force-dist", action="store_true", help="rehearsal only: initialise the process group even at "
                    "world size 1 (exercises the RCCL init / barrier / max-reduce path on a one-GPU box)")
    ap.add_argument("--n-bs", type=int, default=N_BS, help="secondary measurements only (default = BASELINE workload)")
    ap.add_argument("--n-ue", type=int, default=N_UE, help="secondary measurements only (default = BASELINE workload)")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if args.force_device is not None:
            local_rank = args.force_device
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from drl_uav_cellularnet_amd import BatchedMobiEnv
    from drl_uav_cellularnet_amd.sharding import max_over_ranks, shard_for_rank, whole_job_rate

    E, K, W = args.envs, args.steps, args.warmup
    env_id_base, _ = shard_for_rank(rank, world, E)   # rank r owns global envs [r*E, (r+1)*E): no env-path collective
    n_bs, n_ue = args.n_bs, args.n_ue
    baseline_shape = (n_bs, n_ue) == (N_BS, N_UE)
    groups = GROUPS if baseline_shape else [n_ue // 4] * 3 + [n_ue - 3 * (n_ue // 4)]
    env = BatchedMobiEnv(E, nBS=n_bs, nUE=n_ue, grid_n=GRID, groups=groups, device=dev, seed=SEED,
                         env_id_base=env_id_base)
    gen = torch.Generator(device="cpu").manual_seed(1234 + rank)
    n_act = min(K + W, 512)  # action table resident in HBM, cycled
    actions = torch.randint(0, min(env.action_space_dim, 2 ** 62), (n_act, E), generator=gen, dtype=torch.int64).to(dev)
    max_step = int(env.cfg.max_step)

    def run(n, start):
        for t in range(start, start + n):
            env.step(actions[t % n_act])
            if (t + 1) % max_step == 0:  # the reference's callers reset on `done` (main.py:205-211)
                env.reset()

    run(W, 0)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run(K, W)
    ev1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    gpu_ms = ev0.elapsed_time(ev1)
    elapsed, gpu_ms = max_over_ranks([elapsed, gpu_ms], device=dev if args.backend == "nccl" else None)  # slowest rank

    if rank == 0:
        per_launch_s = gpu_ms * 1e-3 / K  # average launch-to-launch time of the step kernel (HIP events)
        b_step = algorithmic_bytes_per_env_step(n_ue, n_bs, len(groups))
        achieved = b_step * E / per_launch_s / 1e9
        line = {
            "metric": "env steps/sec (whole node) at 4-UAV x 20-UE", "value": whole_job_rate(E * K, world, elapsed),
            "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d batched envs/GPU, %d UAV x %d UE (groups %s), G=100, HIP step(), "
                                   "compact outputs, on-device Philox, one launch per step" % (
                                       E, n_bs, n_ue, ",".join(str(g) for g in groups)),
                       "envs_per_gpu": E, "n_bs": n_bs, "n_ue": n_ue, "grid": GRID, "parallelism": "env-shard x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": measured_traffic(E) if baseline_shape else None,
                         "kernel": STEP_KERNEL if baseline_shape else "env kernel of this shape (secondary measurement)", "algorithmic_bytes_per_launch": b_step * E,
                         "avg_launch_us": per_launch_s * 1e6,
                         "transcendental_evals_per_launch": transcendental_evals_per_env_step(n_ue, n_bs) * E,
                         "transcendental_evals_per_s": transcendental_evals_per_env_step(n_ue, n_bs) * E / per_launch_s},
        }
        if world == 1 and not args.no_cpu_baseline and baseline_shape:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
