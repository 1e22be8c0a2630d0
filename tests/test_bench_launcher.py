"""bench.py as a launcher: `python bench.py --gpus N` with no torchrun around it must start N ranks itself, verify that N
ranks really reduce together, and refuse to print a line for any other rank count (VERDICT r1 item 1 / ADVICE bench.py:116).
CPU part: --rehearse-launcher does the spawn + gloo rendezvous + rank census + max-reduce with no GPU work (value null).
GPU part (-m gpu): the real env workload on 2 ranks sharing the one GPU of the box over gloo."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                           "UAVENV_BENCH_CHILD")}
    env.update(extra)
    return env


def _last_json(text):
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert lines, "no JSON line in: %r" % text[-2000:]
    return json.loads(lines[-1])


def test_one_command_starts_and_counts_two_ranks():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rehearse-launcher"], env=_clean_env(), capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = _last_json(p.stdout)
    assert line["rehearsal"] is True and line["value"] is None
    assert line["n_gpus"] == 2 and line["launcher"] == "self"
    assert line["max_rank_plus_1"] == 2.0 and line["max_env_id_base"] == 4096.0      # rank 1 owns envs [4096, 8192)
    assert len([l for l in p.stdout.splitlines() if l.startswith("{")]) == 1         # ONE line, from rank 0


def test_one_command_starts_and_counts_eight_ranks():
    """The size of BASELINE config 4 (8 GPUs): 8 ranks spawned, every one counted, rank 7 owns envs [7 * 4096, 8 * 4096)."""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--rehearse-launcher"], env=_clean_env(), capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = _last_json(p.stdout)
    assert line["rehearsal"] is True and line["value"] is None
    assert line["n_gpus"] == 8 and line["launcher"] == "self"
    assert line["max_rank_plus_1"] == 8.0 and line["max_env_id_base"] == 7 * 4096.0
    assert len([l for l in p.stdout.splitlines() if l.startswith("{")]) == 1


def test_external_launcher_two_ranks():
    """The driver's form: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N (RANK / WORLD_SIZE from the launcher)."""
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29641", BENCH, "--gpus", "2", "--rehearse-launcher"], env=_clean_env(), capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = _last_json(p.stdout)
    assert line["n_gpus"] == 2 and line["launcher"] == "external" and line["max_env_id_base"] == 4096.0


def test_wrong_rank_count_is_refused_not_reported():
    """Under an external launcher WORLD_SIZE is authoritative: --gpus 2 inside a 1-rank job must exit non-zero, never print an
    `n_gpus: 1` line (round 1 did exactly that)."""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rehearse-launcher"],
                       env=_clean_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "{" not in p.stdout
    assert "refusing" in p.stderr


def test_segment_plan_cuts_at_resets():
    """EnvRun.plan: chunks of <= 100 steps, never across a reset boundary (MAXSTEP), reset flagged on the segment that ends there."""
    sys.path.insert(0, ROOT)
    import bench

    class _Cfg:
        max_step = 250

    class _Env:
        cfg = _Cfg()

    r = bench.EnvRun.__new__(bench.EnvRun)
    r.max_step, r.t, r.chunk = 250, 0, bench.CHUNK
    assert r.plan(20) == [(20, False)]
    assert r.plan(250) == [(100, False), (100, False), (50, True)]
    assert r.plan(300) == [(100, False), (100, False), (50, True), (50, False)]
    assert r.plan(120, t0=200) == [(50, True), (70, False)]
    assert sum(n for n, _ in r.plan(1999, t0=7)) == 1999
    r.chunk = 20                                       # bench.py --chunk 20 (the profiling passes of the driver's 20-step launch shape)
    assert r.plan(50) == [(20, False), (20, False), (10, False)] and r.plan(30, t0=240) == [(10, True), (20, False)]


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_run_the_env_workload():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--force-device", "0", "--envs", "512",
                        "--steps", "40", "--warmup", "10", "--no-cpu-baseline", "--no-a2c", "--no-alt"], env=_clean_env(),
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = _last_json(p.stdout)
    assert line["n_gpus"] == 2 and line["steps"] == 40 and line["launch"] == "many"
    assert line["value"] > 0 and line["config"]["parallelism"] == "env-shard x2"


@pytest.mark.gpu
@pytest.mark.parametrize("launch", ["many", "seq", "graph", "eager"])
def test_precompiled_timed_region_steps_the_env_like_run(launch):
    """bench.EnvRun.compile (the list of bound calls the timed region executes) advances the env exactly like EnvRun.run and like
    plain step() calls on the same action rows: the benchmark times real steps, in every launch form."""
    import numpy as np
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path.insert(0, ROOT)
    import bench
    from drl_uav_cellularnet_amd import BatchedMobiEnv

    env = BatchedMobiEnv(300, nBS=4, nUE=20, grid_n=100, max_step=130)
    ref = env.clone()
    pool = torch.randint(0, 625, (400, 300), generator=torch.Generator().manual_seed(5), dtype=torch.int64).to(env.device)
    r = bench.EnvRun(env, launch, pool)
    W, K = 30, 170                                                    # crosses a reset (MAXSTEP 130) and several chunk borders
    r.prepare([n for n, _ in r.plan(W)] + [n for n, _ in r.plan(K, t0=(r.t + W) % r.max_step)])
    r.run(W)
    r.stage(K)
    for f in r.compile(K):
        f()
    t = 0
    for i in range(W + K):
        ref.step(pool[i])
        t += 1
        if t == 130:
            ref.reset()
            t = 0
    torch.cuda.synchronize()
    assert np.array_equal(env.get_state(), ref.get_state())
    assert r.t == t


def test_traffic_entries_are_used_only_for_their_own_dispatch_form():
    """bench.committed_counters: a committed PMC entry describes ONE dispatch form -- (kernel, batch, shape, steps
    per call, schedule) -- and is never scaled onto another one: a run of any other form gets None and the reason."""
    sys.path.insert(0, ROOT)
    import json

    import bench

    with open(os.path.join(ROOT, "profiles", "traffic_current.json")) as f:
        entries = json.load(f)["entries"]
    assert entries and all({"kernel", "envs", "n_bs", "n_ue", "steps_per_launch", "schedule", "dispatches_per_call"} <= set(e) for e in entries)
    keys = [(e["envs"], e["n_bs"], e["n_ue"], e["kernel"], e["steps_per_launch"], e["schedule"]) for e in entries]
    assert len(set(keys)) == len(keys)                                   # one entry per form
    assert all(abs(e["dispatches_per_call"] - 1.0) < 0.05 for e in entries)   # every committed form is one dispatch per call
    e = next(x for x in entries if x["schedule"] == "one_launch_rotation" and x["steps_per_launch"] == 100)
    got, why = bench.committed_counters(e["envs"], e["n_bs"], e["n_ue"], e["kernel"], 100, "one_launch_rotation")
    assert got == e and why is None
    for spl, sched in ((100, "plain"), (50, "one_launch_rotation"), (20, "plain")):
        got, why = bench.committed_counters(e["envs"], e["n_bs"], e["n_ue"], e["kernel"], spl, sched)
        assert got is None and "dispatch form" in why, (spl, sched, why)
    got, why = bench.committed_counters(12345, 4, 20, e["kernel"], 100, "one_launch_rotation")
    assert got is None and "batch size" in why
    # the driver's call (20-step launches) and its plain counterpart are both committed, under their own kernels
    forms = {(x["steps_per_launch"], x["schedule"]) for x in entries if x["envs"] == 4096 and x["n_ue"] == 20}
    assert {(100, "one_launch_rotation"), (20, "one_launch_rotation"), (20, "plain"), (1, "plain")} <= forms
