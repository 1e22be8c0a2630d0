#!/usr/bin/env python3
"""bench.py -- env steps/s of the HIP hot path (BASELINE.json metric), one JSON line on rank 0.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--envs E] [--grid G] [--launch many|seq|graph|eager] [--mode env|a2c]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one batched MobiEnvironment.step() over all envs of a rank (mobile_env.py:150-194).
Workload at N=1: BASELINE.json configs[1] -- 4096 envs, 4 UAV x 20 UE (groups 5,5,5,5), G=100, on-device Philox randomness,
uniform random joint actions resident in HBM before the timed region, compact outputs only (no dense observation).
N>1: the same workload per rank (weak scaling); env instances are independent, so there is NO data-path collective in the
env bench -- only the barrier / max-reduce of the timing.  `python bench.py --gpus N` WITHOUT a launcher starts its own N
ranks (one process per GPU, spawned before anything touches the GPU) and fails loudly if fewer than N joined.

--launch  how the K steps reach the GPU.  Every form computes the same steps bit for bit (tests/test_step_many_gpu.py) and writes
          all nine outputs of every step; the line reports the other forms under "other_launch_forms":
            many   (default) uavenv_step_many: <= 100 consecutive steps per launch, walker / group / UAV state carried in registers
                   between them (the benchmark's actions are resident in HBM and do not depend on observations, which is the
                   case this entry point exists for).  A 4096-env single-step kernel lasts 8 us whatever launches it, 5.8 us of it
                   outside its arithmetic (kernarg fetch, state load round trip, store drain: DESIGN.md section 4).
            seq    uavenv_step_seq: one kernel per step, the launches of <= 100 steps issued by ONE C call
            graph  hipGraph replay of chunks of <= 100 captured uavenv_step launches (same steady state; a replay costs 10-20 us
                   of host latency before the first kernel, which a 20-step run feels)
            eager  one ctypes call per step (round 1's bench)
--mode a2c  BASELINE configs[2] (N=1) / configs[3] (N=8): 8192 envs per GPU, MLP actor-critic, 50-step rollouts, one update per
            rollout with ONE flat RCCL all-reduce of the 80.8 MB gradient inside the timed region (a2c_single_thread.py:107-133).
            The default (env) run appends the same measurement as the "a2c" object of its line unless --no-a2c.
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_BS, N_UE, GRID, GROUPS = 4, 20, 100, [5, 5, 5, 5]
SEED = 0x5EED
HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
SIMD_ISSUE_HZ = 2.4e9 / 4.0   # one VALU wave-instruction per 4 cycles per SIMD at 2.4 GHz (MI355X_MICROARCH.md cycle table)
CHUNK = 100                   # steps per captured graph / per uavenv_step_many launch (divides MAXSTEP = 2000)
A2C_ENVS, A2C_ROLLOUT = 8192, 50
PREWARM_STEPS = 400           # untimed steps on a scratch env right before the headline measurement (clock ramp; see measure_env)
CPU_THREAD_CAP = 16           # cpu_baseline threads: the CPU share of a one-GPU job on this pool (stated in the line)


SCHEDULE_NAMES = {0: "plain", 1: "one_launch_rotation"}   # uavenv_debug_rotation_info: launches per call


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


def committed_counters(envs, n_bs, n_ue, kernel, steps_per_call, schedule):
    """Per-CALL PMC figures of a step kernel from the COMMITTED rocprofv3 passes (profiles/traffic_current.json): bench.py
    cannot run the profiler on itself, so these are constants of the profiled build, labelled as such in the line.  `kernel` is the
    instantiation the timed region launched, as the library's launch census names it.  An entry describes ONE dispatch form: it is used
    only when batch size, shape, kernel, steps per call AND schedule (plain launch / one-launch rotation / several launches) are those of
    this run -- nothing is scaled from another form; otherwise -> (None, why)."""
    path = os.path.join(ROOT, "profiles", "traffic_current.json")
    try:
        with open(path) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None, "profiles/traffic_current.json missing or unreadable"
    near = []
    for e in t.get("entries", []):
        if (e.get("envs"), e.get("n_bs"), e.get("n_ue"), e.get("kernel")) == (envs, n_bs, n_ue, kernel):
            if (e.get("steps_per_launch"), e.get("schedule", "plain")) == (steps_per_call, schedule):
                return e, None
            near.append("%s steps per call, %s" % (e.get("steps_per_launch"), e.get("schedule", "plain")))
    if near:
        return None, ("no committed PMC pass of this dispatch form (%d steps per call, %s); committed for this kernel and batch: %s"
                      % (steps_per_call, schedule, "; ".join(near)))
    return None, "no committed PMC pass of this kernel at this batch size and shape"


def launched_kernels(before, after):
    """Names of the env-kernel instantiations launched between two census snapshots (step kernels only), most launches first."""
    d = [(a[2] - b[2], a[0]) for a, b in zip(after, before) if a[2] > b[2] and ("STEP" in a[0])]
    return [name for _, name in sorted(d, reverse=True)]


def cpu_baseline(target_seconds=12.0):
    """the CPU oracle (oracle/, kind 'port': scalar float64 C restatement of the reference's step(),
    same loop structure as channel.py:249-269) timed on the box's host cores on a bounded sample of the same workload:
    (a) ONE env on ONE core -- the like-for-like stand-in for one reference process (the reference itself, NumPy, measured
    ~670 steps/s at this shape in the build container); (b) one 64-env shard per thread on all usable cores."""
    import numpy as np

    from oracle import oracle as O

    cfg = O.make_config(N_BS, N_UE, GRID, groups=GROUPS)
    rs = np.random.RandomState(1234)
    # (a) N = 1, single core
    one = O.OracleEnv(cfg, 1, seed=SEED)
    one.construct()
    acts1 = rs.randint(0, 625, size=(4096, 1)).astype(np.int64)
    n1, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < target_seconds / 4.0:
        for t in range(512):
            one.step(acts1[(n1 + t) % 4096])
        n1 += 512
    single = n1 / (time.perf_counter() - t0)
    # (b) all usable cores: what this process may run on (affinity mask, cgroup CPU quota), not what the host has.  A GPU box
    # of this pool shows 256 host CPUs to every tenant but gives a one-GPU job a 16-CPU share (r02a: 64 threads ran 4x slower
    # than calibrated), and neither the mask nor cpu.max says so -- hence the explicit cap, stated in `sample`.
    host_cpus = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = host_cpus
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            usable = max(1, min(usable, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    threads = max(1, min(usable, CPU_THREAD_CAP))
    per = 64
    envs = [O.OracleEnv(cfg, per, seed=SEED, env_id_base=i * per) for i in range(threads)]
    for e in envs:
        e.construct()
    acts = rs.randint(0, 625, size=(256, per)).astype(np.int64)
    t0 = time.perf_counter()
    for t in range(20):
        envs[0].step(acts[t])
    dt = (time.perf_counter() - t0) / 20
    steps = int(max(50, min(60000, 0.75 * target_seconds / max(dt, 1e-9))))

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
    return {"value": threads * per * steps / el, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "host_cpus": host_cpus, "usable_cpus": usable,
            "single_core_n1": {"value": single, "unit": "env-steps/s", "cores": 1,
                               "sample": "1 env x %d steps of oracle step() on one thread" % n1},
            "thread_cap": CPU_THREAD_CAP,
            "sample": "%d threads (min(usable CPUs, cap %d = one-GPU CPU share of the pool)) x %d envs x %d steps of oracle "
                      "step() (4 UAV x 20 UE, G=100, Philox), %.1f s" % (threads, CPU_THREAD_CAP, per, steps, el)}


# ----------------------------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` with no launcher around it
# ----------------------------------------------------------------------------------------------------------------------
def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n, argv):
    """Start n ranks of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment) and wait for them.
    Runs in a parent that has not touched the GPU (no torch import, no HIP call); rank 0's stdout carries the JSON line.
    Exit code: 0 only if EVERY rank exited 0 -- a rank that never joined makes the others fail in init / the rank census."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), UAVENV_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    deadline = time.time() + float(os.environ.get("UAVENV_BENCH_TIMEOUT", "1500"))
    pending = list(enumerate(procs))
    while pending:
        for r, p in list(pending):
            code = p.poll()
            if code is not None:
                pending.remove((r, p))
                if code != 0:
                    print("bench.py: rank %d exited with code %d" % (r, code), file=sys.stderr)
                    rc = rc or code or 1
        if rc and pending:              # one rank failed: the others would wait in a collective until the timeout
            time.sleep(5.0)
            for _, p in pending:
                if p.poll() is None:
                    p.terminate()
            for _, p in pending:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        if time.time() > deadline:
            print("bench.py: timeout, terminating the ranks", file=sys.stderr)
            for _, p in pending:
                p.kill()
            rc = rc or 124
            break
        time.sleep(0.05)
    return rc


# ----------------------------------------------------------------------------------------------------------------------
def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs", type=int, default=None, help="env instances per GPU (default 4096; 8192 with --mode a2c)")
    ap.add_argument("--mode", choices=("env", "a2c"), default="env")
    ap.add_argument("--launch", choices=("many", "seq", "graph", "eager"), default="many")
    ap.add_argument("--chunk", type=int, default=CHUNK, help="steps per uavenv_step_many launch / per captured graph (default 100; profiling "
                    "runs use 20 to repeat the launch shape of the driver's --steps 20 call)")
    ap.add_argument("--grid", type=int, default=GRID, help="grid cells per side (100 = every reference script; 200 = the class default, "
                    "mobile_env.py:37; the default run also reports G = 200 under other_grids)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-a2c", action="store_true", help="env mode: skip the appended A2C / gradient all-reduce measurement")
    ap.add_argument("--no-alt", action="store_true", help="env mode: skip the secondary eager / step_many measurements")
    ap.add_argument("--a2c-rollouts", type=int, default=4, help="timed rollouts (+1 untimed) of the appended A2C measurement")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to "
                    "rehearse the multi-rank path with several ranks sharing one GPU, or with --rehearse-launcher on CPU)")
    ap.add_argument("--force-device", type=int, default=None, help="rehearsal only: every rank uses this GPU")
    ap.add_argument("--force-dist", action="store_true", help="rehearsal only: initialise the process group even at "
                    "world size 1 (exercises the RCCL init / barrier / max-reduce path on a one-GPU box)")
    ap.add_argument("--rehearse-launcher", action="store_true", help="no GPU work at all: ranks rendezvous (gloo), run the rank "
                    "census / barrier / max-reduce and rank 0 prints a line marked rehearsal (value null).  CPU test of the launcher")
    ap.add_argument("--n-bs", type=int, default=N_BS, help="secondary measurements only (default = BASELINE workload)")
    ap.add_argument("--n-ue", type=int, default=N_UE, help="secondary measurements only (default = BASELINE workload)")
    return ap.parse_args(argv)


def init_dist(args, rank, local_rank, world):
    """-> (dist module or None, device index).  One process per GPU; backend nccl = RCCL over xGMI."""
    if not (world > 1 or args.force_dist):
        return None, local_rank
    import datetime

    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    if args.force_device is not None:
        local_rank = args.force_device
    tmo = datetime.timedelta(seconds=300)
    if args.rehearse_launcher:
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=tmo)
        return dist, local_rank
    torch.cuda.set_device(local_rank)
    if args.backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank), timeout=tmo)
    else:
        dist.init_process_group(args.backend, rank=rank, world_size=world, timeout=tmo)
    return dist, local_rank


def rank_census(dist, dev):
    """Number of ranks that actually take part in a reduction (sum of ones)."""
    if dist is None:
        return 1
    import torch

    one = torch.ones(1, dtype=torch.float64, device=dev)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    return int(round(float(one.item())))


class EnvRun:
    """K steps of the env workload in segments of <= CHUNK steps, reset() every MAXSTEP steps as the reference's callers do
    (main.py:205-211).  Before each segment the next CHUNK action rows are copied from the resident pool into a tape the
    graph / the multi-step kernel reads (3.3 MB device-to-device per 100 steps at 4096 envs)."""

    def __init__(self, env, launch, pool, chunk=CHUNK):
        import torch

        self.env, self.launch, self.pool, self.chunk = env, launch, pool, int(chunk)
        self.tape = torch.empty((self.chunk, env.n_envs), dtype=torch.int64, device=env.device)
        self.graphs = {}
        self.many_out = {}
        self.max_step = int(env.cfg.max_step)
        self.t = 0            # steps since the last reset (the constructor's reset counts as one)
        self.cursor = 0       # next pool row

    def prepare(self, sizes):
        """Capture / allocate for every segment length that will occur (outside the timed region; executes no step)."""
        import torch

        for n in sorted(set(sizes)):
            if self.launch == "many":
                self.env.prepare_step_many(n)      # (the launch schedule of this segment length: built here, not inside the timed call)
            if self.launch == "graph" and n not in self.graphs:
                self.graphs[n] = self.env.capture_steps(self.tape[:n])
            if self.launch == "many" and n not in self.many_out:
                self.many_out[n] = {k: torch.empty((n,) + tuple(v.shape), dtype=v.dtype, device=v.device)
                                    for k, v in self.env.out.items()}
    def plan(self, n_steps, t0=None):
        """Segment lengths for n_steps more steps, cut at reset boundaries: [(n, reset_after)]."""
        t = self.t if t0 is None else t0
        segs = []
        while n_steps > 0:
            n = min(self.chunk, n_steps, self.max_step - t)
            t += n
            n_steps -= n
            segs.append((n, t == self.max_step))
            if t == self.max_step:
                t = 0
        return segs

    def stage(self, n_steps):
        """Copy the action rows of the NEXT segment into the tape now (before a timed region starts: the benchmark's inputs
        are resident when the clock starts); run() then skips the copy for that segment."""
        segs = self.plan(n_steps)
        if segs:
            self._fill(segs[0][0])
            self._staged = True

    def _fill(self, n):
        if self.cursor + n > self.pool.shape[0]:
            self.cursor = 0
        self.tape[:n].copy_(self.pool[self.cursor:self.cursor + n])
        self.cursor += n

    def compile(self, n_steps):
        """The next n_steps as a list of zero-argument callables built NOW (outside the timed region): the timed loop then is a
        handful of C calls -- tape copies on pre-sliced views, the launch itself bound with its arguments (ctypes / graph replay),
        resets -- with no planning, validation or attribute lookups in between.  Advances the bookkeeping like run()."""
        import ctypes as C
        import functools

        env, prog = self.env, []
        stream = C.c_void_p(env._stream())
        for n, reset_after in self.plan(n_steps):
            if getattr(self, "_staged", False):
                self._staged = False
            else:
                if self.cursor + n > self.pool.shape[0]:
                    self.cursor = 0
                prog.append(functools.partial(self.tape[:n].copy_, self.pool[self.cursor:self.cursor + n]))
                self.cursor += n
            if self.launch == "graph":
                prog.append(self.graphs[n].replay)
            elif self.launch == "seq":
                prog.append(functools.partial(env._lib.uavenv_step_seq, env._h, self.tape.data_ptr(), n, env._out_ref, stream))
            elif self.launch == "many":
                st = env.out_struct_for(self.many_out[n])
                self._keep_structs = getattr(self, "_keep_structs", []) + [st]
                prog.append(functools.partial(env._lib.uavenv_step_many, env._h, self.tape.data_ptr(), n, C.byref(st), stream))
            else:
                prog.extend(functools.partial(env.step, self.tape[t]) for t in range(n))
            self.t += n
            if reset_after:
                prog.append(env.reset)
                self.t = 0
        return prog

    def run(self, n_steps):
        env = self.env
        for n, reset_after in self.plan(n_steps):
            if getattr(self, "_staged", False):
                self._staged = False
            else:
                self._fill(n)
            if self.launch == "graph":
                self.graphs[n].replay()
            elif self.launch == "seq":
                env.step_seq(self.tape[:n])
            elif self.launch == "many":
                env.step_many(self.tape[:n], out=self.many_out[n], refresh_out=False)
            else:
                for t in range(n):
                    env.step(self.tape[t])
            self.t += n
            if reset_after:
                env.reset()
                self.t = 0


def timed(fn, dist, dev, keep_busy=None, events=True):
    """synchronize + barrier, clock, K steps, synchronize, clock, barrier; HIP events on the launch stream inside -> (wall s, gpu ms).
    keep_busy: untimed device work (on a scratch env) queued right before the opening synchronize, so that the region starts microseconds
    after the device was last busy -- a 0.1-ms region that starts on a device left idle for milliseconds of host work measures the clock
    ramp, not the kernel."""
    import torch

    import gc

    gc.disable()                                # no collector pause inside a region that may last 0.1 ms (and no gc.collect() here:
                                                # tens of ms of host work would let the device idle and clock down before the region)
    if keep_busy is not None:
        keep_busy()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    # events=False: nothing but the K steps between the two clock readings.  A pair of HIP events around a 20-step call -- two marker
    # packets on the launch stream, or start / stop events on the dispatch itself -- costs the region 9-10 us of ~110 (tools/region_overhead.py,
    # profiles/r04o_region_overhead.json); the caller then takes the kernel's duration from a repeat of the same launches (measure_env).
    ev0, ev1 = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if events else (None, None)
    t0 = time.perf_counter()
    if events:
        ev0.record()
    fn()
    if events:
        ev1.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0      # this rank's K steps are complete; the slowest rank's time is what is reported (MAX)
    gc.enable()
    if dist is not None:
        dist.barrier()                      # closing bracket: nobody leaves before everybody has finished (not part of any rank's clock:
    return elapsed, (ev0.elapsed_time(ev1) if events else None)   # an RCCL barrier costs tens of microseconds, a 20-step region lasts 0.1 ms)


def measure_env(args, env, launch, K, W, dist, dev, reduce_dev, rank, scratch=None):
    import torch

    from drl_uav_cellularnet_amd.sharding import gather_over_ranks, max_over_ranks

    gen = torch.Generator(device="cpu").manual_seed(1234 + rank)
    ck = int(args.chunk)
    n_pool = max(ck, min(((K + W + ck - 1) // ck) * ck, 5 * max(ck, CHUNK)))    # action pool resident in HBM, cycled
    pool = torch.randint(0, min(env.action_space_dim, 2 ** 62), (n_pool, env.n_envs), generator=gen, dtype=torch.int64).to(dev)
    r = EnvRun(env, launch, pool, chunk=ck)
    sizes = [n for n, _ in r.plan(W)] + [n for n, _ in r.plan(K, t0=(r.t + W) % r.max_step)]
    r.prepare(sizes)
    if scratch is not None:
        # Device pre-warm on a SCRATCH env of the same shape (not the measured one): graph capture and env construction leave
        # the GPU idle for milliseconds, and a short timed region (the driver's --steps 20 is 0.2 ms) would otherwise sit in
        # the clock ramp.  PREWARM_STEPS eager steps, untimed, immediately before the measured env's own W warm-up steps.
        for t in range(PREWARM_STEPS):
            scratch.step(pool[t % n_pool])
        if launch == "many":
            # ... and the scratch env through the SAME launch form and call sizes as the measured one: a multi-step call of another size
            # may take another code path of the kernel (a rotation schedule's pieces, section 4d), and the first execution of a path pays
            # its instruction fetch from memory -- 20-30 us, which a 20-step timed region (0.1 ms) would otherwise carry
            rs = EnvRun(scratch, launch, pool, chunk=ck)
            rs.prepare(sizes)
            for n in sorted(set(sizes)):
                for _ in range(3):
                    rs.run(n)
    r.run(W)
    r.stage(K)
    prog = r.compile(K)

    results = []

    def go():
        for f in prog:
            results.append(f())

    from drl_uav_cellularnet_amd import _capi

    census0 = _capi.launch_census()
    busy = None
    if scratch is not None:
        def busy():                              # ~0.5 ms of single steps on the scratch env (the measured env is not touched)
            for t in range(64):
                scratch.step(pool[t % n_pool])
    many = launch == "many"
    elapsed, gpu_ms = timed(go, dist, dev, keep_busy=busy, events=not many)
    kernels = launched_kernels(census0, _capi.launch_census())          # the instantiation(s) the timed region really launched
    if many:                                                             # (the keep-busy single steps on the scratch env fall between the two
        kernels = [k for k in kernels if "MANY=1" in k]                  #  census snapshots: they are not the region's)
    elif busy is not None:
        kernels = [k for k in kernels if "MANY=0" in k]
    bad = [r for r in results if isinstance(r, int) and r != 0]          # return codes of the bound C-ABI launches
    if bad:
        _capi.check(bad[0])
    launch_us = None
    if many:
        # The multi-step kernel's own duration: the SAME K steps once more, every dispatch carrying its start / stop events
        # (uavenv_launch_timing: hipExtLaunchKernelGGL, the dispatch packet's timestamps) -- live in this process, on the launch stream,
        # but outside the wall-clock region above, which therefore holds nothing but the K steps.
        env.launch_timing(True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for f in r.compile(K):
            f()
        e1.record()
        torch.cuda.synchronize()
        launch_us = env.launch_times_us()
        env.launch_timing(False)
        if launch_us:
            gpu_ms = sum(launch_us) * 1e-3                               # device time of the K steps' launches (tape copies / resets excluded)
        else:                                                            # (a multi-pass handle runs a multi-step call as single-step launches:
            gpu_ms, launch_us = e0.elapsed_time(e1), None                #  no multi-step dispatch to time -- stream events around the repeat)
    per_rank = gather_over_ranks([elapsed, gpu_ms], device=reduce_dev)   # every rank's own clock: a straggler must be visible
    elapsed, gpu_ms = max_over_ranks([elapsed, gpu_ms], device=reduce_dev)          # slowest rank
    sched = None
    if launch == "many":                                                   # how the library runs a call of this many steps on this handle
        import ctypes as C

        nl, sl = C.c_int(0), C.c_longlong(0)
        env._lib.uavenv_debug_rotation_info(env._h, min(ck, K), C.byref(nl), C.byref(sl))
        sched = {"form": SCHEDULE_NAMES.get(nl.value, "?"), "dispatches_per_call": max(1, nl.value), "wavefronts_per_dispatch": sl.value or None}
    return elapsed, gpu_ms, {"kernels": kernels, "per_rank_elapsed_s": [p[0] for p in per_rank],
                             "per_rank_gpu_ms": [p[1] for p in per_rank], "schedule": sched, "device_error": env.device_error(),
                             "launch_us": launch_us}


def measure_a2c(args, dist, dev, reduce_dev, rank, world, envs, rollouts, rollout_len=A2C_ROLLOUT):
    """`rollouts` timed rollouts (after one untimed) of synchronous A2C: policy + sampling + env step per time step, then ONE
    update over all T*N samples with one flat gradient all-reduce (RCCL when world > 1) and the TF1-semantics RMSProp step."""
    import torch

    from drl_uav_cellularnet_amd import BatchedMobiEnv
    from drl_uav_cellularnet_amd.agent import A2CRunner, grad_allreduce_bytes
    from drl_uav_cellularnet_amd.sharding import gather_over_ranks, max_over_ranks, shard_for_rank, whole_job_rate

    base, _ = shard_for_rank(rank, world, envs)
    env = BatchedMobiEnv(envs, nBS=N_BS, nUE=N_UE, grid_n=GRID, groups=GROUPS, device=dev, seed=SEED, env_id_base=base)
    runner = A2CRunner(env, rollout=rollout_len, tune_gemms=True)     # opt-in: the shipped TunableOp picks for the config-3 GEMM shapes
    runner.train_rollout()                                         # untimed: allocator, rocBLAS / RCCL first use, graph capture
    prof = {"collect": 0.0, "update": 0.0}

    def go():
        for _ in range(rollouts):
            a = time.perf_counter()
            batch = runner.collect()
            torch.cuda.synchronize()
            b = time.perf_counter()
            runner.update(*batch)
            torch.cuda.synchronize()
            prof["collect"] += b - a
            prof["update"] += time.perf_counter() - b

    elapsed, _ = timed(go, dist, dev)
    per_rank = [p[0] for p in gather_over_ranks([elapsed], device=reduce_dev)]     # every rank's own clock (a straggler must be visible)
    elapsed = max_over_ranks([elapsed], device=reduce_dev)[0]
    n = envs * rollout_len * rollouts
    st = dict(runner.stats)
    roof = None
    if True:                                                       # every rank: the extra update ends in the same collective(s) as a timed one
        try:
            roof = a2c_roofline(runner, envs, rollout_len)
        except Exception as ex:                                    # an annotation: never a reason to lose the measurement
            roof = {"error": "%s: %s" % (type(ex).__name__, ex)}
    ar_ms, ar_bytes = st.get("allreduce_ms"), grad_allreduce_bytes(runner.net)
    ov_ms, buckets = st.get("allreduce_overlapped_ms"), st.get("allreduce_buckets")
    f = 2.0 * (world - 1) / world
    busbw = busbw_buckets = None
    if world > 1 and ar_ms:
        if buckets and ov_ms:     # two buckets: the critic trunk's (hidden behind the actor's backward pass) and the actor trunk's (exposed)
            busbw = f * sum(buckets) / ((ov_ms + ar_ms) * 1e-3) / 1e9
            busbw_buckets = [f * buckets[0] / (ov_ms * 1e-3) / 1e9, f * buckets[1] / (ar_ms * 1e-3) / 1e9]
        else:
            busbw = f * ar_bytes / (ar_ms * 1e-3) / 1e9
    return {"metric": "A2C end-to-end env steps/sec (policy + sampling + env step + update incl. gradient all-reduce)",
            "value": whole_job_rate(n, world, elapsed), "unit": "env-steps/s", "n_gpus": world, "envs_per_gpu": envs,
            "rollout_len": rollout_len, "rollouts": rollouts, "ms_per_rollout": elapsed / rollouts * 1e3,
            "collect_ms_per_rollout": prof["collect"] / rollouts * 1e3, "update_ms_per_rollout": prof["update"] / rollouts * 1e3,
            "per_rank_elapsed_s": per_rank,
            "allreduce_ms_per_update": ar_ms, "grad_allreduce_bytes": ar_bytes,
            "allreduce_busbw_GBps": busbw, "allreduce_busbw_per_bucket_GBps": busbw_buckets,
            "allreduce_busbw_note": "2 (N-1)/N x bytes / time of the last update's collective(s) (device events); with two buckets the time is "
            "hidden + exposed (allreduce_overlapped_ms + allreduce_ms_per_update: as if they ran one after the other) and the per-bucket figures "
            "use each bucket's own time; null at 1 rank.  xGMI ring: 7 links x ~153 GB/s per GPU (MI355X_MICROARCH.md)",
            "allreduce_overlapped_ms": st.get("allreduce_overlapped_ms"), "allreduce_buckets": st.get("allreduce_buckets"),
            "allreduce": (("RCCL (nccl backend), " + ("two buckets per update: critic trunk on a side stream behind the actor's backward pass, then "
                           "the actor trunk" if buckets else "one flat bucket per update")) if (world > 1 and args.backend == "nccl")
                          else ("%s backend (rehearsal)" % args.backend if world > 1 else "none (1 rank)")),
            "collect_launch": getattr(runner, "collect_launch", "eager"), "gemm_tuning": bool(getattr(runner, "gemm_tuning", False)),
            "a_loss": st.get("a_loss"), "c_loss": st.get("c_loss"), "mean_reward": st.get("mean_reward"),
            "roofline": roof, "pipeline_halves": getattr(runner, "_halves", None) is not None,
            "persistent_rollout": bool(getattr(runner, "_persistent", False)),
            "config": "%d envs/GPU x 4 UAV x 20 UE, MLP 50000->200->200->{625,1}, fp32, %d-step rollouts, 1 update per rollout"
                      % (envs, rollout_len)}


MFMA_F32_PEAK_TFLOPS = 157.3     # dense v_mfma_f32_16x16x4_f32 peak (MI355X_MICROARCH.md; the learner computes in float32 like the reference)
GATHER_CACHE_TBPS = 8.6          # MI355X_MICROARCH.md "Indexed rows": 38 MB table, uniformly random rows (Infinity Cache) 8.6 TB/s chip-wide
GATHER_LARGE_TBPS = 7.9          # same table: 151 MB table, uniformly random rows 7.4-7.9 TB/s chip-wide (upper figure)


def a2c_roofline(runner, envs, T):
    """Per dominant learner kernel: time from HIP events around each launch of ONE extra eager rollout + update of this run
    (_agent_capi.profile_begin / profile_end; the timed region itself replays a hipGraph and cannot be bracketed per kernel), its
    algorithmic flops or bytes, the roof that bounds it and the achieved fraction.  Roofs: 157.3 TFLOP/s float32 MFMA; 8 TB/s HBM for
    streaming kernels; for the two indexed-row kernels the fabric's gather rate the guide measured for a table of that size (both
    cited in `roof_source`) -- their rows come out of the Infinity Cache / across the fabric, not out of HBM at its peak."""
    import torch

    from drl_uav_cellularnet_amd import _agent_capi as A

    N, H, NA, K = envs, 200, runner.net.n_action, runner.env.nBS + runner.env.nUE
    M = N * T
    parts = getattr(runner, "_halves", None) or ((0, N),)      # pipelined halves: the rollout's kernels run per half, on two streams
    n_parts = len(parts)
    Nr = (parts[0][1] - parts[0][0])                           # rows per rollout-kernel launch (the parts are equal at 8192 envs)
    was = runner.collect_launch
    # (the timed region may have collected with the two persistent rollout kernels -- A2CRunner(persistent_rollout) --, whose launches last a
    #  whole rollout each and wait for each other; the per-kernel figures below are of the per-step kernels, which compute the same rows)
    was_persistent = bool(getattr(runner, "_persistent", False))
    pair_ms = None
    if was_persistent:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        runner.u_buf.uniform_(generator=runner.gen)
        runner._refresh_transposed()
        torch.cuda.synchronize()
        e0.record()
        runner._rollout_steps()                      # first layer of the start state, gates, the two persistent launches on two streams, join
        e1.record()
        torch.cuda.synchronize()
        pair_ms = e0.elapsed_time(e1)
        runner._persistent = False
    runner.collect_launch = "eager"
    torch.cuda.synchronize()
    A.profile_begin()
    try:
        batch = runner.collect()
        runner.update(*batch)
    finally:
        times = A.profile_end()
        runner.collect_launch = was
        runner._persistent = was_persistent
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    act = runner.act_buf[0]
    e0.record()
    for _ in range(T):
        runner.env.step(act, reward_out=runner.rew_buf[0])
    e1.record()
    torch.cuda.synchronize()
    env_us = e0.elapsed_time(e1) * 1e3 / T

    def entry(keys, work, unit, roof, roof_src, what, calls_per_rollout):
        ms = [m for k in sorted(set(keys)) for m in times.get(k, [])]
        if not ms:
            return None
        avg_us = sum(ms) / len(ms) * 1e3
        rate = work / (avg_us * 1e-6) / 1e12                     # TFLOP/s or TB/s
        return {"what": what, "calls_per_rollout": calls_per_rollout, "avg_us": avg_us, "ms_per_rollout": avg_us * calls_per_rollout * 1e-3,
                "work_per_call": work, "work_unit": "flop" if unit == "TFLOP/s" else "bytes", "achieved": rate, "unit": unit,
                "peak": roof, "frac": rate / roof, "roof_source": roof_src, "launches_timed": len(ms)}

    mfma = ("dense float32 MFMA peak (MI355X_MICROARCH.md)", MFMA_F32_PEAK_TFLOPS)
    out = {}
    rows = [
        ("actor_head", ["uavagent_actor_head_f32[rows=%d]" % (hi - lo) for lo, hi in parts], 2.0 * Nr * (H * H + H * NA), "TFLOP/s", mfma[1], mfma[0],
         "rollout step: layer 2 + policy head + action draw, one kernel (main.py:147-150,165-169)%s" % (
             "; per HALF of the batch, timed beside the other half's gather / env step" if n_parts > 1 else ""), T * n_parts),
        ("first_layer_gather", ["uavagent_first_layer_from_obs_f32[rows=%d]" % (hi - lo) for lo, hi in parts], float(Nr) * K * 2 * H * 4, "TB/s", GATHER_CACHE_TBPS,
         "gather of 1 600-byte row pairs out of two 40 MB tables: MI355X_MICROARCH.md 'Indexed rows', 38 MB table 8.6 TB/s chip-wide",
         "rollout step: first layer of both trunks from the compact observation (sum of B + U table rows per env)%s" % (
             "; per HALF of the batch, timed beside the other half's actor head" if n_parts > 1 else ""), (T - 1) * n_parts),
        ("table_gradient", ["uavagent_rows_grad_sums_f32[M=%d,K=%d]" % (M, K)], float(M) * K * 2 * H * 4, "TB/s", GATHER_LARGE_TBPS,
         "gather of one 1 600-byte g row per (sample, index) pair from a 655 MB array: MI355X_MICROARCH.md 'Indexed rows', 151 MB table "
         "7.4-7.9 TB/s chip-wide", "update: x^T g for the 0/1 first-layer input (both tables)", 1),
        ("dW_policy_head", ["uavagent_gemm_tn_f32[M=%d,I=%d,J=%d]" % (M, H, NA)], 2.0 * M * H * NA, "TFLOP/s", mfma[1], mfma[0], "update: h2a^T dlogits", 1),
        ("dW_200x200", ["uavagent_gemm_tn_f32[M=%d,I=%d,J=%d]" % (M, H, H)], 2.0 * M * H * H, "TFLOP/s", mfma[1], mfma[0], "update: two of them (actor, critic layer 2)", 2),
        ("dX_policy_head", ["uavagent_gemm_rows_f32[M=%d,K=%d,N=%d,relu6_mask]" % (M, (NA + 15) // 16 * 16, H)], 2.0 * M * NA * H, "TFLOP/s", mfma[1], mfma[0],
         "update: dlogits @ W3^T with the relu6 mask fused", 1),
        ("dX_200x200", ["uavagent_gemm_rows_f32[M=%d,K=%d,N=%d,relu6_mask]" % (M, H, H)], 2.0 * M * H * H, "TFLOP/s", mfma[1], mfma[0],
         "update: two of them, relu6 mask + bias gradient fused (W^T resident in LDS)", 2),
        ("forward_critic_layer2", ["uavagent_gemm_rows_f32[M=%d,K=%d,N=%d,bias]" % (M, H, H)], 2.0 * M * H * H, "TFLOP/s", mfma[1], mfma[0],
         "update: the one forward GEMM the rollout did not already compute", 1),
        ("loss_grad", ["uavagent_a2c_loss_grad[M=%d,A=%d]" % (M, NA)], 2.0 * M * NA * 4, "TB/s", HBM_PEAK_GBPS / 1e3, "HBM peak (MI355X_MICROARCH.md)",
         "update: softmax, entropy, both losses and d loss / d logits in one pass (logits read and overwritten)", 1),
    ]
    for name, keys, work, unit, roof, src, what, calls in rows:
        e = entry(keys, work, unit, roof, src, what, calls)
        if e is not None:
            out[name] = e
    b_step = algorithmic_bytes_per_env_step(runner.env.nUE, runner.env.nBS, int(runner.env.cfg.n_groups))
    out["env_step"] = {"what": "rollout step: uavenv_step of all envs (the closed-loop form: one kernel per step)", "calls_per_rollout": T,
                       "avg_us": env_us, "ms_per_rollout": env_us * T * 1e-3, "work_per_call": b_step * N, "work_unit": "bytes (SURVEY 8(d) algorithmic)",
                       "achieved": b_step * N / (env_us * 1e-6) / 1e12, "unit": "TB/s", "peak": HBM_PEAK_GBPS / 1e3,
                       "frac": b_step * N / (env_us * 1e-6) / 1e12 / (HBM_PEAK_GBPS / 1e3), "roof_source": "HBM peak (nominal, as the headline's roofline.frac)",
                       "launches_timed": T}
    covered = sum(v["ms_per_rollout"] for v in out.values())
    pair = None
    if pair_ms is not None:
        flops = 2.0 * N * T * (H * H + H * NA)
        gbytes = float(N) * (T - 1) * K * 2 * H * 4
        pair = {"what": "the rollout the timed region ran: uavagent_actor_head_gated_f32 beside uavenv_rollout_gated, two persistent launches that hand "
                        "16-env blocks to each other through step counters in device memory (per-step kernels above: the same rows, one launch each)",
                "ms_per_rollout": pair_ms, "policy_flop": flops, "policy_TFLOPs": flops / (pair_ms * 1e-3) / 1e12,
                "policy_frac_of_mfma_peak": flops / (pair_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                "encoder_bytes": gbytes, "encoder_TBps": gbytes / (pair_ms * 1e-3) / 1e12,
                "encoder_frac_of_gather_roof": gbytes / (pair_ms * 1e-3) / 1e12 / GATHER_CACHE_TBPS,
                "encoder_traffic_bytes_committed_pmc": 16.0e9 * (N / 8192.0) * ((T - 1) / 49.0),
                "encoder_frac_of_fabric_gather_rate": gbytes / (pair_ms * 1e-3) / 1e12 / 7.4,
                "note": "both kernels run for the whole rollout, so each rate is its work over the PAIR's time; traffic: FETCH_SIZE x 2 of the env kernel "
                        "alone, profiles/r04pc_gated_kernels_alone_pmc_digest.txt (16.0 GB at 8192 x 50, scaled; a --pmc run serialises dispatches, so the "
                        "pair itself cannot be counted); 7.4 TB/s: what this gather pattern reaches chip-wide (DESIGN 10c, 10e)"}
    return {"kernels": out, "persistent_rollout_pair": pair, "ms_per_rollout_covered": covered, "rollout_parts_on_streams": n_parts,
            "ms_per_rollout_covered_note": "sum of avg_us x calls; with the rollout pipelined over two streams the halves' kernels overlap, so the "
                                           "sum exceeds the wall time of a rollout + update",
            "method": "HIP events around every launch of one extra EAGER rollout + update of this run (after the timed region); back-to-back launches "
                      "on one stream, so a pair brackets its kernel plus ~2 us of launch gap; PMC passes of the same kernels: profiles/ (DESIGN 10c/10d)",
            "all_launches_ms": {k: round(sum(v), 4) for k, v in sorted(times.items())}}


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    launched = "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched:
        sys.exit(self_launch(args.gpus, argv))          # parent: spawns the ranks, never touches the GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE %d: refusing to print a line for the wrong rank count" % (args.gpus, world),
                  file=sys.stderr)
        sys.exit(2)

    if args.rehearse_launcher:                          # CPU-only rehearsal of spawn + rendezvous + census + reductions
        import torch

        from drl_uav_cellularnet_amd.sharding import max_over_ranks, shard_for_rank

        dist, _ = init_dist(args, rank, local_rank, world)
        n = rank_census(dist, None)
        base, _ = shard_for_rank(rank, world, 4096)
        t = max_over_ranks([float(rank + 1), float(base)], device=None)
        if n != args.gpus:
            sys.exit(3)
        if rank == 0:
            print(json.dumps({"rehearsal": True, "value": None, "n_gpus": n, "max_rank_plus_1": t[0], "max_env_id_base": t[1],
                              "launcher": "self" if os.environ.get("UAVENV_BENCH_CHILD") else "external"}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    import torch

    dist, dev_index = init_dist(args, rank, local_rank, world)
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    reduce_dev = dev if (dist is not None and args.backend == "nccl") else None
    n_ranks = rank_census(dist, reduce_dev)
    if n_ranks != args.gpus:
        print("bench.py: %d ranks reduced, --gpus %d" % (n_ranks, args.gpus), file=sys.stderr)
        sys.exit(3)

    if args.mode == "a2c":
        envs = args.envs or A2C_ENVS
        rollouts = max(1, args.steps // A2C_ROLLOUT) if args.steps != 2000 else 6      # (2000 = the env bench's default: 6 rollouts)
        res = measure_a2c(args, dist, dev, reduce_dev, rank, world, envs, rollouts)
        if rank == 0:
            line = {"metric": res["metric"], "value": res["value"], "unit": "env-steps/s", "n_gpus": n_ranks,
                    "steps": rollouts * A2C_ROLLOUT, "warmup": A2C_ROLLOUT, "ms_per_step": res["ms_per_rollout"] / A2C_ROLLOUT,
                    "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64 env / f32 learner",
                    "data": "synthetic", "config": {"workload": res["config"], "envs_per_gpu": envs,
                                                    "parallelism": "env-shard x%d + grad all-reduce" % world}, "a2c": res}
            print(json.dumps(line), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    from drl_uav_cellularnet_amd import BatchedMobiEnv
    from drl_uav_cellularnet_amd.sharding import shard_for_rank, whole_job_rate

    E, K, W = args.envs or 4096, args.steps, args.warmup
    env_id_base, _ = shard_for_rank(rank, world, E)   # rank r owns global envs [r*E, (r+1)*E): no env-path collective
    n_bs, n_ue, grid = args.n_bs, args.n_ue, args.grid
    baseline_shape = (n_bs, n_ue, grid) == (N_BS, N_UE, GRID)
    groups = GROUPS if (n_bs, n_ue) == (N_BS, N_UE) else [n_ue // 4] * 3 + [n_ue - 3 * (n_ue // 4)]

    def make_env(g=None):
        return BatchedMobiEnv(E, nBS=n_bs, nUE=n_ue, grid_n=g or grid, groups=groups, device=dev, seed=SEED, env_id_base=env_id_base)

    # Order: the other launch forms first (they bring clocks and caches up), then the headline behind its own scratch-env
    # pre-warm, and the A2C leg LAST: measured on one box, a 20-step headline region ran at 5.6e8 env-steps/s when the A2C leg
    # (TunableOp, BLAS handles, a captured graph, a side stream, several GB of buffers) came before it and at 6.5-6.7e8 when it
    # comes after.  Every measurement has its own env, its own W warm-up steps and its own bracketed timed region.
    alt, grids = {}, {}
    if not args.no_alt:                                # secondary: the other launch forms on the same box, same K / W
        for other in (("eager", "many", "graph", "seq") if world == 1 else ("seq",)):   # N > 1: the one-kernel-per-step form only
            if other != args.launch:
                el, gm, info = measure_env(args, make_env(), other, K, W, dist, dev, reduce_dev, rank)
                alt[other] = {"value": whole_job_rate(E * K, world, el), "unit": "env-steps/s", "us_per_step_wall": el / K * 1e6,
                              "us_per_step_gpu": gm * 1e3 / K, "kernel": (info["kernels"] or [None])[0]}
        if world == 1 and grid == GRID:                # "G=100 ... also report G=200" (the class default, mobile_env.py:37)
            el, gm, info = measure_env(args, make_env(200), args.launch, K, W, dist, dev, reduce_dev, rank)
            grids["200"] = {"value": whole_job_rate(E * K, world, el), "unit": "env-steps/s", "us_per_step_gpu": gm * 1e3 / K,
                            "launch": args.launch, "note": "same workload on a 200 x 200 grid: compact outputs do not grow with G (the dense "
                            "observation would: 800 000 B per env-step instead of 200 000)"}
    elapsed, gpu_ms, info = measure_env(args, make_env(), args.launch, K, W, dist, dev, reduce_dev, rank, scratch=make_env())
    a2c = None
    if not args.no_a2c and baseline_shape:
        try:
            a2c = measure_a2c(args, dist, dev, reduce_dev, rank, world, A2C_ENVS, args.a2c_rollouts)
        except Exception as ex:                        # the headline line must survive a failure of the appended measurement
            a2c = {"error": "%s: %s" % (type(ex).__name__, ex)}

    if rank == 0:
        per_step_s = gpu_ms * 1e-3 / K   # average per-step device time (HIP events on the launch stream around the timed region)
        b_step = algorithmic_bytes_per_env_step(n_ue, n_bs, len(groups))
        achieved = b_step * E / per_step_s / 1e9
        many = args.launch == "many"
        kernel = (info["kernels"] or ["?"])[0]                  # the instantiation the timed region launched (library's launch census)
        spl = min(int(args.chunk), K) if many else 1            # steps one CALL of the step entry point processes
        schedule = (info.get("schedule") or {}).get("form", "plain") if many else "plain"
        cnt, why_not = committed_counters(E, n_bs, n_ue, kernel, spl, schedule)
        n_simd = 4 * torch.cuda.get_device_properties(dev).multi_processor_count
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "achieved_is": "NOMINAL: SURVEY 8(d)'s algorithmic bytes per env-step x env-steps per launch / launch time, as the bench "
                               "contract defines it -- not bytes that crossed the bus (see moved_*; a multi-step launch reads and writes "
                               "the state once per launch, not once per step)",
                "algorithmic_GBps": achieved,
                "traffic": None, "traffic_source": None, "moved_bytes_per_step": None, "moved_GBps": None, "moved_frac": None,
                "bound_actual": "valu_issue", "valu_issue_frac": None,
                "kernel": kernel, "kernels_launched_in_timed_region": info["kernels"],
                "schedule": info.get("schedule"), "traffic_unavailable_because": why_not,
                "steps_per_launch": spl,
                "algorithmic_bytes_per_env_step": b_step,
                "algorithmic_bytes_per_launch": b_step * E * spl,
                "avg_launch_us": per_step_s * 1e6 * spl,
                "avg_launch_us_is": ("mean over the %d launch(es) of a REPEAT of the timed region's K steps, each dispatch carrying its own start / stop "
                                     "events (uavenv_launch_timing); the wall-clock region itself holds no events (they cost a 20-step region 9-10 us)"
                                     % len(info["launch_us"])) if info.get("launch_us") else "HIP events on the launch stream around the timed region",
                "launch_us": info.get("launch_us"),
                "avg_step_us": per_step_s * 1e6,
                "transcendental_evals_per_step": transcendental_evals_per_env_step(n_ue, n_bs) * E,
                "transcendental_evals_per_s": transcendental_evals_per_env_step(n_ue, n_bs) * E / per_step_s}
        if cnt is not None:
            # FETCH_SIZE on gfx950 counts 64 B per 128-B request for 16 B/lane streaming reads (MI355X_MICROARCH.md, HBM):
            # every state load of these kernels is such a dwordx4 record load, hence the x2; WRITE_SIZE is exact.  The committed
            # figures are per launch of cnt["steps_per_launch"] steps; scaled to this run's steps per launch.
            scale = 1.0                                      # (the entry IS this dispatch form: committed_counters matched steps per call and schedule)
            roof["traffic"] = int((2 * int(cnt["fetch_size_bytes_raw"]) + int(cnt["write_size_bytes_raw"])) * scale)
            roof["traffic_over_algorithmic"] = roof["traffic"] / float(b_step * E * spl)
            roof["traffic_source"] = ("profiles/traffic_current.json (%s): committed rocprofv3 PMC passes of this kernel at this batch size "
                                      "(FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), NOT measured by this run" % cnt.get("source", "?"))
            roof["moved_bytes_per_step"] = roof["traffic"] / float(spl)
            roof["moved_GBps"] = roof["moved_bytes_per_step"] / per_step_s / 1e9
            roof["moved_frac"] = roof["moved_GBps"] / HBM_PEAK_GBPS
            roof["valu_issue_frac"] = (cnt["valu_insts_per_launch"] * scale) / (n_simd * SIMD_ISSUE_HZ * per_step_s * spl)
            roof["valu_issue_note"] = ("SQ_INSTS_VALU per launch (committed profile) / (%d SIMDs x 0.6 G wave-instr/s) / measured "
                                       "launch time: the roof that actually bounds this float64-ALU kernel" % n_simd)
            roof["rocprof_avg_kernel_us"] = cnt.get("rocprof_avg_kernel_ns", 0) / 1e3
        if world > 1:                                   # per GPU above (rank 0's events, MAX over ranks); the node moves world x that
            roof["per"] = "GPU"
            roof["whole_node_algorithmic_GBps"] = achieved * world
            roof["whole_node_peak_GBps"] = HBM_PEAK_GBPS * world
        form = {"seq": "one kernel per step, launches issued by one C call (uavenv_step_seq) per <=100 steps",
                "graph": "one kernel per step, hipGraph replay of <=100-step chunks",
                "eager": "one kernel launch per step from Python",
                "many": "OPEN-LOOP action tape through uavenv_step_many: <=100 consecutive steps per launch, state carried in registers between "
                        "them, all nine outputs of every step written (a policy in the loop gets single_step_launch_value)"}[args.launch]
        line = {
            "metric": "env steps/sec (whole node) at 4-UAV x 20-UE", "value": whole_job_rate(E * K, world, elapsed),
            "unit": "env-steps/s", "n_gpus": n_ranks, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic", "launch": args.launch, "prewarm_steps_on_scratch_env": PREWARM_STEPS,
            "config": {"workload": "%d batched envs/GPU, %d UAV x %d UE (groups %s), G=%d, HIP step(), compact outputs, "
                                   "on-device Philox, %s" % (E, n_bs, n_ue, ",".join(str(g) for g in groups), grid, form),
                       "envs_per_gpu": E, "n_bs": n_bs, "n_ue": n_ue, "grid": grid, "parallelism": "env-shard x%d" % world},
            "per_rank_elapsed_s": info["per_rank_elapsed_s"], "per_rank_gpu_ms": info["per_rank_gpu_ms"],
            "roofline": roof,
        }
        if alt:
            line["other_launch_forms"] = alt
            if "seq" in alt:             # one kernel launch per step (what a closed-loop caller, e.g. the A2C rollout, gets)
                line["single_step_launch_value"] = alt["seq"]["value"]
        if grids:
            line["other_grids"] = grids
        if a2c is not None:
            line["a2c"] = a2c
        if world == 1 and not args.no_cpu_baseline and baseline_shape:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
