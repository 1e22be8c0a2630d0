"""GPU, ONE fresh child process with backend nccl (= RCCL) at world size 1: the exact call sequence an 8-GPU run of BASELINE configs[3]
executes first -- init_process_group("nccl"), the update's two-bucket exchange (dist.all_reduce(flat.g[lo:hi], async_op=True) issued on
the side stream, work.wait(), the main stream's wait_stream) and the one-bucket form -- has run on real hardware, and leaves the
parameters a run without any process group leaves (a2c_single_thread.py:107-133: one synchronous update over all workers' samples).
The 8-GPU node itself is the driver's to run."""
import os
import socket

import pytest

pytestmark = pytest.mark.gpu


def _worker(port, q):
    try:
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(0)
        try:
            dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
            probe = torch.ones(8, device="cuda:0")
            dist.all_reduce(probe)                                  # first collective: creates the RCCL communicator
            torch.cuda.synchronize()
        except Exception as ex:                                     # RCCL cannot initialise here: report, do not retry
            q.put(("rccl_unavailable", "%s: %s" % (type(ex).__name__, ex)))
            return
        from drl_uav_cellularnet_amd import BatchedMobiEnv
        from drl_uav_cellularnet_amd.agent import A2CRunner

        env0 = BatchedMobiEnv(512, nBS=4, nUE=20, grid_n=100, groups=[5, 5, 5, 5], device="cuda:0")
        out = {}
        for name, kw in (("two_buckets", dict(force_exchange=True, overlap_allreduce=True)),
                         ("one_bucket", dict(force_exchange=True, overlap_allreduce=False)),
                         ("no_exchange", dict(force_exchange=False))):
            r = A2CRunner(env0.clone(), rollout=6, **kw)
            stats = []
            for _ in range(3):
                st = r.train_rollout()
                stats.append((st["allreduce_buckets"], st["allreduce_overlapped_ms"], st["allreduce_ms"], st["grad_elems"]))
            torch.cuda.synchronize()
            out[name] = (r.flat.w.cpu(), r.flat.ms.cpu(), stats)
        q.put(("ok", {k: (v[0].numpy().tobytes(), v[1].numpy().tobytes(), v[2]) for k, v in out.items()}, dist.get_backend()))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as ex:                                         # anything else is a failure of the test, with its text
        import traceback

        q.put(("error", traceback.format_exc() + "\n%s" % ex))


def test_the_rccl_exchange_runs_at_one_rank_and_changes_nothing():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(port, q))
    p.start()
    res = q.get(timeout=600)
    p.join(timeout=120)
    if res[0] == "rccl_unavailable":
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        d = os.path.join(root, "gpurun_out")
        if os.path.isdir(d):
            with open(os.path.join(d, "rccl_one_rank_error.txt"), "w") as f:
                f.write(res[1] + "\n")
        pytest.skip("RCCL could not initialise on this box: " + res[1][:300])
    assert res[0] == "ok", res[1]
    assert p.exitcode == 0
    _, out, backend = res
    assert backend == "nccl"
    two, one, none = out["two_buckets"], out["one_bucket"], out["no_exchange"]
    for st in two[2]:
        assert st[0] is not None and len(st[0]) == 2 and st[1] is not None and st[3] == 20206626      # two buckets, the first one timed on the side stream
    for st in one[2]:
        assert st[0] is None and st[2] is not None
    assert two[0] == one[0] == none[0]                  # parameters: bit-identical bytes
    assert two[1] == one[1] == none[1]                  # RMSProp accumulators
