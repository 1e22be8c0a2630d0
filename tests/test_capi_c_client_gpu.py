"""GPU: the C ABI used from plain C (tests/native/capi_client.c: no Python, no torch, hipMalloc'd buffers) gives exactly what
BatchedMobiEnv gives for the same seed and actions -- the drop-in boundary is the C library, not its Python wrapper."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_client_matches_the_python_binding(tmp_path):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from drl_uav_cellularnet_amd import BatchedMobiEnv, build

    lib_dir = os.path.dirname(build.build())
    exe = os.path.join(tmp_path, "capi_client")
    rocm = os.path.dirname(os.path.dirname(os.path.realpath(build.hipcc_path())))          # .../rocm/bin/hipcc -> .../rocm
    subprocess.check_call(["gcc", "-O2", "-D__HIP_PLATFORM_AMD__", os.path.join(ROOT, "tests", "native", "capi_client.c"),
                           "-I", os.path.join(rocm, "include"), "-I", os.path.join(ROOT, "include"), "-L", lib_dir, "-luavenv",
                           "-L", os.path.join(rocm, "lib"), "-lamdhip64", "-Wl,-rpath," + lib_dir,
                           "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", exe])
    N, T = 300, 5
    lines = subprocess.check_output([exe, str(N)], text=True, timeout=300).splitlines()
    assert lines[-1].startswith("error step: null handle or actions")
    rows = [l.split() for l in lines if l.startswith("step ")]
    assert len(rows) == 3 + T
    env = BatchedMobiEnv(N, nBS=4, nUE=20, grid_n=100, seed=0x5EED)             # ctor = init + 200 warm-up ticks + reset
    acts = ((torch.arange(N * T, dtype=torch.int64) * 7919 + 13) % 625).reshape(T, N).to(env.device)
    got = []
    for t in range(3):
        env.step(acts[t])
        got.append((int(env.out["step_n"][0]), float(env.out["reward"].double().sum()), int(env.out["serving"].long().sum()),
                    int(env.out["ue_xy"][0, 0, 0]), int(env.out["ue_xy"][0, 0, 1])))
    many = env.step_many(acts)
    for t in range(T):
        got.append((int(many["step_n"][t, 0]), float(many["reward"][t].double().sum()), int(many["serving"][t].long().sum()),
                    int(many["ue_xy"][t, 0, 0, 0]), int(many["ue_xy"][t, 0, 0, 1])))
    for r, g in zip(rows, got):
        assert int(r[1]) == g[0] and int(r[3]) == g[2] and int(r[4]) == g[3] and int(r[5]) == g[4], (r, g)
        np.testing.assert_allclose(float(r[2]), g[1], rtol=1e-6)              # (float32 rewards summed in double on both sides)
    assert [g[0] for g in got] == [1, 2, 3, 4, 5, 6, 7, 8]


def test_integration_md_ctypes_stub_runs():
    """The stand-alone ctypes binding printed in INTEGRATION.md section 2(b) is executed verbatim: documentation that runs."""
    import re

    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = [b for b in blocks if "lib.uavenv_create" in b]
    assert len(stub) == 1
    cwd = os.getcwd()
    os.chdir(ROOT)                                   # the stub loads the library by its in-tree relative path
    try:
        ns = {}
        exec(compile(stub[0], "INTEGRATION.md:2b", "exec"), ns)
    finally:
        os.chdir(cwd)
    torch.cuda.synchronize()
    assert int(ns["bufs"]["step_n"].min()) == 1 and bool(torch.isfinite(ns["bufs"]["reward"]).all())
