"""Builds libuavenv.so (env kernels + C ABI) and libuavagent.so (the learner's sparse first layer) for gfx950 with hipcc, in-tree.

``python -m drl_uav_cellularnet_amd.build`` or ``__graft_entry__.build()``.  hipcc cross-compiles
without a GPU; the built library is git-ignored but travels to the GPU box with the tree.
"""
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
SRC = os.path.join(PKG_DIR, "csrc", "uavenv_capi.hip")
ENV_SRCS = [SRC, os.path.join(PKG_DIR, "csrc", "uavenv_gated.hip")]     # one object per translation unit: a change to one does not rebuild the other
ENV_HDRS = [os.path.join(PKG_DIR, "csrc", f) for f in ("uavenv_kernels.h", "uavenv_handle.h", "philox.h", "lean_math.h", "intdiv.h",
                                                        "state_layout.h")] + [os.path.join(ROOT, "include", "uavenv.h")]
ENV_EXTRA = {"uavenv_gated.hip": [os.path.join(PKG_DIR, "csrc", "uavenv_gated_kernel.h")]}      # headers of one translation unit only
DEPS = ENV_SRCS + ENV_HDRS + [h for hs in ENV_EXTRA.values() for h in hs]
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB = os.path.join(LIB_DIR, "libuavenv.so")
AGENT_SRCS = [os.path.join(PKG_DIR, "csrc", f) for f in ("agent_kernels.hip", "agent_learner.hip", "agent_gemm.hip")]
AGENT_SRC = AGENT_SRCS[0]
AGENT_DEPS = AGENT_SRCS + [os.path.join(PKG_DIR, "csrc", "agent_common.h"), os.path.join(ROOT, "include", "uavagent.h")]
AGENT_LIB = os.path.join(LIB_DIR, "libuavagent.so")
ARCH = "gfx950"


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.isfile(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def _stale(lib, deps):
    if not os.path.isfile(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build():
    return _stale(LIB, DEPS)


def build_agent(force=False, verbose=False):
    """libuavagent.so (include/uavagent.h): used by agent.py for CUDA tensors; the env library does not depend on it."""
    if not force and not _stale(AGENT_LIB, AGENT_DEPS):
        return AGENT_LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc_path(), "-O3", "-std=c++17", "--offload-arch=" + ARCH, "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
           "-o", AGENT_LIB] + AGENT_SRCS
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return AGENT_LIB


def build(force=False, verbose=False, extra_flags=()):
    build_agent(force=force, verbose=verbose)
    if not force and not needs_build():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    # -amdgpu-kernarg-preload-count: leading scalar kernel arguments arrive in SGPRs at wave launch (the packed env
    # kernel starts its global loads from them while the parameter struct is still being fetched)
    flags = ["-O3", "-std=c++17", "--offload-arch=" + ARCH, "-fPIC", "-Wall", "-Wno-unused-function", "-mllvm", "-amdgpu-kernarg-preload-count=16",
             *extra_flags]
    tag = os.path.join(obj_dir, "flags.txt")
    same_flags = os.path.isfile(tag) and open(tag).read() == " ".join(flags)
    objs = []
    for src in ENV_SRCS:
        obj = os.path.join(obj_dir, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or not same_flags or _stale(obj, [src] + ENV_HDRS + ENV_EXTRA.get(os.path.basename(src), [])):
            cmd = [hipcc_path(), *flags, "-c", "-o", obj, src]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    with open(tag, "w") as f:
        f.write(" ".join(flags))
    cmd = [hipcc_path(), "--offload-arch=" + ARCH, "-fPIC", "-shared", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
