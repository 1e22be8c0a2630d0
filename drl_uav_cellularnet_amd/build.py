"""Builds libuavenv.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

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
DEPS = [SRC] + [os.path.join(PKG_DIR, "csrc", f) for f in ("uavenv_kernels.h", "philox.h", "lean_math.h", "intdiv.h",
                                                            "state_layout.h")] + [
    os.path.join(ROOT, "include", "uavenv.h")]
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB = os.path.join(LIB_DIR, "libuavenv.so")
ARCH = "gfx950"


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.isfile(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def needs_build():
    if not os.path.isfile(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False, extra_flags=()):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    # -amdgpu-kernarg-preload-count: leading scalar kernel arguments arrive in SGPRs at wave launch (the packed env
    # kernel starts its global loads from them while the parameter struct is still being fetched)
    cmd = [hipcc_path(), "-O3", "-std=c++17", "--offload-arch=" + ARCH, "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function", "-mllvm", "-amdgpu-kernarg-preload-count=16", *extra_flags, "-o", LIB, SRC]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
