#!/usr/bin/env python3
"""Per-kernel ISA summary of a gfx950 assembly listing (hipcc -S --cuda-device-only): instructions, VGPRs, SGPRs, spills, LDS.

    hipcc -O3 --offload-arch=gfx950 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=16 -S --cuda-device-only -o k.s csrc/uavenv_capi.hip
    python tools/isa_stats.py k.s [name-substring]

Used to check that a source restructuring leaves the tuned instantiations' code unchanged (compare two listings)."""
import re
import subprocess
import sys


def main():
    s = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    labels = re.findall(r"^(_Z\w+):", s, re.M)
    names = subprocess.run(["c++filt"] + labels, capture_output=True, text=True).stdout.splitlines()
    for lab, name in zip(labels, names):
        if want not in name:
            continue
        body = s[s.index("\n" + lab + ":"):]
        body = body[:body.index(".Lfunc_end")]
        n = sum(1 for l in body.splitlines() if re.match(r"^\s+[sv]_|^\s+(global|flat|buffer|ds|scratch)_", l))
        meta = s[s.index(".amdhsa_kernel " + lab):]
        meta = meta[:meta.index(".end_amdhsa_kernel")]
        g = lambda k: (re.search(r"\." + k + r"\s+(\S+)", meta) or [None, "?"])[1]
        tail = s[s.index(".name:", s.index("amdhsa.kernels")):]
        m = re.search(r"\.name:\s+" + re.escape(lab) + r"\n(.*?)(?=\n  - |\Z)", s, re.S)
        info = {}
        if m:
            for k in ("sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count", "group_segment_fixed_size"):
                mm = re.search(r"\." + k + r":\s+(\d+)", m.group(0))
                info[k] = int(mm.group(1)) if mm else None
        print("%-70s instr %5d  vgpr %s sgpr %s spill s/v %s/%s lds %s" % (
            name[:70], n, info.get("vgpr_count"), info.get("sgpr_count"), info.get("sgpr_spill_count"),
            info.get("vgpr_spill_count"), info.get("group_segment_fixed_size")))


if __name__ == "__main__":
    main()
