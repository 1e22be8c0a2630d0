#!/usr/bin/env python3
"""Charge every instruction of one compiled kernel to the source line it came from.

    python tools/attribute_lines.py                      # the shipped step kernel: env_kernel_packed<4, 2, true, true, true>
    python tools/attribute_lines.py 'env_kernel_packed<4, 2, true, true, false>' --top 40

Runs here (no GPU): compiles csrc/uavenv_capi.hip for gfx950 with -g -S, which keeps `.loc file line` markers in the
assembly without changing the generated code (the -g build of the step kernel has the same instruction count as the shipped
one), then totals instructions per (file, line) and per class (f64 / integer VALU / v_mov+v_cndmask / SALU / memory).
tools/issue_cost.hip showed that on gfx950 every opcode these kernels use costs 4-6 issue cycles for a lone wave (except
v_rcp/rsq_f64 and ds_bpermute), so instructions per line is a usable cost profile.  Line 0 = compiler-generated code without
a source position.  This is how the v16-v18 cuts in DESIGN.md section 4 were found.
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "drl_uav_cellularnet_amd", "csrc")


def classify(op):
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "flat_", "buffer_", "ds_", "scratch_")):
        return "mem"
    if op.startswith(("v_mov", "v_cndmask", "v_accvgpr")):
        return "movsel"
    if "f64" in op:
        return "f64"
    return "int"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kernel", nargs="?", default="env_kernel_packed<4, 2, true, true, true>",
                    help="substring of the demangled kernel name")
    ap.add_argument("--top", type=int, default=30)
    ap.add_argument("--asm", help="reuse an existing -g assembly file instead of compiling")
    a = ap.parse_args()

    asm = a.asm
    if asm is None:
        asm = os.path.join(tempfile.mkdtemp(prefix="uavenv_attr_"), "k.s")
        cmd = ["hipcc", "-O3", "-g", "--offload-arch=gfx950", "-std=c++17", "-mllvm", "-amdgpu-kernarg-preload-count=16",
               "-S", "--cuda-device-only", "-o", asm, os.path.join(CSRC, "uavenv_capi.hip")]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    s = open(asm).read()

    files = {}
    for m in re.finditer(r'^\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s, re.M):
        files[int(m.group(1))] = os.path.basename(m.group(3) or m.group(2))
    labels = re.findall(r"^(_Z\w+):", s, re.M)
    names = subprocess.run(["c++filt"] + labels, capture_output=True, text=True).stdout.splitlines()
    hits = [(l, n) for l, n in zip(labels, names) if a.kernel in n]
    if len(hits) != 1:
        sys.exit("kernel name matches %d symbols: %s" % (len(hits), [n.split("(")[0] for _, n in hits][:8]))
    label, name = hits[0]
    start = re.search(r"^" + re.escape(label) + ":", s, re.M).start()
    body = s[start:s.index("s_endpgm", start)]

    total, kinds, cur = Counter(), {}, ("?", 0)
    for line in body.splitlines():
        t = line.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        if not line.startswith("\t") or not t or t.startswith((".", ";")):
            continue
        total[cur] += 1
        kinds.setdefault(cur, Counter())[classify(t.split()[0])] += 1

    cache = {}

    def source(f, ln):
        p = os.path.join(CSRC, f)
        if ln <= 0 or not os.path.exists(p):
            return ""
        if p not in cache:
            cache[p] = open(p).read().splitlines()
        return cache[p][ln - 1].strip()[:90] if ln <= len(cache[p]) else ""

    n = sum(total.values())
    cls = Counter()
    for k in kinds.values():
        cls.update(k)
    print("%s\n%d instructions: %s" % (name.split("(")[0], n, ", ".join("%s %d" % kv for kv in cls.most_common())))
    per_file = Counter()
    for (f, _), v in total.items():
        per_file[f] += v
    print("per file: " + ", ".join("%s %d" % kv for kv in per_file.most_common(8)))
    print("\n%5s  %-46s %s" % ("instr", "classes", "source line"))
    for key, v in total.most_common(a.top):
        print("%5d  %-46s %s:%d  %s" % (v, ", ".join("%s %d" % kv for kv in kinds[key].most_common()), key[0], key[1], source(*key)))


if __name__ == "__main__":
    main()
