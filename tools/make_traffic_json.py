#!/usr/bin/env python3
"""profiles/traffic_current.json from the rocprofv3 output directories of ONE gpurun call of tools/r04_pmc.sh.

    python tools/make_traffic_json.py gpurun_out/r04e profiles/r04e   (second argument: prefix the cited evidence files get in profiles/)

Per workload (tag of r03_pmc.sh) the PMC passes are reduced to figures PER CALL of the step entry point: a uavenv_step_many call is one
kernel dispatch, or D dispatches of the same kernel under the rotation schedule (csrc/uavenv_capi.hip: rotation_plan), so counters and
kernel time are SUMMED over the dispatches of the kernel and divided by the number of calls the bench made (steps / steps per call).
Kernels are keyed by the name the library's launch census gives them (bench.py reads the census, not the profiler)."""
import csv
import glob
import json
import os
import re
import sys

MODES = {0: "WARMUP", 1: "RESET", 2: "STEP", 3: "TRACE", 4: "RESET_TRACE"}


def census_name(rocprof_name):
    m = re.search(r"env_kernel_packed<(\d+), (\d+), (true|false), (true|false), (true|false), (true|false), (true|false)>", rocprof_name)
    if m:
        bt, mode = int(m.group(1)), int(m.group(2))
        plc, fast, pin, many, sched = [g == "true" for g in m.groups()[2:]]
        return "env_kernel_packed<BT=%d, %s, PLC=%d, FAST=%d, PIN=%d, MANY=%d%s>" % (bt, MODES[mode], plc, fast, pin, many, ", SCHED=1" if sched else "")
    m = re.search(r"env_kernel_multipass<(\d+), (\d+), (true|false), (true|false)>", rocprof_name)
    if m:
        return "env_kernel_multipass<BT=%d, %s, PLC=%d, FAST=%d>" % (int(m.group(1)), MODES[int(m.group(2))], m.group(3) == "true", m.group(4) == "true")
    return None


def counters(d):
    """{census kernel name: {counter: (dispatches, sum)}} of one pass directory."""
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = census_name(r["Kernel_Name"])
            if k:
                n, s = out.setdefault(k, {}).get(r["Counter_Name"], (0, 0.0))
                out[k][r["Counter_Name"]] = (n + 1, s + float(r["Counter_Value"]))
    return out


def stats(d):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = census_name(r["Name"])
            if k:
                out[k] = (int(r["Calls"]), float(r["TotalDurationNs"]))
    return out


def main():
    root, cite = sys.argv[1], sys.argv[2]
    # the manifest tools/r04_pmc.sh wrote in the same gpurun call: per tag the workload, the dispatch form and the step counts it ran
    # (the bench runs warm-up + timed steps through the entry point)
    runs = json.load(open(os.path.join(root, "runs.json")))["runs"]
    entries = []
    for rn in runs:
        tag, envs, n_bs, n_ue, many, spc = rn["tag"], rn["envs"], rn["n_bs"], rn["n_ue"], bool(rn["many"]), rn["steps_per_call"]
        rep = 1 + rn.get("timed_steps_repeated_for_launch_timing", 0)          # bench.py runs the timed steps of a multi-step form twice
        trace_steps, pmc_steps = rn["warmup"] + rep * rn["trace_steps"], rn["warmup"] + rep * rn["pmc_steps"]
        c = {}
        for p in ("FETCH_SIZE", "WRITE_SIZE", "sq"):
            for k, v in counters(os.path.join(root, "pmc_%s_%s" % (tag, p))).items():
                c.setdefault(k, {}).update(v)
        st = stats(os.path.join(root, "trace_" + tag))
        for k in sorted(c):
            if "STEP" not in k or ("MANY=1" in k) != many and "packed" in k:
                continue
            if many and "MANY=0" in k:
                continue                              # (the scratch-env pre-warm's single steps: described by the seq passes)
            cc = c[k]
            if many:
                calls_pmc, calls_tr = pmc_steps // spc + rn.get("prewarm_calls", 0), trace_steps // spc + rn.get("prewarm_calls", 0)
            else:
                calls_pmc, calls_tr = cc["FETCH_SIZE"][0], st[k][0]        # one dispatch per call
            dpc = cc["FETCH_SIZE"][0] / float(calls_pmc)
            e = {"envs": envs, "n_bs": n_bs, "n_ue": n_ue, "kernel": k, "steps_per_launch": spc, "schedule": rn["schedule"],
                 "dispatches_per_call": round(dpc, 3),
                 "fetch_size_bytes_raw": int(round(cc["FETCH_SIZE"][1] * 1024 / calls_pmc)),
                 "write_size_bytes_raw": int(round(cc["WRITE_SIZE"][1] * 1024 / calls_pmc)),
                 "valu_insts_per_launch": round(cc["SQ_INSTS_VALU"][1] / (cc["SQ_INSTS_VALU"][0] / dpc), 1),
                 "waves_per_launch": round(cc["SQ_WAVES"][1] / (cc["SQ_WAVES"][0] / dpc), 1),
                 "sq_wait_any_over_wave_cycles": round(cc["SQ_WAIT_ANY"][1] / cc["SQ_WAVE_CYCLES"][1], 4),
                 "rocprof_avg_kernel_ns": round(st[k][1] / calls_tr, 1),
                 "source": "%s_pmc_and_trace_digest.txt (PMC: %d dispatches), %s_%s_kernel_stats.csv (%d dispatches)" % (
                     cite, cc["FETCH_SIZE"][0], cite, tag, st[k][0])}
            expect = {"plain": 1.0, "one_launch_rotation": 1.0}.get(rn["schedule"])
            if expect is not None and abs(dpc - expect) > 0.05:
                e["WARNING"] = "expected %.0f dispatch(es) per call for schedule %s" % (expect, rn["schedule"])
            entries.append(e)
    doc = {"note": "Per-CALL rocprofv3 figures of the step kernels, one entry per DISPATCH FORM (kernel, batch, steps per call, schedule: plain launch / "
                   "one-launch rotation): bench.py uses an entry only for a run of exactly that form and prints traffic: null otherwise.  Three "
                   "separate PMC passes each (FETCH_SIZE / WRITE_SIZE / SQ counters) plus a kernel trace, all from one gpurun call (tools/r04_pmc.sh; its "
                   "runs.json manifest carries the step counts).  fetch/write are RAW counter values (KB x 1024).  Every state load of these kernels "
                   "is a 16 B/lane dwordx4 record load, the case where gfx950 reports exactly half (MI355X_MICROARCH.md, HBM): bench.py doubles "
                   "FETCH_SIZE.  Generated by tools/make_traffic_json.py; kernel = the library's launch-census name.",
           "entries": entries}
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
