#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of tools/profile_bench.sh (gpurun_out/r01_*) into the committed summaries under
profiles/: the kernel-trace statistics, the PMC means of the dominant kernel and profiles/traffic.json
(HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB: FETCH_SIZE counts 32-B units as 64 on gfx950 for 16-B/lane
streams, see MI355X_MICROARCH.md, HBM section)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
kernel = "specinv::fast::k_fused4<16, 0, false>"
out = os.path.join(ROOT, "profiles")
stats = sorted(glob.glob(os.path.join(ROOT, f"gpurun_out/{tag}_kt/*/*kernel_stats.csv")), key=os.path.getmtime)
if stats:
    shutil.copy(stats[-1],   # (gpurun merges new files next to older runs' files: take the newest)
                os.path.join(out, f"{tag}_bench_kernel_stats.csv"))
# steady-state duration of the dominant kernel: launches of the timed steps only (the warm-up step pays first-touch)
trace = sorted(glob.glob(os.path.join(ROOT, f"gpurun_out/{tag}_kt/*/*kernel_trace.csv")), key=os.path.getmtime)[-1:]
steady = None
if trace:
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(trace[0]))
         if kernel in r["Kernel_Name"]]
    if len(d) >= 180:
        tail = d[90:]
        steady = {"launches": len(tail), "mean_us": sum(tail) / len(tail), "min_us": min(tail), "max_us": max(tail),
                  "all_launches_mean_us": sum(d) / len(d)}
counters = {}
newest = {}
for f in glob.glob(os.path.join(ROOT, f"gpurun_out/{tag}_pmc_*/*/*counter_collection.csv")):
    d = os.path.dirname(f)
    if d not in newest or os.path.getmtime(f) > os.path.getmtime(newest[d]):
        newest[d] = f
for f in sorted(newest.values()):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kernel in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        counters[k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
summary = {
    "command": "rocprofv3 --pmc <counters> --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "
               "(one run per counter group, tools/profile_bench.sh)",
    "kernel": kernel, "workload": "C2 (batch 64, n_fft 2048, hop 512, 1024 frames)",
    "kernel_trace_steady_state": steady, "counters": counters}
json.dump(summary, open(os.path.join(out, f"{tag}_bench_pmc_k_fused.json"), "w"), indent=1)
if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
    fs, ws = counters["FETCH_SIZE"]["mean_per_launch"], counters["WRITE_SIZE"]["mean_per_launch"]
    json.dump({"C2": (2 * fs + ws) * 1024,
               "_note": f"HBM bytes per launch of k_fused4<16,0,false>: (2*FETCH_SIZE + WRITE_SIZE)*1024 from "
                        f"profiles/{tag}_bench_pmc_k_fused.json; gfx950 FETCH_SIZE correction per MI355X_MICROARCH.md "
                        f"(HBM section)",
               "_fetch_size_kib": fs, "_write_size_kib": ws}, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(json.dumps({"steady": steady, "traffic": (2 * counters["FETCH_SIZE"]["mean_per_launch"] + counters["WRITE_SIZE"]["mean_per_launch"]) * 1024
                  if "FETCH_SIZE" in counters else None}, indent=1))
