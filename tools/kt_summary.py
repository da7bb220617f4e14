#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace CSV (second half of the trace: steady state): count, total and mean duration,
and the span / busy time of the window.  usage: kt_summary.py <dir> [top]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 14
f = sorted(glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / 1e6
print(f"kernels {len(rows)}  span {span:.2f} ms  busy {busy:.2f} ms")
acc = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r["Kernel_Name"].split("(")[0][-60:]
    acc[n][0] += 1
    acc[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"  {n:<60} {c:6d} {t:10.1f} us total {t / c:8.1f} us each")
