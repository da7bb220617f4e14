#!/usr/bin/env python3
"""Summarise tools/prof_generic.sh's output: per shape, the kernels' average durations and the counters per launch."""
import collections, csv, glob, json, os
out = {}
for d in sorted(glob.glob("gpurun_out/prof_gen/*")):
    tag = os.path.basename(d)
    rec = {"kernels": {}, "counters": {}}
    for f in glob.glob(d + "/kt/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Name"].split("(")[0][:90]
            rec["kernels"][name] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"])}
    for f in glob.glob(d + "/pmc*/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            for c, v in cs.items():
                rec["counters"].setdefault(k, {})[c] = sum(v) / len(v)
    out[tag] = rec
    print("==", tag)
    for k, v in sorted(rec["kernels"].items(), key=lambda kv: -kv[1]["pct"])[:4]:
        c = rec["counters"].get(k, {})
        extra = ""
        if c.get("SQ_LDS_IDX_ACTIVE"):
            extra += " lds_conflict %.2f" % (c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"])
        if c.get("GRBM_GUI_ACTIVE") and c.get("SQ_ACTIVE_INST_VALU"):
            cyc = c["GRBM_GUI_ACTIVE"] / 8
            extra += " valu_busy %.2f lds_busy %.2f" % (c["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * 1024), c.get("SQ_ACTIVE_INST_LDS", 0) * 4 / (cyc * 1024))
        if c.get("SQ_WAVE_CYCLES") and c.get("SQ_WAIT_INST_ANY"):
            extra += " wait_inst %.2f" % (c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"])
        print("  %-70s %6d x %9.1f us %5.1f %%%s" % (k[:70], v["calls"], v["avg_us"], v["pct"], extra))
json.dump(out, open("gpurun_out/prof_gen/summary.json", "w"), indent=1)
