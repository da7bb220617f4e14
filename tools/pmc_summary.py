#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc CSVs under gpurun_out/pmc_* for the fused kernel (dev tool)."""
import collections, csv, glob, json, sys
kern = sys.argv[1] if len(sys.argv) > 1 else "k_fused"
frames = float(sys.argv[2]) if len(sys.argv) > 2 else 71488.0
res = {}
for f in sorted(glob.glob("gpurun_out/pmc_*/*/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        res[k] = sum(v) / len(v)
for k, v in sorted(res.items()):
    print(f"{k:26s}{v:.5g}")
g = res.get
if g("SQ_INSTS_VALU"):
    print("per frame: VALU %.0f LDS %.0f SALU %.0f VMEM_RD %.1f VMEM_WR %.1f" % tuple(
        g(k, 0) / frames for k in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")))
if g("GRBM_GUI_ACTIVE") and g("SQ_ACTIVE_INST_VALU"):
    cyc = g("GRBM_GUI_ACTIVE") / 8
    print("kernel cycles %.0f ; VALU busy %.2f ; wave-cycles: issuing %.2f waiting %.2f issue-stalled %.2f" % (
        cyc, g("SQ_ACTIVE_INST_VALU") * 4 / (cyc * 1024), g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"),
        g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")))
if g("FETCH_SIZE"):
    print("FETCH_SIZE %.1f MB (x2 for 16 B/lane streams on gfx950), WRITE_SIZE %.1f MB" % (
        g("FETCH_SIZE") / 1024, g("WRITE_SIZE", 0) / 1024))
