#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of tools/profile_workloads.sh (gpurun_out/<tag>_<W>_*) into the committed summaries under
profiles/: per workload the kernel-trace statistics (<tag>_<W>_kernel_stats.csv), the PMC means of its dominant kernels
(<tag>_<W>_pmc.json) and profiles/traffic.json (HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB: on gfx950
FETCH_SIZE counts a 16-B/lane streaming read at half its bytes, MI355X_MICROARCH.md, HBM section)."""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
sys.path.insert(0, ROOT)
import bench                                           # noqa: E402  (the dominant-kernel table is bench.py's)

# the dominant kernel of each workload first (bench.py:DOMINANT), then the other kernels worth a row in the summary
DOMINANT = {
    "C2": [bench.DOMINANT["C2"][0], "specinv::fast::k_fused4_td<16, true, false>", "specinv::fast::k_eval_td<16, 4>",
           "specinv::fast::k_phase_init_pairs<16>"],
    "C4": [bench.DOMINANT["C4"][0], "specinv::fast::k_fused4<8, 1, true>"],
    "C3": [bench.DOMINANT["C3"][0]],
    "C5": [bench.DOMINANT["C5"][0], "specinv::k_objective_epilogue", "specinv::k_lbd_direction_lean<float>", "specinv::k_lbd_settle_x<float>"],
    "F64": [bench.DOMINANT["F64"][0], "specinv::wave::k_wave_seams<double>"],
    "S32": [bench.DOMINANT["S32"][0], "specinv::wave::k_wave_seams<float>"],
    "W400": [bench.DOMINANT["W400"][0], "specinv::wave::k_wave_seams<float>"],
}
ALGO = {"C2": 64 * 1024 * 8196, "C4": 32 * 2048 * 12308, "C3": None, "C5": 16 * 1024 * 8512,
        "F64": 16 * 1024 * 2 * (8 * 512 + 20 * 1025), "S32": 64 * 4096 * (8 * 64 + 20 * 129),
        "W400": 64 * 2048 * (8 * 160 + 20 * 201)}   # (C5: 16 hop + 4 mels - the walk applies the step too)
out = os.path.join(ROOT, "profiles")
traffic_path = os.path.join(out, "traffic.json")
try:
    traffic = json.load(open(traffic_path))
except (OSError, ValueError):
    traffic = {}


def newest(pattern):
    f = sorted(glob.glob(os.path.join(ROOT, pattern)), key=os.path.getmtime)
    return f[-1] if f else None


for W, kernels in DOMINANT.items():
    stats = newest(f"gpurun_out/{tag}_{W}_kt/*/*kernel_stats.csv")
    if not stats:
        continue
    shutil.copy(stats, os.path.join(out, f"{tag}_{W}_kernel_stats.csv"))
    trace = newest(f"gpurun_out/{tag}_{W}_kt/*/*kernel_trace.csv")
    summary = {"command": f"rocprofv3 --pmc <counters> --output-format csv -- python3 bench.py --workload {W} --steps 2 --warmup 1 "
                          f"--no-cpu-baseline --no-check (one run per counter group, tools/profile_workloads.sh)", "kernels": {}}
    rows = list(csv.DictReader(open(trace))) if trace else []
    for kern in kernels:
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if kern in r["Kernel_Name"]]
        entry = {}
        if d:
            tail = d[len(d) // 3:]                       # launches of the two timed steps (the warm-up step pays first touch)
            entry["kernel_trace"] = {"launches": len(d), "mean_us": sum(d) / len(d), "steady_mean_us": sum(tail) / len(tail),
                                     "min_us": min(d), "max_us": max(d)}
        counters = {}
        for grp in ("fetch", "write", "sq", "inst", "mfma"):
            f = newest(f"gpurun_out/{tag}_{W}_pmc_{grp}/*/*counter_collection.csv")
            if not f:
                continue
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if kern in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, v in agg.items():
                counters[k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
        entry["counters"] = counters
        if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
            entry["hbm_bytes_per_launch"] = (2 * counters["FETCH_SIZE"]["mean_per_launch"] + counters["WRITE_SIZE"]["mean_per_launch"]) * 1024
        c = counters
        if "SQ_WAVE_CYCLES" in c:
            wc = c["SQ_WAVE_CYCLES"]["mean_per_launch"]
            entry["wave_cycle_shares"] = {k: c[k]["mean_per_launch"] / wc for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
                                                                                   "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS") if k in c}
        if "GRBM_GUI_ACTIVE" in c and "SQ_ACTIVE_INST_VALU" in c:
            # share of the chip's VALU issue slots in use: SQ_ACTIVE_INST_VALU counts 4-cycle quanta over all SIMDs (1024),
            # GRBM_GUI_ACTIVE the busy cycles summed over the 8 XCDs
            entry["valu_issue_frac"] = c["SQ_ACTIVE_INST_VALU"]["mean_per_launch"] * 4 / (c["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8 * 1024)
        if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE", {}).get("mean_per_launch"):
            entry["lds_conflict_frac"] = c["SQ_LDS_BANK_CONFLICT"]["mean_per_launch"] / c["SQ_LDS_IDX_ACTIVE"]["mean_per_launch"]
        summary["kernels"][kern] = entry
    json.dump(summary, open(os.path.join(out, f"{tag}_{W}_pmc.json"), "w"), indent=1)
    first = summary["kernels"].get(kernels[0], {})
    if "hbm_bytes_per_launch" in first:
        traffic[W] = first["hbm_bytes_per_launch"]
        traffic[f"_{W}_note"] = (f"{kernels[0]}: (2*FETCH_SIZE + WRITE_SIZE)*1024 from profiles/{tag}_{W}_pmc.json"
                                 + (f"; algorithmic {ALGO[W]} B" if ALGO[W] else ""))
    for key in ("valu_issue_frac", "lds_conflict_frac"):
        if key in first:
            traffic[f"{W}_{key}"] = first[key]
    if "wave_cycle_shares" in first and "SQ_ACTIVE_INST_VALU" in first["wave_cycle_shares"]:
        traffic[f"{W}_valu_share_of_wave_life"] = first["wave_cycle_shares"]["SQ_ACTIVE_INST_VALU"]
    kt = first.get("kernel_trace", {})
    print(W, kernels[0], "steady %.1f us" % kt.get("steady_mean_us", float("nan")),
          "traffic %.4g B" % first.get("hbm_bytes_per_launch", float("nan")),
          ("= %.3f x algorithmic" % (first["hbm_bytes_per_launch"] / ALGO[W])) if ALGO[W] and "hbm_bytes_per_launch" in first else "")
sha = newest(f"gpurun_out/{tag}_csrc_sha1.txt")
if sha:
    traffic["csrc_sha1"] = open(sha).read().strip()
    traffic["_csrc_sha1_note"] = (f"sha1 of the kernel sources the {tag} profiles were taken on (python -m spectrogram_inversion_amd.build "
                                  "--hash); bench.py prints traffic_stale when the tree it runs differs")
json.dump(traffic, open(traffic_path, "w"), indent=1)
