#!/usr/bin/env python3
"""Performance off the BASELINE shapes: ms per (late) iteration and iterations*frames/s of griffin_lim (alpha 0.3) and ADMM (rho 0.1)
over batch x frames x (n_fft, hop), with the kernel each shape lands on.  Writes a JSON table (default profiles/<tag>_sweep.json)
and prints one row per shape; `rel` = the shape's per-frame rate over that of its BASELINE-sized neighbour (B 64, T 1024, same
n_fft / hop, same method).

    python tools/sweep.py --tag r05 [--batches 1,8,...] [--frames 300,1024] [--shapes 2048:512,...]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from spectrogram_inversion_amd.plan import Plan, args_helper, clear_plan_cache

ap = argparse.ArgumentParser()
ap.add_argument("--tag", default="r05")
ap.add_argument("--batches", default="1,8,32,48,64,65,100,256")
ap.add_argument("--frames", default="300,1024")
ap.add_argument("--shapes", default="2048:512,2048:1024,2048:256,1024:256,512:128,2048:333")
ap.add_argument("--methods", default="gla,admm")
ap.add_argument("--out", default=None)
args = ap.parse_args()
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
rows = []


def run(method, n_fft, hop, B, T):
    F = n_fft // 2 + 1
    w = torch.from_numpy((0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)).astype(np.float32))
    mag = torch.rand(B, F, T, device=dev)
    plan = Plan(args_helper(mag, hop_length=hop, window=w), B, T, torch.float32, dev)
    (plan.gla_init if method == "gla" else plan.admm_init)(None, mag, 0.3 if method == "gla" else 0.1)
    plan.iterate(20)                       # (past the launches that still add the starting spectrum's share)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plan.iterate(10)
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / 10
    n = max(10, min(400, int(0.15 / max(per, 1e-6))))
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        plan.iterate(n)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    g = plan.launch_geometry
    return {"method": method, "n_fft": n_fft, "hop": hop, "batch": B, "frames": T, "ms_per_iteration": best,
            "it_frames_per_s": B * T / best * 1e3, "kernel": g["kernel"], "chunks": g["chunks"], "waves": g["waves"],
            "waves_per_workgroup": g["waves_per_workgroup"], "path": plan.path}


for shp in args.shapes.split(","):
    n_fft, hop = map(int, shp.split(":"))
    for method in args.methods.split(","):
        for T in map(int, args.frames.split(",")):
            for B in map(int, args.batches.split(",")):
                try:
                    r = run(method, n_fft, hop, B, T)
                except Exception as e:          # (a shape the box cannot hold must not end the sweep)
                    r = {"method": method, "n_fft": n_fft, "hop": hop, "batch": B, "frames": T, "error": f"{type(e).__name__}: {e}"[:200]}
                rows.append(r)
                clear_plan_cache()
                torch.cuda.empty_cache()
base = {(r["method"], r["n_fft"], r["hop"]): r["it_frames_per_s"] for r in rows if r.get("batch") == 64 and r.get("frames") == 1024 and "error" not in r}
for r in rows:
    b = base.get((r["method"], r["n_fft"], r["hop"]))
    if b and "error" not in r:
        r["rel"] = r["it_frames_per_s"] / b
    if "error" in r:
        print(r)
    else:
        print(f"{r['method']:5s} {r['n_fft']:5d}/{r['hop']:<5d} B{r['batch']:<4d} T{r['frames']:<5d} {r['ms_per_iteration']:8.4f} ms "
              f"{r['it_frames_per_s'] / 1e6:8.1f} M  rel {r.get('rel', float('nan')):5.2f}  {r['kernel']:12s} chunks {r['chunks']:4d} waves {r['waves']:6d} "
              f"wg {r['waves_per_workgroup']}", flush=True)
out = args.out or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"{args.tag}_sweep.json")
with open(out, "w") as fh:
    json.dump({"what": __doc__.strip().splitlines()[0], "rows": rows}, fh, indent=0)
print("wrote", out)
