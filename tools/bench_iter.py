#!/usr/bin/env python3
"""Micro-benchmark of the per-iteration launch (dev tool): ms per launch and algorithmic GB/s
for a workload under the current environment (SPECINV_FAST_CHUNK, ...)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from spectrogram_inversion_amd.plan import Plan, args_helper

ap = argparse.ArgumentParser()
ap.add_argument("--n-fft", type=int, default=2048)
ap.add_argument("--frames", type=int, default=1024)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--method", default="gla")
ap.add_argument("--launches", type=int, default=50)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--chunks", default="")
ap.add_argument("--hop", type=int, default=0)
ap.add_argument("--generic", action="store_true", help="force the generic kernels")
ap.add_argument("--keep-state", action="store_true", help="iterate on the reference's spectral state itself (specinv_plan_keep_state)")
args = ap.parse_args()

dev = torch.device("cuda", 0)
n_fft, hop = args.n_fft, args.hop or args.n_fft // 4
F = n_fft // 2 + 1
w = torch.from_numpy((0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)).astype(np.float32))
mag = torch.from_numpy(np.random.default_rng(0).random((args.batch, F, args.frames), dtype=np.float32)).to(dev)
a = args_helper(mag, hop_length=hop, window=w)
per_unit = 8 * hop + 20 * F
for chunk in ([None] + [int(c) for c in args.chunks.split(",") if c]):
    if chunk:
        os.environ["SPECINV_FAST_CHUNK"] = str(chunk)
    plan = Plan(a, args.batch, args.frames, torch.float32, dev)
    if args.generic:
        plan.force_generic(True)
    plan.keep_state(args.keep_state)
    if args.method == "gla":
        plan.gla_init(None, mag, 0.3)
    else:
        plan.admm_init(None, mag, 0.1)
    plan.iterate(20)           # (past the launches that still add the starting spectrum's share)
    best = 1e9
    for _ in range(args.rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        plan.iterate(args.launches)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / args.launches)
    gbs = per_unit * args.batch * args.frames / (best * 1e-3) / 1e9
    print(f"chunk={chunk} path={plan.path} kernel={plan.launch_geometry['kernel']} {best:.4f} ms/launch  {gbs:.0f} GB/s algorithmic "
          f"({100 * gbs / 8000:.1f}% of 8 TB/s)", flush=True)
    del plan
