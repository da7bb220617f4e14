#!/usr/bin/env python3
"""ms per iteration of configurations off the fused path (dev tool): shows what the generic 2-kernel path costs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from spectrogram_inversion_amd.plan import Plan, args_helper

dev = torch.device("cuda", 0)
CASES = [  # n_fft, hop, frames, batch, dtype, force_generic
    (2048, 512, 1024, 64, torch.float32, False),
    (2048, 512, 1024, 64, torch.float32, True),
    (2048, 256, 1024, 32, torch.float32, False),
    (2048, 1024, 1024, 64, torch.float32, False),
    (1024, 256, 2048, 32, torch.float32, True),
    (1024, 128, 2048, 16, torch.float32, False),
    (512, 128, 2048, 64, torch.float32, False),
    (4096, 1024, 512, 64, torch.float32, False),
    (400, 160, 2048, 64, torch.float32, False),
    (2048, 512, 1024, 16, torch.float64, False),
]
for n_fft, hop, frames, batch, dtype, force in CASES:
    F = n_fft // 2 + 1
    w = torch.hann_window(n_fft, dtype=dtype)
    mag = torch.rand((batch, F, frames), dtype=dtype, device=dev)
    a = args_helper(mag, hop_length=hop, window=w)
    plan = Plan(a, batch, frames, dtype, dev)
    if force:
        plan.force_generic(True)
    plan.gla_init(None, mag, 0.3)
    plan.iterate(3)
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        plan.iterate(20)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    es = 4 if dtype == torch.float32 else 8
    per_unit = (2 * hop + 5 * F) * es
    gbs = per_unit * batch * frames / (best * 1e-3) / 1e9
    print(f"n_fft {n_fft:5d} hop {hop:5d} T {frames:5d} B {batch:3d} {str(dtype)[6:]:8s} fast={plan.fast_path and not force!s:5s} "
          f"{best:8.3f} ms/it  {batch * frames / best / 1e3:8.1f} M frames/s  {gbs:7.0f} GB/s algorithmic ({100 * gbs / 8000:.1f}%)",
          flush=True)
    del plan
