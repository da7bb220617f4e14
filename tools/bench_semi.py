#!/usr/bin/env python3
"""ms per iteration of shapes that run on the frame kernel (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spectrogram_inversion_amd.plan import Plan, args_helper
dev = torch.device("cuda", 0)
for n_fft, hop, frames, batch, method in [(2048, 333, 1024, 32, "gla"), (1024, 160, 2048, 32, "gla"), (2048, 768, 1024, 64, "gla"),
                                          (1024, 160, 2048, 32, "admm"), (512, 100, 4096, 32, "gla")]:
    mag = torch.rand((batch, n_fft // 2 + 1, frames), device=dev)
    plan = Plan(args_helper(mag, hop_length=hop, window=torch.hann_window(n_fft)), batch, frames, torch.float32, dev)
    (plan.gla_init if method == "gla" else plan.admm_init)(None, mag, 0.3)
    plan.iterate(3)
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); plan.iterate(20); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    F = n_fft // 2 + 1
    per = 8 * hop + (20 if method == "gla" else 36) * F
    print(f"{method} n_fft {n_fft} hop {hop} T {frames} B {batch} path={plan.path}: {best:.3f} ms/it  "
          f"{per * batch * frames / best / 1e6:.0f} GB/s algorithmic ({per * batch * frames / best / 8e7:.1f}%)")
