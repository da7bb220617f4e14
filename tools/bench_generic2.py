#!/usr/bin/env python3
"""ms per iteration of shapes on the generic LDS-FFT kernels (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spectrogram_inversion_amd.plan import Plan, args_helper
dev = torch.device("cuda", 0)
for n_fft, hop, frames, batch, dt in [(400, 160, 2048, 64, torch.float32), (256, 64, 4096, 64, torch.float32), (1000, 250, 1024, 64, torch.float32),
                                      (512, 128, 2048, 32, torch.float64), (2048, 512, 1024, 16, torch.float64), (128, 32, 4096, 64, torch.float32),
                                      (8192, 2048, 256, 32, torch.float32)]:
    mag = torch.rand((batch, n_fft // 2 + 1, frames), device=dev, dtype=dt)
    plan = Plan(args_helper(mag, hop_length=hop, window=torch.hann_window(n_fft, dtype=dt)), batch, frames, dt, dev)
    plan.gla_init(None, mag, 0.3)
    plan.iterate(3)
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); plan.iterate(20); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    print(f"n_fft {n_fft:5d} hop {hop:4d} T {frames} B {batch} {str(dt)[6:]} path={plan.path}: {best:.3f} ms/it  {batch * frames / best / 1e3:.1f} M frames/s")
