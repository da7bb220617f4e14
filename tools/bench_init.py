#!/usr/bin/env python3
"""ms of the one-off part of a C2 step (dev tool): phase_init alone and gla_init (phase_init + layout conversion +
initial ISTFT)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from spectrogram_inversion_amd.plan import Plan, args_helper

dev = torch.device("cuda", 0)
mag = torch.rand((64, 1025, 1024), device=dev)
plan = Plan(args_helper(mag, hop_length=512, window=torch.hann_window(2048)), 64, 1024, torch.float32, dev)


def timed(fn, n=10):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print(f"phase_init {timed(lambda: plan.phase_init(mag)):.3f} ms   gla_init {timed(lambda: plan.gla_init(None, mag, 0.3)):.3f} ms   "
      f"wave {timed(lambda: plan.wave()):.3f} ms")
