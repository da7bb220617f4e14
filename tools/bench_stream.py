#!/usr/bin/env python3
"""Latency of one RTISIStream.push (dev tool): ms per pushed frame at the C3 shape for a few batch sizes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import spectrogram_inversion_amd as si

dev = torch.device("cuda", 0)
for batch, k in [(1, 1), (32, 1), (32, 8), (256, 1)]:
    s = si.RTISIStream(1025, batch=batch, look_ahead=3, asymmetric_window=True, max_iter=25, alpha=0.99, max_push=8,
                       hop_length=512, window=torch.hann_window(2048), device=dev)
    mag = torch.rand(batch, 1025, k, device=dev)
    for _ in range(5):
        s.push(mag)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 40
    for _ in range(n):
        s.push(mag)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"batch {batch:4d}  {k} frame(s)/push: {dt * 1e3:.2f} ms per push, {dt * 1e3 / k:.2f} ms per frame "
          f"({25 * k / dt / 1e3:.1f} k dependent steps/s per stream; real time at 22.05 kHz needs <= 23.2 ms per frame)",
          flush=True)
