#!/usr/bin/env python3
"""Host round trip of one scalar read-back (dev tool): two tiny kernels + 8-byte D2H copy + stream synchronise."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spectrogram_inversion_amd.lbfgs import HipVecOps
dev = torch.device("cuda", 0)
ops = HipVecOps(torch.float32, dev)
a = torch.ones(1024, device=dev)
for _ in range(20):
    ops.dot(a, a)
t0 = time.perf_counter()
n = 2000
for _ in range(n):
    ops.dot(a, a)
dt = (time.perf_counter() - t0) / n
print(f"scalar read-back round trip: {dt * 1e6:.1f} us per call")
b = torch.ones(8 << 20, device=dev)
for _ in range(5):
    ops.dot(b, b)
t0 = time.perf_counter()
for _ in range(200):
    ops.dot(b, b)
print(f"dot of 8M floats incl. read-back: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per call")
