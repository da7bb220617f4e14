#!/usr/bin/env python3
"""Rounding-noise check (dev tool): fused f32 and generic f32 against the generic kernels in float64
at the C2 frame size, from an identical phase_init start."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spectrogram_inversion_amd.plan import Plan, args_helper

dev = torch.device("cuda", 0)
n_fft, hop, frames, batch = 2048, 512, int(os.environ.get("FRAMES", 1024)), 2
hann = lambda n, dt: torch.from_numpy((0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n) / n)).astype(dt))
mag = torch.from_numpy(np.random.default_rng(1234).random((batch, n_fft // 2 + 1, frames), dtype=np.float32)).to(dev)
a32 = args_helper(torch.empty((1, 1025, 1)), hop_length=hop, window=hann(n_fft, np.float32))
a64 = args_helper(torch.empty((1, 1025, 1), dtype=torch.float64), hop_length=hop, window=hann(n_fft, np.float64))
fast = Plan(a32, batch, frames, torch.float32, dev)
gen = Plan(a32, batch, frames, torch.float32, dev); gen.force_generic(True)
ref = Plan(a64, batch, frames, torch.float64, dev)
init = fast.phase_init(mag)
def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm())
for its in (1, 3, 10, 30):
    out = {}
    for name, p, x0 in (("fast", fast, init), ("gen", gen, init), ("f64", ref, init.to(torch.complex128))):
        p.gla_init(x0, None, 0.3); p.iterate(its); out[name] = p.wave()
    print(f"iters={its:3d} fast-vs-f64 {rel(out['fast'], out['f64']):.3e}  generic-vs-f64 {rel(out['gen'], out['f64']):.3e}", flush=True)
