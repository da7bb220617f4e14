import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from spectrogram_inversion_amd.plan import Plan, args_helper
dev = torch.device("cuda", 0)
n_fft, hop = 2048, 512
hann = lambda n, dt: torch.from_numpy((0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n) / n)).astype(dt))
for frames in (6, 8, 40):
    mag = torch.from_numpy(np.random.default_rng(2048+frames).random((1, 1025, frames), dtype=np.float32)).to(dev)
    a32 = args_helper(torch.empty((1, 1025, 1)), hop_length=hop, window=hann(n_fft, np.float32))
    a64 = args_helper(torch.empty((1, 1025, 1), dtype=torch.float64), hop_length=hop, window=hann(n_fft, np.float64))
    fast = Plan(a32, 1, frames, torch.float32, dev)
    gen = Plan(a32, 1, frames, torch.float32, dev); gen.force_generic(True)
    ref = Plan(a64, 1, frames, torch.float64, dev)
    init = fast.phase_init(mag)
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    for its in (1, 10):
        out = {}
        for name, p, x0 in (("fast", fast, init), ("gen", gen, init), ("f64", ref, init.to(torch.complex128))):
            p.gla_init(x0, None, 0.3); p.iterate(its); out[name] = p.wave()
        print(f"T={frames} iters={its:3d} fast-vs-f64 {rel(out['fast'], out['f64']):.3e}  generic-vs-f64 {rel(out['gen'], out['f64']):.3e}", flush=True)
