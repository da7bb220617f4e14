#!/usr/bin/env python3
"""Forward + backward time of the differentiable methods at a training-like shape (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spectrogram_inversion_amd as si
dev = torch.device("cuda", 0)
B, n_fft, hop, T, iters = 16, 1024, 256, 256, 32
w = torch.hann_window(n_fft, device=dev)
mag = (torch.rand(B, n_fft // 2 + 1, T, device=dev) + 0.05)
def run(fn, **kw):
    spec = mag.clone().requires_grad_(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    y = fn(spec, verbose=False, hop_length=hop, window=w, **kw)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    y.square().mean().backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3
for name, fn, kw in (("griffin_lim", si.griffin_lim, dict(max_iter=iters, alpha=0.3, tol=0)),
                     ("ADMM", si.ADMM, dict(max_iter=iters, rho=0.2, tol=0)),
                     ("RTISI_LA", si.RTISI_LA, dict(max_iter=4, look_ahead=3, asymmetric_window=True))):
    run(fn, **kw)
    f, b = run(fn, **kw)
    with torch.no_grad():
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(mag, verbose=False, hop_length=hop, window=w, **kw); torch.cuda.synchronize()
        inf = (time.perf_counter() - t0) * 1e3
    print(f"{name:12s} B{B} n_fft {n_fft} hop {hop} T{T}: recorded forward {f:7.2f} ms  backward {b:7.2f} ms  (inference {inf:6.2f} ms)")
