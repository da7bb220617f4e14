#!/usr/bin/env python3
"""Per-phase cycle counts of k_fused4_td (diagnostic build -DSPECINV_TD_STAMPS=1: every wave sums s_memtime differences per
phase; iterations 5 and 40 print the table to stderr).  C2 geometry by default."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "spectrogram_inversion_amd", "variants", "libspecinv_tdstamps.so")
if not os.path.exists(lib):
    raise SystemExit(f"build it first: build_lib(extra_flags=['-DSPECINV_TD_STAMPS=1'], out='{lib}')")
os.environ["SPECINV_LIB"] = lib
import numpy as np, torch, time
from spectrogram_inversion_amd.plan import Plan, args_helper
dev = torch.device("cuda", 0)
n_fft = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
B, T = (64, 1024) if n_fft == 2048 else (32, 2048)
hop = n_fft // 4
w = torch.from_numpy((0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)).astype(np.float32))
mag = torch.rand(B, n_fft // 2 + 1, T, device=dev)
p = Plan(args_helper(mag, hop_length=hop, window=w), B, T, torch.float32, dev)
p.gla_init(None, mag, 0.3)
print(p.launch_geometry, file=sys.stderr)
p.iterate(39)
p.iterate(1)
torch.cuda.synchronize()
t0 = time.perf_counter()
p.iterate(20)
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 20 * 1e3:.4f} ms per iteration (stamped build)", file=sys.stderr)
