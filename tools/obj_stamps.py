#!/usr/bin/env python3
"""Per-phase cycle counts of the one-launch log-mel objective (diagnostic build -DSPECINV_OBJ_STAMPS=1, s_memtime stamps by
wave 0 of every workgroup; the fifth evaluation prints the table to stderr)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.environ.get("SPECINV_STAMP_LIB") or os.path.join(ROOT, "spectrogram_inversion_amd", "variants", "libspecinv_stamps.so")
if not os.path.exists(lib):
    raise SystemExit(f"build it first: build_lib(extra_flags=['-DSPECINV_OBJ_STAMPS=1'], out='{lib}')")
os.environ["SPECINV_LIB"] = lib
import numpy as np, torch
import spectrogram_inversion_amd as si
dev = torch.device("cuda", 0)
B, T, n_fft, hop = 16, 1024, 2048, 512
w = torch.from_numpy((0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)).astype(np.float32))
fb = torch.from_numpy(si.mel_filterbank(22050, n_fft, 80)).to(dev)
tr = si.LogMelSTFT(fb, n_fft, hop_length=hop, window=w)
xs = 0.1 * torch.randn(B, (T - 1) * hop, device=dev)
target = tr(xs)
x0 = 1e-3 * torch.randn_like(xs)
_, fg = tr.bind(x0, target)
for _ in range(6):
    fg(x0)
torch.cuda.synchronize()
