#!/usr/bin/env python3
"""ms of the log-mel forward transform at the C5 shape (dev tool; SPECINV_LIB selects a variant)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spectrogram_inversion_amd as si
from spectrogram_inversion_amd.mel import mel_filterbank
dev = torch.device("cuda", 0)
B, T, n_fft, hop = 16, 1024, 2048, 512
fb = torch.from_numpy(mel_filterbank(22050, n_fft, 80)).float().to(dev)
tf = si.LogMelSTFT(fb, n_fft, hop_length=hop, window=torch.hann_window(n_fft))
x = 0.1 * torch.randn(B, (T - 1) * hop, device=dev)
for _ in range(3):
    v = tf(x)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(20):
    v = tf(x)
e1.record(); torch.cuda.synchronize()
print(f"{os.path.basename(os.environ.get('SPECINV_LIB', 'default'))}: forward {e0.elapsed_time(e1) / 20:.3f} ms")
