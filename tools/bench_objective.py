#!/usr/bin/env python3
"""ms per evaluation of the one-launch log-mel objective (loss + gradient) over n_fft / number of mel bands (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spectrogram_inversion_amd as si
from spectrogram_inversion_amd.mel import mel_filterbank
dev = torch.device("cuda", 0)
B, T = 16, 1024
for n_fft, hop in ((2048, 512), (1024, 256)):
    for n_mels in (40, 64, 80, 96, 112, 128):
        fb = torch.from_numpy(mel_filterbank(22050, n_fft, n_mels)).float().to(dev)
        tf = si.LogMelSTFT(fb, n_fft, hop_length=hop, window=torch.hann_window(n_fft))
        xs = 0.1 * torch.randn(B, (T - 1) * hop, device=dev)
        target = tf(xs)
        x0 = 1e-3 * torch.randn_like(xs)
        _, fg = tf.bind(x0, target)
        for _ in range(3):
            fg(x0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20):
            fg(x0)
        e1.record(); torch.cuda.synchronize()
        print(f"n_fft {n_fft} hop {hop} mels {n_mels}: {e0.elapsed_time(e1) / 20:.3f} ms per evaluation (B {B}, T {T})")
