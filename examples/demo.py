#!/usr/bin/env python3
"""The reference's demo (main.py) without its audio / plotting dependencies: a synthetic 10 s signal, the same STFT
arguments (n_fft 1024, hop 128, Hann), and every method of the package on its magnitude spectrogram, each scored by
spectral convergence (dB, lower is better).  Needs an MI355X.

    python examples/demo.py [--seconds 10] [--iters 100]
"""
import argparse
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import spectrogram_inversion_amd as specinv
from spectrogram_inversion_amd import MagSTFT, RTISIStream, sc


def chirpy_signal(sr, seconds, device):
    """A few drifting partials plus noise bursts: something with structure in time and frequency."""
    t = torch.arange(int(sr * seconds), device=device) / sr
    x = torch.zeros_like(t)
    for k, f0 in enumerate((220.0, 330.0, 495.0, 880.0)):
        f = f0 * (1.0 + 0.1 * torch.sin(2 * math.pi * (0.2 + 0.1 * k) * t))
        x += torch.sin(2 * math.pi * torch.cumsum(f, 0) / sr) / (k + 1)
    x += 0.05 * torch.randn(t.shape, device=device, generator=torch.Generator(device=device).manual_seed(0)) * \
        (torch.sin(2 * math.pi * 1.5 * t) > 0.9)
    return 0.3 * x


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--iters", type=int, default=100)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    sr, n_fft, hop = 22050, 1024, 128                                  # main.py:12-14
    y = chirpy_signal(sr, args.seconds, dev)
    window = torch.hann_window(n_fft, device=dev)
    kw = dict(win_length=n_fft, window=window, hop_length=hop, pad_mode="reflect", onesided=True, normalized=False,
              center=True)                                               # main.py:26-34
    spec = torch.stft(y, n_fft, return_complex=True, **kw).abs()

    def score(name, est, dt):
        s = torch.stft(est, n_fft, return_complex=True, **kw).abs()
        n = min(s.shape[-1], spec.shape[-1])
        print(f"{name:34s} SC {float(sc(s[..., :n], spec[..., :n])):7.2f} dB   {dt * 1e3:8.1f} ms", flush=True)

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        return out, time.perf_counter() - t0

    print(f"{spec.shape[0]} bins x {spec.shape[1]} frames, {args.iters} iterations")
    score("phase_init only", torch.istft(specinv.phase_init(spec, **kw), n_fft, hop_length=hop, window=window), 0.0)
    est, dt = timed(lambda: specinv.griffin_lim(spec, max_iter=args.iters, alpha=0.3, tol=0, verbose=False, **kw))
    score("griffin_lim(alpha=0.3)", est, dt)                             # main.py:44
    est, dt = timed(lambda: specinv.ADMM(spec, max_iter=args.iters, rho=0.2, tol=0, verbose=False, **kw))
    score("ADMM(rho=0.2)", est, dt)                                      # main.py:45
    est, dt = timed(lambda: specinv.RTISI_LA(spec, max_iter=4, look_ahead=3, asymmetric_window=True, verbose=False, **kw))
    score("RTISI_LA(4 it, look_ahead=3)", est, dt)                       # main.py:47
    stream = RTISIStream(spec.shape[0], look_ahead=3, asymmetric_window=True, max_iter=4, max_push=8, device=dev, **kw)
    (pieces, dt) = timed(lambda: [stream.push(spec[:, t:t + 8]) for t in range(0, spec.shape[1], 8)] + [stream.flush()])
    score("RTISIStream, 8 frames per push", torch.cat(pieces), dt)
    func = MagSTFT(n_fft, **kw)                                          # main.py:22-37 (p = 1)
    est, dt = timed(lambda: specinv.L_BFGS(spec, func, [len(y)], outer_max_iter=max(1, args.iters // 50), max_iter=50, lr=1,
                                           history_size=10, eva_iter=1, verbose=False,
                                           line_search_fn="strong_wolfe"))   # main.py:43, with the line search that makes it converge
    score("L_BFGS(MagSTFT, strong_wolfe)", est, dt)


if __name__ == "__main__":
    main()
