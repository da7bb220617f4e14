#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configurations (C1, C3, C4 shard, C5) on one MI355X (dev tool;
the driver's headline benchmark is bench.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spectrogram_inversion_amd as si
from spectrogram_inversion_amd.plan import args_helper, get_plan

dev = torch.device("cuda", 0)
hann = lambda n: torch.from_numpy((0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n) / n)).astype(np.float32))

def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best

which = sys.argv[1:] or ["C1", "C4", "C3", "C5"]
rng = np.random.default_rng(1234)
if "C1" in which:
    mag = torch.from_numpy(rng.random((1, 513, 512), dtype=np.float32)).to(dev)
    dt = timed(lambda: si.griffin_lim(mag, max_iter=50, alpha=0.0, tol=0, verbose=False, hop_length=256, window=hann(1024)))
    print(f"C1 griffin_lim B=1 n_fft=1024 hop=256 T=512 50 it alpha=0: {dt*1e3:.2f} ms  {50*512/dt/1e6:.2f} M it*frames/s")
if "C4" in which:
    B, T = 32, 2048
    mag = torch.from_numpy(rng.random((B, 513, T), dtype=np.float32)).to(dev)
    dt = timed(lambda: si.ADMM(mag, max_iter=200, rho=0.1, tol=0, verbose=False, hop_length=256, window=hann(1024)))
    units = 200 * B * T
    print(f"C4 ADMM shard B={B} n_fft=1024 hop=256 T={T} rho=0.1 200 it: {dt*1e3:.1f} ms  {units/dt/1e6:.1f} M it*frames/s  "
          f"{units*(8*256+36*513)/dt/1e12:.2f} TB/s algorithmic")
if "C3" in which:
    B, T = 32, 1024
    mag = torch.from_numpy(rng.random((B, 1025, T), dtype=np.float32)).to(dev)
    for asym in (True, False):
        dt = timed(lambda: si.RTISI_LA(mag, look_ahead=3, asymmetric_window=asym, max_iter=25, verbose=False,
                                       hop_length=512, window=hann(2048)), reps=1)
        print(f"C3 RTISI_LA B={B} n_fft=2048 hop=512 T={T} LA=3 25 it asym={asym}: {dt:.2f} s  {25*B*T/dt/1e3:.1f} k it*frames/s  "
              f"{(T+3)*25/dt:.0f} dependent steps/s")
if "C5" in which:
    B, T, n_fft, hop = 16, 1024, 2048, 512
    L = (T - 1) * hop
    fb = torch.from_numpy(si.mel_filterbank(22050, n_fft, 80)).to(dev)
    tr = si.LogMelSTFT(fb, n_fft, hop_length=hop, window=hann(n_fft))
    xs = (0.1 * torch.randn(B, L, device=dev))
    target = tr(xs)
    x0 = 1e-6 * torch.randn(B, L, device=dev)
    fwd, fg = tr.bind(x0, target)
    dt = timed(lambda: fg(x0), reps=5)
    print(f"C5 log-mel fwd+loss+grad B={B} T={T}: {dt*1e3:.2f} ms per evaluation  {B*T/dt/1e6:.2f} M evals*frames/s")
    for kw, tag in ((dict(), "defaults (max_iter=20, history=100)"), (dict(max_iter=50, history_size=10), "main.py:43 variant")):
        dt = timed(lambda: si.L_BFGS(target, tr, init_x0=x0.clone(), outer_max_iter=2, tol=0, eva_iter=10, verbose=False, **kw), reps=1)
        print(f"C5 L_BFGS 2 outer steps, {tag}: {dt:.3f} s")
    trm = si.MagSTFT(n_fft, hop_length=hop, window=hann(n_fft))           # the objective of the reference's demo (main.py:22-43)
    fwd, fg = trm.bind(x0, trm(xs))
    dt = timed(lambda: fg(x0), reps=5)
    print(f"C5' magnitude fwd+loss+grad B={B} T={T}: {dt*1e3:.2f} ms per evaluation  {B*T/dt/1e6:.2f} M evals*frames/s")
