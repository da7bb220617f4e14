#!/usr/bin/env python3
"""Device-resident L-BFGS against the host-driven loop, step by step (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import spectrogram_inversion_amd as si
from spectrogram_inversion_amd.lbfgs import LBFGS
from spectrogram_inversion_amd.transforms import LogMelSTFT, MagSTFT
from _util import hann, rel_l2
dev = torch.device("cuda", 0)
def problem(kind, seed=7):
    rng = np.random.default_rng(seed)
    if kind == "logmel":
        n_fft, hop, frames, batch = 2048, 512, 24, 2
        tr = LogMelSTFT(torch.from_numpy(si.mel_filterbank(22050, n_fft, 80)).to(dev), n_fft, hop_length=hop, window=torch.from_numpy(hann(n_fft)))
    else:
        n_fft, hop, frames, batch = 1024, 256, 30, 3
        tr = MagSTFT(n_fft, hop_length=hop, window=torch.from_numpy(hann(n_fft)))
    length = (frames - 1) * hop
    xs = torch.from_numpy((0.1 * rng.standard_normal((batch, length))).astype(np.float32)).to(dev)
    x0 = torch.from_numpy((1e-2 * rng.standard_normal((batch, length))).astype(np.float32)).to(dev)
    return tr, tr(xs), x0
def run(device_path, tr, target, x0, steps, **kw):
    os.environ["SPECINV_LBFGS_DEVICE"] = "1" if device_path else "0"
    x = x0.clone(); _, fg = tr.bind(x, target); opt = LBFGS(x, device=dev, **kw)
    out = []
    for _ in range(steps):
        l = opt.step(fg)
        out.append((l, x.clone(), opt.total_iters, opt.func_evals, int(opt.pairs_accepted), int(opt.pairs_rejected), opt.history_len))
    return out
for kind, kw, steps in [("mag", dict(max_iter=1), 4), ("mag", dict(max_iter=2), 3), ("mag", dict(max_iter=3), 3), ("mag", dict(max_iter=10), 2), ("logmel", dict(max_iter=4), 3), ("logmel", dict(), 3),
                        ("logmel", dict(max_iter=12, history_size=3), 4)]:
    tr, target, x0 = problem(kind)
    a = run(False, tr, target, x0, steps, **kw); b = run(True, tr, target, x0, steps, **kw)
    print(kind, kw)
    for i, (p, q) in enumerate(zip(a, b)):
        print("  step", i, "host", p[2:], "loss %.6e" % p[0], "| dev", q[2:], "loss %.6e" % q[0], "| rel x %.2e" % rel_l2(q[1].cpu().numpy(), p[1].cpu().numpy()))
