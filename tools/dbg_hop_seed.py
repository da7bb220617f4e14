#!/usr/bin/env python3
"""One seed of tests/test_gpu_semi.py::test_chunked_frame_kernel_random_shapes: the chunked frame kernel (k_hop) and
the frame-at-a-time kernel (k_semi + k_ola) against the float64 oracle, error per hop-block (dev tool)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import oracle
from _util import hann
from spectrogram_inversion_amd.plan import Plan, args_helper, clear_plan_cache

seed = int(sys.argv[1])
rng = np.random.default_rng(4000 + seed)
n_fft = int(rng.choice([512, 1024, 2048]))
hop = int(rng.choice([int(rng.integers(n_fft // 16, n_fft + 1)), n_fft // 3, n_fft // 5, n_fft // 2 + 1]))
frames = int(rng.integers(20, 260))
batch = int(rng.integers(1, 4))
center = bool(rng.random() < 0.75)
pad_mode = str(rng.choice(["reflect", "constant", "replicate", "circular"]))
method = "gla" if rng.random() < 0.6 else "admm"
w = (hann(n_fft) + np.float32(0.05)) if rng.random() < 0.7 else np.ones(n_fft, dtype=np.float32)
mag = rng.random((batch, n_fft // 2 + 1, frames), dtype=np.float32) + 0.02
print(n_fft, hop, frames, batch, center, pad_mode, method, "rect" if w[0] == 1 else "hann+0.05")
kw = dict(hop_length=hop, window=w, center=center, pad_mode=pad_mode)
init = oracle.phase_init(mag, **kw)
fn = oracle.griffin_lim if method == "gla" else oracle.admm
ck = dict(alpha=0.3) if method == "gla" else dict(rho=0.5)
for iters in (1, 2, 3):
    ref64 = fn(init.astype(np.complex128), max_iter=iters, tol=0, **ck, **{**kw, "window": w.astype(np.float64)})
    ref32 = fn(init, max_iter=iters, tol=0, **ck, **kw)
    out = {}
    for name, env in (("k_hop", "0"), ("k_semi", "1")):
        os.environ["SPECINV_SMALL_FRAMES"] = "0"
        os.environ["SPECINV_DISABLE_HOP"] = env
        clear_plan_cache()
        p = Plan(args_helper(torch.empty(1, n_fft // 2 + 1, 1), hop_length=hop, window=torch.from_numpy(w), center=center,
                             pad_mode=pad_mode), batch, frames, torch.float32, torch.device("cuda", 0))
        (p.gla_init if method == "gla" else p.admm_init)(torch.from_numpy(init).cuda(), None, 0.3 if method == "gla" else 0.5)
        p.iterate(iters)
        out[name] = p.wave().cpu().numpy()
    ref64 = np.asarray(ref64).reshape(out["k_hop"].shape)
    ref32 = np.asarray(ref32).reshape(ref64.shape)
    scale = np.abs(ref64[np.isfinite(ref64)]).max()
    n = (ref64.shape[1] // hop) * hop
    for name, y in list(out.items()) + [("oracle32", ref32)]:
        e = np.nan_to_num(np.abs(y - ref64))[:, :n].reshape(batch, -1, hop).max(-1) / scale
        worst = np.argsort(e.reshape(-1))[-4:][::-1]
        print(iters, f"{name:9s} max {e.max():.2e} median {np.median(e):.2e}  worst blocks", [(int(i // e.shape[1]), int(i % e.shape[1]), f"{e.reshape(-1)[i]:.1e}") for i in worst])
