#!/usr/bin/env python3
"""One coverage-path configuration, a few iterations: the process tools/prof_generic.sh wraps in rocprofv3 for
profiles/r04_generic_{kernel_stats.csv,pmc.json}.  usage: bench_generic_one.py <n_fft> <hop> <frames> <batch> <f32|f64> [twosided|onesided] [win_length]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from spectrogram_inversion_amd.plan import Plan, args_helper

n_fft, hop, frames, batch = (int(v) for v in sys.argv[1:5])
dtype = torch.float64 if sys.argv[5] == "f64" else torch.float32
onesided = not (len(sys.argv) > 6 and sys.argv[6] == "twosided")
dev = torch.device("cuda", 0)
F = n_fft // 2 + 1 if onesided else n_fft
mag = torch.rand((batch, F, frames), dtype=dtype, device=dev)
wl = int(sys.argv[7]) if len(sys.argv) > 7 else n_fft
kw = dict(hop_length=hop, window=torch.hann_window(wl, dtype=dtype), onesided=onesided)
if wl != n_fft:
    kw["win_length"] = wl
plan = Plan(args_helper(mag, **kw), batch, frames, dtype, dev)
plan.force_generic(True)
plan.gla_init(None, mag, 0.3)
plan.iterate(12)
torch.cuda.synchronize()
