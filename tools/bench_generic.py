#!/usr/bin/env python3
"""ms per iteration of the coverage path (`force_generic`: kernels_wave.h where it applies since round 6 - n_fft 128 ... 16384 and 400 / 800 / 1000 -
kernels_generic.h elsewhere: other odd sizes): float32 / float64, one- and two-sided, Griffin-Lim
and ADMM, with the HBM fraction of 8 hop + 20 F + 8 N elements per frame and iteration (ADMM: 36 F) - the bytes of the frames + k_ola
form, so that rounds compare; the register overlap-add moves 8 N fewer - and the kernel that ran.  A/B through the environment:
SPECINV_GENERIC_WAVE=0 keeps the workgroup-level kernels, SPECINV_WAVE_OLA=0 the frames buffer, SPECINV_GENERIC_DR=0 the Stockham
kernels.  (One script since round 6: the r04 / r05 copies are in the history.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from spectrogram_inversion_amd.plan import Plan, args_helper

dev = torch.device("cuda", 0)
CASES = [  # n_fft, win_length, hop, frames, batch, dtype, onesided, method
    (2048, None, 512, 1024, 16, torch.float64, True, "gla"),
    (2048, None, 512, 1024, 16, torch.float64, True, "admm"),
    (4096, None, 1024, 512, 16, torch.float64, True, "gla"),
    (1024, None, 256, 2048, 16, torch.float64, True, "gla"),
    (512, None, 128, 2048, 32, torch.float64, True, "gla"),
    (512, 300, 100, 2048, 64, torch.float64, False, "gla"),
    (512, 300, 100, 2048, 64, torch.float32, False, "gla"),
    (256, None, 64, 4096, 64, torch.float32, True, "gla"),
    (1024, None, 256, 2048, 32, torch.float32, True, "gla"),
    (2048, None, 512, 1024, 32, torch.float32, False, "gla"),
    (8192, None, 2048, 256, 16, torch.float32, True, "gla"),
    (16384, None, 4096, 128, 16, torch.float32, True, "gla"),
    (8192, None, 2048, 128, 16, torch.float64, True, "gla"),
    (128, None, 32, 4096, 64, torch.float32, True, "gla"),
    (256, None, 64, 4096, 64, torch.float64, True, "gla"),
    (1000, None, 250, 1024, 16, torch.float64, True, "gla"),         # the sizes that are not 128 * 2^k
    (1000, None, 250, 1024, 16, torch.float32, True, "gla"),
    (400, None, 160, 2048, 64, torch.float32, True, "gla"),
    (400, None, 160, 2048, 64, torch.float64, True, "gla"),
    (800, None, 200, 2048, 32, torch.float32, True, "gla"),
    (400, None, 100, 2048, 64, torch.float32, True, "admm"),
    (512, 300, 100, 2048, 64, torch.float64, True, "gla"),           # hops that divide nothing: the LDS ring
    (1024, 800, 200, 2048, 32, torch.float64, True, "gla"),
    (256, 200, 50, 4096, 64, torch.float32, True, "gla"),
]
for n_fft, wl, hop, frames, batch, dtype, onesided, method in CASES:
    F = n_fft // 2 + 1 if onesided else n_fft
    w = torch.hann_window(wl or n_fft, dtype=dtype)
    mag = torch.rand((batch, F, frames), dtype=dtype, device=dev)
    kw = dict(hop_length=hop, window=w, onesided=onesided)
    if wl:
        kw["win_length"] = wl
    plan = Plan(args_helper(mag, **kw), batch, frames, dtype, dev)
    plan.force_generic(True)
    (plan.gla_init if method == "gla" else plan.admm_init)(None, mag, 0.3 if method == "gla" else 0.1)
    plan.iterate(3)
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        plan.iterate(20)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    es = 4 if dtype == torch.float32 else 8
    per_frame = (2 * hop + (5 if method == "gla" else 9) * F + 2 * n_fft) * es
    gbs = per_frame * batch * frames / (best * 1e-3) / 1e9
    print(f"{method:4s} n_fft {n_fft:5d} win {wl or n_fft:5d} hop {hop:5d} T {frames:5d} B {batch:3d} {str(dtype)[6:]:8s} onesided={onesided!s:5s} "
          f"{best:8.3f} ms/it {batch * frames / best / 1e3:8.1f} M frames/s {100 * gbs / 8000:5.1f} % of 8 TB/s  {plan.launch_geometry['kernel']}"
          f"{' (overlap-add: ' + plan.launch_geometry['overlap_add'] + ')' if plan.launch_geometry['kernel'] == 'k_wave_iter' else ''}",
          flush=True)
    del plan

# the same two-sided float32 shapes where they run since round 5: the wave-level frame kernels k_hop2 / k_semi2 + k_ola
print("two-sided float32 on the frame kernels (k_hop2 at these frame counts; below 12 k / 32 k frames k_semi2 + k_ola_f4):", flush=True)
for n_fft, wl, hop, frames, batch, method in [(512, 300, 100, 2048, 64, "gla"), (512, 300, 100, 2048, 64, "admm"),
                                              (2048, None, 512, 1024, 32, "gla"), (1024, None, 256, 2048, 32, "gla")]:
    dtype = torch.float32
    w = torch.hann_window(wl or n_fft, dtype=dtype)
    mag = torch.rand((batch, n_fft, frames), dtype=dtype, device=dev)
    kw = dict(hop_length=hop, window=w, onesided=False)
    if wl:
        kw["win_length"] = wl
    res = {}
    for arm in ("frame", "coverage"):
        plan = Plan(args_helper(mag, **kw), batch, frames, dtype, dev)
        plan.force_generic(arm == "coverage")
        (plan.gla_init if method == "gla" else plan.admm_init)(None, mag, 0.3 if method == "gla" else 0.1)
        plan.iterate(3)
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            plan.iterate(20)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20)
        res[arm] = best
        del plan
    per_frame = (2 * hop + (5 if method == "gla" else 5) * n_fft + 2 * n_fft) * 4       # (ADMM carries Y alone on the frame kernel)
    gbs = per_frame * batch * frames / (res["frame"] * 1e-3) / 1e9
    print(f"{method:4s} n_fft {n_fft:5d} win {wl or n_fft:5d} hop {hop:5d} T {frames:5d} B {batch:3d} frame kernel {res['frame']:7.3f} ms/it "
          f"({100 * gbs / 8000:5.1f} % of 8 TB/s on (8h+20F+8N)/4 elements)   coverage kernels {res['coverage']:7.3f} ms/it", flush=True)

