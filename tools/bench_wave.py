#!/usr/bin/env python3
"""The wave-level coverage kernel (csrc/kernels_wave.h) against the workgroup-level ones it replaces (k_iter_pair / k_iter_pair_dr),
arms switched by SPECINV_GENERIC_WAVE inside one run: agreement of waveform, state and evaluation sums after a few iterations, then
ms per iteration with the HBM fraction of 8 hop + 20 F + 8 N elements per frame and iteration (ADMM: 36 F).
`bench_wave.py check` only checks (small shapes, every size / dtype / sidedness); `bench_wave.py` times the shapes of
profiles/r05_generic.txt."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from spectrogram_inversion_amd.plan import Plan, args_helper

dev = torch.device("cuda", 0)


def make(n_fft, wl, hop, frames, batch, dtype, onesided, method, arm, mag, init=None, **extra):
    os.environ["SPECINV_GENERIC_WAVE"] = "0" if arm == "ref" else "1"
    os.environ["SPECINV_WAVE_OLA"] = "0" if arm == "wave_frames" else "1"     # (overlap-add in registers where it applies / frames + k_ola)
    w = torch.hann_window(wl or n_fft, dtype=dtype)
    kw = dict(hop_length=hop, window=w, onesided=onesided, **extra)
    if wl:
        kw["win_length"] = wl
    plan = Plan(args_helper(mag, **kw), batch, frames, dtype, dev)
    plan.force_generic(True)
    (plan.gla_init if method == "gla" else plan.admm_init)(init, mag if init is None else None, 0.3 if method == "gla" else 0.1)
    return plan


def rel(a, b):
    """relative L2 distance over the finite samples; the non-finite ones (the reference's 0 / 0 where the envelope vanishes,
    methods.py:132) must sit in the same places"""
    fa, fb = torch.isfinite(a), torch.isfinite(b)
    if not torch.equal(fa, fb):
        return float("inf")
    a, b = torch.where(fa, a, torch.zeros_like(a)), torch.where(fb, b, torch.zeros_like(b))
    return float((a - b).norm() / b.norm())


def check():
    bad = 0
    torch.manual_seed(3)
    for dtype in (torch.float32, torch.float64):
        for n_fft in (128, 256, 512, 1024, 2048, 400, 800, 1000, 4096, 8192, 16384):
            if n_fft == 16384 and dtype == torch.float64:
                continue
            for onesided, hop, frames, batch, extra in ((True, n_fft // 4, 21, 3, {}), (False, n_fft // 4 + 3, 10, 2, {}),
                                                        (True, n_fft // 2, 9, 1, dict(center=False)),
                                                        (True, n_fft // 8, 13, 2, dict(normalized=True, pad_mode="constant")),
                                                        (True, n_fft // 8, 45, 3, dict(pad_mode="replicate")), (True, n_fft // 2, 37, 2, {}),
                                                        (True, n_fft // 4, 64, 5, dict(center=False))):
                for method in ("gla", "admm"):
                    F = n_fft // 2 + 1 if onesided else n_fft
                    mag = torch.rand((batch, F, frames), dtype=dtype, device=dev) + 0.05
                    cd = torch.complex64 if dtype == torch.float32 else torch.complex128
                    init = (mag * torch.exp(1j * 6.28 * torch.rand(mag.shape, device=dev, dtype=dtype))).to(cd)
                    out = {}
                    for arm in ("wave", "ref"):
                        p = make(n_fft, None, hop, frames, batch, dtype, onesided, method, arm, mag, init, **extra)
                        assert p.launch_geometry["kernel"] == ("k_wave_iter" if arm == "wave" else "k_iter_pair"), p.launch_geometry
                        p.iterate(2)
                        s = p.iterate(1, eval_last=True)
                        out[arm] = (p.wave().clone(), p.state_spec(0).clone(), np.array(s[:2]))
                        del p
                    tol = (5e-5 if method == 'admm' else 2e-5) if dtype == torch.float32 else 1e-12   # (rho = 0.1 amplifies rounding ~10x per iteration)
                    e = (rel(out["wave"][0], out["ref"][0]), rel(out["wave"][1], out["ref"][1]),
                         float(np.nanmax(np.abs(out["wave"][2] / out["ref"][2] - 1))) if np.isfinite(out["ref"][2]).any() else 0.0)
                    ok = e[0] < tol and e[1] < tol and e[2] < max(tol, 1e-6)
                    bad += not ok
                    print(f"{'ok ' if ok else 'BAD'} {str(dtype)[6:]:8s} n_fft {n_fft:5d} hop {hop:4d} onesided={onesided!s:5s} {method:4s} {extra} "
                          f"wave {e[0]:.2e} state {e[1]:.2e} sums {e[2]:.2e}", flush=True)
    print("check:", "all ok" if not bad else f"{bad} BAD")
    return bad


CASES = [  # n_fft, win_length, hop, frames, batch, dtype, onesided, method
    (2048, None, 512, 1024, 16, torch.float64, True, "gla"),
    (2048, None, 512, 1024, 16, torch.float64, True, "admm"),
    (1024, None, 256, 2048, 16, torch.float64, True, "gla"),
    (512, None, 128, 2048, 32, torch.float64, True, "gla"),
    (512, 300, 100, 2048, 64, torch.float64, False, "gla"),
    (256, None, 64, 4096, 64, torch.float64, True, "gla"),
    (256, None, 64, 4096, 64, torch.float32, True, "gla"),
    (128, None, 32, 4096, 64, torch.float32, True, "gla"),
    (256, None, 64, 4096, 64, torch.float32, True, "admm"),
    (512, 300, 100, 2048, 64, torch.float32, False, "gla"),
    (1024, None, 256, 2048, 32, torch.float32, True, "gla"),
    (2048, None, 512, 1024, 32, torch.float32, False, "gla"),
    (512, 300, 100, 2048, 64, torch.float32, True, "gla"),      # 12 ...: hops that do not divide n_fft (the LDS ring)
    (512, 300, 100, 2048, 64, torch.float64, True, "gla"),
    (1024, 800, 200, 2048, 32, torch.float32, True, "gla"),
    (2048, 1200, 300, 1024, 32, torch.float32, True, "admm"),
    (256, 200, 50, 4096, 64, torch.float32, True, "gla"),
    (1024, 800, 200, 2048, 16, torch.float64, True, "gla"),
    (256, 200, 50, 4096, 64, torch.float64, True, "gla"),
    (400, None, 160, 2048, 64, torch.float32, True, "gla"),     # 19 ...: n_fft 400 / 800 / 1000 (profiles/r06_generic.txt's shapes first)
    (1000, None, 250, 1024, 16, torch.float32, True, "gla"),
    (1000, None, 250, 1024, 16, torch.float64, True, "gla"),
    (400, None, 100, 2048, 64, torch.float32, True, "admm"),
    (800, None, 200, 2048, 32, torch.float32, True, "gla"),
    (400, None, 160, 2048, 64, torch.float64, True, "gla"),
    (800, None, 200, 2048, 32, torch.float64, True, "gla"),
    (1000, None, 250, 2048, 32, torch.float32, False, "gla"),
    (400, None, 160, 2048, 64, torch.float32, False, "gla"),
    (400, None, 160, 2048, 32, torch.float64, False, "gla"),
    (1000, 800, 200, 1024, 32, torch.float64, True, "admm"),
    (4096, None, 1024, 512, 16, torch.float64, True, "gla"),     # 30 ...: n_fft 4096 on teams of four / two waves
    (4096, None, 1024, 512, 32, torch.float32, True, "gla"),
    (4096, 3000, 1000, 512, 16, torch.float64, True, "admm"),
    (4096, None, 1024, 512, 16, torch.float64, False, "gla"),
    (8192, None, 2048, 256, 16, torch.float32, True, "gla"),     # 34 ...: n_fft 8192 (profiles/r06_generic.txt's shapes)
    (8192, None, 2048, 128, 16, torch.float64, True, "gla"),
    (8192, 6000, 1500, 256, 16, torch.float32, True, "admm"),
    (4096, 3000, 1000, 512, 32, torch.float32, True, "gla"),
    (4096, None, 1024, 512, 32, torch.float32, False, "gla"),
    (8192, None, 2048, 256, 16, torch.float32, False, "gla"),
    (16384, None, 4096, 128, 16, torch.float32, True, "gla"),    # 40 ...: n_fft 16384, float32
    (16384, None, 2048, 128, 16, torch.float32, True, "admm"),
]


def bench(cases=CASES):
    for n_fft, wl, hop, frames, batch, dtype, onesided, method in cases:
        F = n_fft // 2 + 1 if onesided else n_fft
        mag = torch.rand((batch, F, frames), dtype=dtype, device=dev)
        res = {}
        for arm in ("ref", "wave_frames", "wave"):
            plan = make(n_fft, wl, hop, frames, batch, dtype, onesided, method, arm, mag)
            geo = plan.launch_geometry
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
            res[arm] = (best, geo)
            del plan
        es = 4 if dtype == torch.float32 else 8
        per_frame = (2 * hop + (5 if method == "gla" else 9) * F + 2 * n_fft) * es
        frac = {k: per_frame * batch * frames / (v[0] * 1e-3) / 8e12 for k, v in res.items()}
        g = res["wave"][1]
        print(f"{method:4s} n_fft {n_fft:5d} win {wl or n_fft:5d} hop {hop:5d} T {frames:5d} B {batch:3d} {str(dtype)[6:]:8s} onesided={onesided!s:5s} "
              f"k_iter_pair {res['ref'][0]:7.3f} ms/it ({100 * frac['ref']:4.1f} %)   k_wave_iter + k_ola {res['wave_frames'][0]:7.3f} ({100 * frac['wave_frames']:4.1f} %)   "
              f"k_wave_iter {res['wave'][0]:7.3f} ms/it ({100 * frac['wave']:4.1f} % of 8 TB/s; "
              f"{g['waves']} waves, {g['waves_per_workgroup']} per workgroup)", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "check":
        sys.exit(1 if check() else 0)
    if len(sys.argv) > 1 and sys.argv[1] == "one":          # one or more cases by index (for a profiler run)
        bench([CASES[int(i)] for i in sys.argv[2:]])
        sys.exit(0)
    bench()
