#!/usr/bin/env python3
"""A/B of k_fused4_td library variants on one box: ms per late iteration at the C2 (or C4-like) geometry, interleaved rounds.
usage: td_variants.py <n_fft> <variant.so> ... (the shipped library is always included)"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch, time
    from spectrogram_inversion_amd.plan import Plan, args_helper
    dev = torch.device("cuda", 0)
    n_fft = int(sys.argv[2])
    B, T = (64, 1024) if n_fft == 2048 else (32, 2048)
    hop = n_fft // 4
    w = torch.from_numpy((0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)).astype(np.float32))
    mag = torch.rand(B, n_fft // 2 + 1, T, device=dev)
    p = Plan(args_helper(mag, hop_length=hop, window=w), B, T, torch.float32, dev)
    p.gla_init(None, mag, 0.3)
    p.iterate(40)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        p.iterate(50)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 50 * 1e3)
    print(json.dumps(best))
    sys.exit(0)
n_fft = sys.argv[1]
libs = [os.path.join(ROOT, "spectrogram_inversion_amd", "libspecinv.so")] + sys.argv[2:]      # "lib.so" or "lib.so:ENV=V,ENV=V"
res = {l: [] for l in libs}
for rnd in range(3):
    for l in libs:
        path, _, extra = l.partition(":")
        env = dict(os.environ, SPECINV_LIB=os.path.abspath(path))
        for kv in filter(None, extra.split(",")):
            env[kv.split("=")[0]] = kv.split("=")[1]
        out = subprocess.run([sys.executable, __file__, "--child", n_fft], env=env, capture_output=True, text=True)
        try:
            res[l].append(float(out.stdout.strip().splitlines()[-1]))
        except Exception:
            print(out.stderr[-500:])
for l in libs:
    print(f"{os.path.basename(l):60s} " + " ".join(f"{v:.4f}" for v in res[l]), flush=True)
