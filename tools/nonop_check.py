#!/usr/bin/env python3
"""The nop-stripped variant of tools/strip_nops_variant.py against the shipped library: C2 waveforms after 40 iterations (early and
late launches) bit for bit, and the time per late launch / per step, arms interleaved."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VAR = os.path.join(ROOT, "spectrogram_inversion_amd", "variants", "libspecinv_nonop.so")
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch, hashlib
    from spectrogram_inversion_amd.plan import Plan, args_helper
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(1234)
    mag = torch.from_numpy(rng.random((64, 1025, 1024), dtype=np.float32)).to(dev)
    w = torch.from_numpy((0.5 - 0.5 * np.cos(2 * np.pi * np.arange(2048) / 2048)).astype(np.float32))
    p = Plan(args_helper(mag, hop_length=512, window=w), 64, 1024, torch.float32, dev)
    p.gla_init(None, mag, 0.3)
    p.iterate(40)
    h = hashlib.sha1(p.wave().cpu().numpy().tobytes()).hexdigest()[:16]
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); p.iterate(50); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 50)
    steps = []
    for _ in range(4):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); p.gla_init(None, mag, 0.3); p.run(100, 10, 0.0, "sc"); p.wave(); e1.record(); torch.cuda.synchronize()
        steps.append(e0.elapsed_time(e1))
    print(json.dumps({"sha": h, "late_ms": best, "step_ms": min(steps)}))
    sys.exit(0)
for rnd in range(3):
    for tag, lib in (("shipped", None), ("nonop", VAR)):
        env = dict(os.environ)
        if lib:
            env["SPECINV_LIB"] = lib
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
        print(tag, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:], flush=True)
