#!/usr/bin/env python3
"""One seed of tests/test_gpu_random_configs.py on every kernel path (fused / frame / generic) and the float32 oracle,
error against the float64 oracle per hop-block of the last item after 1, 2, 3 iterations: tells an ill-conditioned draw
from a kernel bug (dev tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle
from test_gpu_random_configs import draw
import spectrogram_inversion_amd as si
from spectrogram_inversion_amd.plan import clear_plan_cache
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1291
n_fft, kw, mag, method, coef = draw(seed, wave_level=True)
print(n_fft, {k: (v if k != "window" else None) for k, v in kw.items()}, mag.shape, method, coef)
tkw = dict(kw); tkw["window"] = torch.from_numpy(kw["window"]) if kw["window"] is not None else None
init = oracle.phase_init(mag, **kw)
for iters in (1, 2, 3):
    ref = oracle.griffin_lim(init, max_iter=iters, alpha=coef, tol=0, **kw)
    ref64 = oracle.griffin_lim(init.astype(np.complex128), max_iter=iters, alpha=coef, tol=0, **{**kw, "window": None if kw["window"] is None else kw["window"].astype(np.float64)})
    outs = {}
    for name, env in (("fused", {}), ("semi", {"SPECINV_DISABLE_FUSED": "1"}), ("generic", {"SPECINV_DISABLE_FAST": "1"})):
        for k in ("SPECINV_DISABLE_FUSED", "SPECINV_DISABLE_FAST"):
            os.environ.pop(k, None)
        os.environ.update(env)
        os.environ["SPECINV_SMALL_FRAMES"] = "0"
        clear_plan_cache()
        y = si.griffin_lim(torch.from_numpy(init).cuda(), max_iter=iters, alpha=coef, tol=0, verbose=False, **tkw).cpu().numpy()
        outs[name] = y
    ref = np.asarray(ref).reshape(outs["fused"].shape); ref64 = np.asarray(ref64).reshape(ref.shape)
    hop = kw["hop_length"]
    for name, y in outs.items():
        e = np.abs(y - ref64)
        blocks = e.reshape(e.shape[0], -1, hop).max(-1)
        print(iters, name, "max err vs f64 per hop-block (last item):", np.array2string(blocks[-1], precision=2, max_line_width=200))
    e = np.abs(ref - ref64).reshape(ref.shape[0], -1, hop).max(-1)
    print(iters, "oracle32", np.array2string(e[-1], precision=2, max_line_width=200))
