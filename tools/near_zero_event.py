#!/usr/bin/env python3
"""Where do two float32 Griffin-Lim runs of the g2 fixture (alpha 0.3) part ways?  Runs the float32 frame kernel (default arithmetic)
and the float64 generic kernels from the same starting spectrum, one iteration at a time, and reports the first iteration at which
a hop-sized segment of the waveforms differs by more than 1e-4 of the RMS segment energy, the frame that segment belongs to, and
the bin of that frame's pre-projection spectrum S (methods.py:243) that is closest to zero relative to its target magnitude - the
projection S m / |S| is discontinuous at S = 0, so a bin passing within rounding distance of it lands on either side.

    python tools/near_zero_event.py [alpha] [path: frame|fused|fused_prespec]      # on the GPU box
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from _util import load_golden, segment_errors
from spectrogram_inversion_amd.plan import Plan, args_helper

alpha = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
path = sys.argv[2] if len(sys.argv) > 2 else "frame"
if path != "frame":
    os.environ["SPECINV_SMALL_FRAMES"] = "0"
dev = torch.device("cuda", 0)
g = load_golden("g2_gla")
hop, w = int(g["hop"]), torch.from_numpy(g["window"])
init = torch.from_numpy(g["init"]).to(dev)
mag = init.abs()
p32 = Plan(args_helper(init, hop_length=hop, window=w), init.shape[0], init.shape[2], torch.float32, dev)
p32.keep_state(True)                                   # pre_spec readable (the fused shapes then iterate on pre_spec itself)
p32.gla_init(init, None, alpha)
i64 = init.to(torch.complex128)
p64 = Plan(args_helper(i64, hop_length=hop, window=w.double()), init.shape[0], init.shape[2], torch.float64, dev)
p64.gla_init(i64, None, alpha)
print(f"g2, alpha {alpha}, {p32.launch_geometry['kernel']} (float32) vs generic float64; hop {hop}, {init.shape[2]} frames x {init.shape[1]} bins")
found = False
for it in range(1, 101):
    p32.iterate(1)
    p64.iterate(1)
    x32, x64 = p32.wave().double().cpu().numpy(), p64.wave().cpu().numpy()
    seg = segment_errors(x32, x64, hop).reshape(x32.shape[0], -1)
    worst = np.unravel_index(np.argmax(seg), seg.shape)
    if it in (1, 10, 50, 90) or seg.max() > 1e-4 or it == 100:
        print(f"  iteration {it:3d}: max segment error {seg.max():.2e} (item {worst[0]}, segment {worst[1]}), median {np.median(seg):.2e}")
    if seg.max() > 1e-4 and not found:
        found = True
        s32, s64 = p32.state_spec(0), p64.state_spec(0)          # S of this iteration (pre_spec), (B, F, T)
        b = int(worst[0])
        frames = range(max(0, worst[1] - 1), min(init.shape[2], worst[1] + 5))   # frames that cover the segment (centre padding: +2)
        print(f"  -> first event at iteration {it}; bins of the frames around segment {worst[1]} with the smallest |S| / m:")
        for t in frames:
            r = (s64[b, :, t].abs() / (mag[b, :, t].double() + 1e-30)).cpu().numpy()
            k = int(np.argmin(r))
            a32, a64 = complex(s32[b, k, t].cpu()), complex(s64[b, k, t].cpu())
            print(f"     frame {t:3d} bin {k:4d}: |S64| / m = {r[k]:.2e}   S32 = {a32.real:+.3e}{a32.imag:+.3e}j   S64 = {a64.real:+.3e}{a64.imag:+.3e}j"
                  f"   phase difference {abs(np.angle(a32 * np.conj(a64))):.2f} rad   (typical |S| / m in this frame: {np.median(r):.2f})")
if not found:
    print("  no segment ever differs by more than 1e-4: the two runs stay together")
