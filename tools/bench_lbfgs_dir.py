#!/usr/bin/env python3
"""L-BFGS direction on the device for a memory of m pairs of C5-sized vectors (16 x 523 776 floats): the two-loop
recursion with one dot / axpy per pair (specinv_lbfgs_direction) against the Gram form (one multi-dot pass + one linear
combination: specinv_vec_multi_dot / specinv_vec_lincomb).  Dev tool."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from spectrogram_inversion_amd.lbfgs import HipVecOps

dev = torch.device("cuda", 0)
n = 16 * 523776
ops = HipVecOps(torch.float32, dev)
g = torch.randn(n, device=dev)
for m in (5, 10, 20, 50, 100):
    ss = [torch.randn(n, device=dev) for _ in range(m)]
    ys = [torch.randn(n, device=dev) for _ in range(m)]
    rho = [1.0] * m

    def old():
        return ops.direction(g, ss, ys, rho, 1.0)

    def gram():
        ops.multi_dot(g, ss + ys)
        return ops.lincomb([g] + ys + ss, [1.0] * (2 * m + 1))

    res = []
    for fn in (old, gram):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 5 * 1e3)
    print(f"m={m:3d}: two-loop {res[0]:7.3f} ms   gram {res[1]:7.3f} ms   ({res[0] / res[1]:.2f}x)", flush=True)
    del ss, ys
