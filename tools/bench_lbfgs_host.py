#!/usr/bin/env python3
"""Host cost (enqueue time, microseconds) of the calls one L-BFGS inner iteration makes at the C5 shape: what the GPU waits
for between the read-back of an iteration and the first kernel of the next."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import spectrogram_inversion_amd as si
from spectrogram_inversion_amd.lbfgs import HipVecOps
dev = torch.device("cuda", 0)
B, T, n_fft, hop = 16, 1024, 2048, 512
w = torch.from_numpy((0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)).astype(np.float32))
fb = torch.from_numpy(si.mel_filterbank(22050, n_fft, 80)).to(dev)
tr = si.LogMelSTFT(fb, n_fft, hop_length=hop, window=w)
xs = 0.1 * torch.randn(B, (T - 1) * hop, device=dev)
target = tr(xs)
x = 1e-3 * torch.randn_like(xs)
_, fg = tr.bind(x, target)
ops = HipVecOps(torch.float32, dev)
board = ops.board(32)
g = fg.dev(x, board.data_ptr())
gp, d = torch.randn_like(g), torch.randn_like(g)

def timeit(name, fn, n=200, sync_each=False):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
        if sync_each:
            torch.cuda.synchronize()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:34s} enqueue {1e6 * (t1 - t0) / n:7.1f} us   incl. drain {1e6 * (t2 - t0) / n:7.1f} us", flush=True)

timeit("fg.dev (objective + epilogue)", lambda: fg.dev(x, board.data_ptr()))
timeit("pair_stats_into", lambda: ops.pair_stats_into(g, gp, d, 1.0, board, 1))
timeit("lincomb_step([g], [-1])", lambda: ops.lincomb_step([g], [-1.0], 1e-9, x))
timeit("read (idle device)", lambda: ops.read(board, 9))
timeit("torch.empty_like", lambda: torch.empty_like(g))
timeit("plan._sync_stream", lambda: ops.plan._sync_stream())
def one_iteration():
    dd = ops.lincomb_step([g], [-1.0], 1e-9, x)
    gg = fg.dev(x, board.data_ptr())
    ops.pair_stats_into(gg, gp, dd, 1.0, board, 1)
    return ops.read(board, 9)
timeit("one iteration (4 calls + read)", one_iteration)
