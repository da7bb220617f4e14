"""The N > 1 path on CPU: two gloo processes shard a batch, run the (oracle-backed) per-rank inversion,
all-reduce the evaluation sums and gather the waveforms - the same host code the RCCL runs use."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OraclePlan:
    """Stands in for spectrogram_inversion_amd.plan.Plan on CPU (test-only): iterate() steps the oracle's
    Griffin-Lim state and returns the same four sums the HIP plan returns."""

    def __init__(self, mag, hop, window, alpha):
        import oracle
        self.o = oracle
        self.device = torch.device("cpu")
        self.a = oracle.args_helper(mag.shape[1], np.float32, hop_length=hop, window=window)
        self.target = mag
        self.pre = oracle.phase_init(mag, hop_length=hop, window=window)
        self.x, self.env = oracle.istft(self.pre, self.a)
        self.lr = np.float32(alpha / (1 + alpha))

    def iterate(self, n, eval_last=False):
        o = self.o
        out = None
        for _ in range(n):
            new = o.stft(self.x.astype(np.float32), self.a)
            out = np.abs(new)
            new = new - self.pre * self.lr
            self.pre = new
            new = new * self.target / (np.abs(new) + np.float32(1e-16))
            self.x, _ = o.istft(new, self.a, envelope=self.env)
        if eval_last:
            d = out.astype(np.float64) - self.target
            return [float((d * d).sum()), float((out.astype(np.float64) ** 2).sum()),
                    float((self.target.astype(np.float64) ** 2).sum()), float(out.size)]
        return None

    def wave(self):
        return torch.from_numpy(self.x.astype(np.float32))


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from spectrogram_inversion_amd.distributed import gather_waveforms, run_loop_global, shard_bounds
    rng = np.random.default_rng(5)
    n_items, hop = 5, 64                                    # 5 items over 2 ranks: ragged shards (3 + 2)
    mag = rng.random((n_items, 129, 12), dtype=np.float32)
    win = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(256) / 256)).astype(np.float32)
    lo, hi = shard_bounds(n_items, world, rank)
    plan = OraclePlan(mag[lo:hi], hop, win, 0.3)
    done, evals = run_loop_global(plan, 6, eva_iter=3, tol=0.0, metric="sc")
    x = gather_waveforms(plan.wave(), dst=0)
    if rank == 0:
        np.savez(os.path.join(tmp, "out.npz"), x=x.numpy(), evals=np.array(evals), done=done)
    else:
        assert x is None
    # equal shards take the dist.gather branch
    y = gather_waveforms(torch.full((2, 7), float(rank)), dst=0)
    if rank == 0:
        assert y.shape == (4, 7) and y[:2].eq(0).all() and y[2:].eq(1).all()
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds():
    from spectrogram_inversion_amd.distributed import shard_bounds
    for n, w in ((64, 8), (5, 2), (7, 3), (2, 4)):
        spans = [shard_bounds(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_two_rank_sharded_inversion_matches_single_process(tmp_path):
    import oracle
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(tmp_path, "out.npz"))
    rng = np.random.default_rng(5)
    mag = rng.random((5, 129, 12), dtype=np.float32)
    win = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(256) / 256)).astype(np.float32)
    trace = []
    ref = oracle.griffin_lim(mag, max_iter=6, alpha=0.3, tol=0, eva_iter=3, hop_length=64, window=win, trace=trace)
    assert int(got["done"]) == 6
    np.testing.assert_allclose(got["x"], ref, rtol=1e-5, atol=1e-6)          # gathered in batch order
    # whole-batch metric / loss (methods.py:181-182) from all-reduced per-rank sums
    want = np.array([[i, m, l] for i, m, l in trace])
    np.testing.assert_allclose(got["evals"], want, rtol=1e-6)


def _async_worker(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from spectrogram_inversion_amd.distributed import gather_waveforms
    handles = [gather_waveforms(torch.full((3, 5), float(10 * k + rank)), dst=0, sizes=[3] * world, async_op=True)
               for k in range(3)]                       # several gathers in flight, completed later in order
    for k, h in enumerate(handles):
        y = h.result()
        if rank == 0:
            assert y.shape == (6, 5) and y[:3].eq(10 * k).all() and y[3:].eq(10 * k + 1).all()
        else:
            assert y is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_async_gather_overlaps_and_completes():
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_async_worker, args=(2, port), nprocs=2, join=True)
