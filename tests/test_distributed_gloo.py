"""The N > 1 path on CPU: two gloo processes shard a batch through the one-call entry points
(`griffin_lim_sharded`, `ADMM_sharded`), run the (oracle-backed) per-rank inversion, all-reduce the evaluation
sums and gather the waveforms - the same host code the RCCL runs use (`tests/test_gpu_nccl.py` drives it with the
HIP plan on a real device)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OraclePlan:
    """Stands in for spectrogram_inversion_amd.plan.Plan on CPU (test-only): the same `gla_init` / `admm_init` /
    `iterate` / `run` / `wave` surface, stepping the oracle's Griffin-Lim / ADMM state; iterate() returns the same
    four sums the HIP plan returns."""

    def __init__(self, mag_like, stft_kwargs):
        import oracle
        self.o = oracle
        self.device = torch.device("cpu")
        self.kw = {k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in stft_kwargs.items()}
        self.a = oracle.args_helper(mag_like.shape[1], np.float32, **self.kw)

    def _init(self, init_spec, mag):
        o = self.o
        if init_spec is not None:
            c = init_spec.numpy()
            self.target = np.abs(c)
        else:
            self.target = mag.numpy()
            c = o.phase_init(self.target, **self.kw)
        self.x, self.env = o.istft(c, self.a)
        return c

    def gla_init(self, init_spec, mag, alpha):
        self.pre = self._init(init_spec, mag)
        self.lr = np.float32(alpha / (1 + alpha))
        self.method = "gla"

    def admm_init(self, init_spec, mag, rho):
        c = self._init(init_spec, mag)
        self.X, self.Y, self.U = c, c.copy(), np.zeros_like(c)
        self.rho = np.float32(rho)
        self.method = "admm"

    def iterate(self, n, eval_last=False):
        o = self.o
        out = None
        for _ in range(n):
            new = o.stft(self.x.astype(np.float32), self.a)
            out = np.abs(new)
            if self.method == "gla":
                new = new - self.pre * self.lr
                self.pre = new
                new = new * self.target / (np.abs(new) + np.float32(1e-16))
            else:
                z = (self.rho * self.Y + new) / (np.float32(1) + self.rho)
                u = self.U + self.X - z
                x_ = z - u
                x_ = x_ * self.target / (np.abs(x_) + np.float32(1e-16))
                self.X, self.U, self.Y = x_, u, x_ + u
                new = self.Y
            self.x, _ = o.istft(new, self.a, envelope=self.env)
        if eval_last:
            d = out.astype(np.float64) - self.target
            return [float((d * d).sum()), float((out.astype(np.float64) ** 2).sum()),
                    float((self.target.astype(np.float64) ** 2).sum()), float(out.size)]
        return None

    def run(self, max_iter, eva_iter=10, tol=0.0, metric="sc", callback=None):
        """Rank-local `_training_loop` (what `Plan.run` does inside the library); only reached with tol == 0."""
        from spectrogram_inversion_amd.metrics import _from_sums
        assert tol == 0.0
        done, evals = 0, []
        while done < max_iter:
            until = eva_iter - (done % eva_iter)
            if done + until > max_iter:
                self.iterate(max_iter - done)
                done = max_iter
                break
            s = self.iterate(until, eval_last=True)
            done += until
            evals.append((done - 1, _from_sums(metric.upper(), s), s[0] / s[3]))
        return done, evals

    def wave(self):
        return torch.from_numpy(self.x.astype(np.float32))


def _factory(local, stft_kwargs):
    return OraclePlan(local, stft_kwargs)


def _hann(n):
    return (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n) / n)).astype(np.float32)


def _init(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _worker(rank, world, port, tmp):
    _init(rank, world, port)
    from spectrogram_inversion_amd.distributed import ADMM_sharded, gather_waveforms, griffin_lim_sharded
    rng = np.random.default_rng(5)
    n_items, hop = 5, 64                                    # 5 items over 2 ranks: ragged shards (3 + 2)
    mag = torch.from_numpy(rng.random((n_items, 129, 12), dtype=np.float32))
    kw = dict(hop_length=hop, window=torch.from_numpy(_hann(256)))
    out = {}
    # (1) fixed iteration count, progress watched on rank 0 only: whole-batch sums by all-reduce
    x, done, evals = griffin_lim_sharded(mag, max_iter=6, tol=0.0, alpha=0.3, verbose=False, eva_iter=3, return_info=True,
                                         _plan_factory=_factory, **kw)
    # the no-coupling branch returns each rank's LOCAL evaluation sums; the coupled one is asked for next
    x2, done2, evals2 = griffin_lim_sharded(mag, max_iter=6, tol=1e-30, alpha=0.3, verbose=False, eva_iter=3,
                                            return_info=True, _plan_factory=_factory, **kw)
    # (2) ADMM with a tolerance that fires: both ranks must stop at the same iteration (methods.py:186-190)
    y, done_a, evals_a = ADMM_sharded(mag, max_iter=60, tol=2e-2, rho=1.0, verbose=False, eva_iter=2, return_info=True,
                                      _plan_factory=_factory, **kw)
    both = [None, None]
    dist.all_gather_object(both, (done_a, len(evals_a)))
    assert both[0] == both[1], both
    if rank == 0:
        np.savez(os.path.join(tmp, "out.npz"), x=x.numpy(), x2=x2.numpy(), evals2=np.array(evals2), done=done, done2=done2,
                 y=y.numpy(), done_a=done_a, evals_a=np.array(evals_a))
    else:
        assert x is None and x2 is None and y is None
    # equal shards take the dist.gather branch
    z = gather_waveforms(torch.full((2, 7), float(rank)), dst=0)
    if rank == 0:
        assert z.shape == (4, 7) and z[:2].eq(0).all() and z[2:].eq(1).all()
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
    win = _hann(256)
    trace = []
    ref = oracle.griffin_lim(mag, max_iter=6, alpha=0.3, tol=0, eva_iter=3, hop_length=64, window=win, trace=trace)
    assert int(got["done"]) == 6 and int(got["done2"]) == 6
    np.testing.assert_allclose(got["x"], ref, rtol=1e-5, atol=1e-6)          # gathered in batch order
    np.testing.assert_allclose(got["x2"], ref, rtol=1e-5, atol=1e-6)
    # whole-batch metric / loss (methods.py:181-182) from all-reduced per-rank sums
    want = np.array([[i, m, l] for i, m, l in trace])
    np.testing.assert_allclose(got["evals2"], want, rtol=1e-6)
    # ADMM, tolerance fires: the sharded run stops where the single-process whole-batch run stops
    trace_a = []
    ref_a, st = oracle.admm(mag, max_iter=60, rho=1.0, tol=2e-2, eva_iter=2, hop_length=64, window=win, trace=trace_a,
                            return_state=True)
    assert 2 < st["iters"] < 60, st["iters"]                                  # (the rule really fired)
    assert int(got["done_a"]) == st["iters"]
    np.testing.assert_allclose(got["y"], ref_a, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(got["evals_a"], np.array([[i, m, l] for i, m, l in trace_a]), rtol=1e-6)


def _async_worker(rank, world, port):
    _init(rank, world, port)
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


def _subgroup_worker(rank, world, port):
    _init(rank, world, port)
    from spectrogram_inversion_amd.distributed import gather_waveforms, griffin_lim_sharded
    g = dist.new_group([1, 2])                          # a group that does not start at global rank 0
    if rank in (1, 2):
        gr = dist.get_rank(g)                           # 0 / 1 inside the group
        # `dst` is a group rank: group rank 0 = global rank 1 receives (equal shards: dist.gather; ragged: send / recv)
        z = gather_waveforms(torch.full((2, 3), float(rank)), dst=0, group=g)
        r = gather_waveforms(torch.full((1 + gr, 3), float(rank)), dst=1, group=g)
        if gr == 0:
            assert z.shape == (4, 3) and z[:2].eq(1).all() and z[2:].eq(2).all()
            assert r is None
        else:
            assert z is None
            assert r.shape == (3, 3) and r[:1].eq(1).all() and r[1:].eq(2).all()
        # the one-call entry point on the sub-group: shards by group rank, result on the group's `dst`
        rng = np.random.default_rng(5)
        mag = torch.from_numpy(rng.random((3, 129, 12), dtype=np.float32))
        x, done, _ = griffin_lim_sharded(mag, max_iter=4, tol=1e-30, alpha=0.3, verbose=False, eva_iter=2, dst=1, group=g,
                                         return_info=True, _plan_factory=_factory, hop_length=64,
                                         window=torch.from_numpy(_hann(256)))
        assert done == 4 and ((x is not None and x.shape[0] == 3) if gr == 1 else x is None)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sub_group_that_does_not_start_at_rank_zero():
    """`dst` and the rank it is compared with are both ranks of `group` (the collectives get `group_dst` / `group_src`)."""
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_subgroup_worker, args=(3, port), nprocs=3, join=True)
