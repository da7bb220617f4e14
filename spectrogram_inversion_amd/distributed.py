"""Multi-GPU driver: batches of independent spectrograms shard across ranks (one process
per GPU, `torch.distributed` with the `nccl` backend = RCCL over xGMI) and the waveforms
are gathered to one rank at the end.

The reference has no distributed layer; every batch item is independent
in griffin_lim / ADMM / RTISI_LA (torch_specinv/methods.py:237-250, :458-483, :363-404), so the
data path needs no collective.  Two couplings are handled:

  * the final gather: each rank sends its (B_local, L) block straight to the root
    (`dist.gather` lowers to grouped send/recv, so the 7 xGMI links into the root carry one
    peer's payload each, concurrently - not a ring);
  * `_training_loop`'s metric / early stop use whole-batch sums (methods.py:181-190): when
    `tol > 0` or a progress callback is wanted, the three per-rank sums are all-reduced at
    every evaluation so that all ranks take the same decision.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import _lib
from .metrics import _from_sums


def shard_bounds(n_items: int, world: int, rank: int):
    """Contiguous, balanced batch slices: the first (n_items % world) ranks get one extra."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class PendingGather:
    """Handle of a gather that was started without waiting (`gather_waveforms(..., async_op=True)`): the
    collective runs on RCCL's own stream while the caller enqueues the next inversion; `result()` waits."""

    def __init__(self, work, out, keep):
        self._work, self._out, self._keep = work, out, keep

    def result(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
        self._keep = None
        return self._out


def gather_waveforms(x_local: torch.Tensor, dst: int = 0, group=None, sizes=None, async_op: bool = False):
    """Gather (B_r, L) blocks of possibly different B_r to `dst`.  Returns the concatenated
    (sum B_r, L) tensor on `dst`, None elsewhere.  `sizes` (the per-rank batch sizes) may be
    passed when known, e.g. equal shards, to skip the size exchange.  With `async_op=True` (equal
    shards only) a `PendingGather` is returned instead and the transfer overlaps later work."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return PendingGather(None, x_local, None) if async_op else x_local
    if sizes is None:
        # batch sizes of all ranks: one small tensor all-gather on the data's own device / backend
        mine = torch.tensor([x_local.shape[0]], dtype=torch.int64, device=x_local.device)
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine, group=group)
        sizes = [int(t.item()) for t in every]
    if len(set(sizes)) == 1:
        # receive straight into the slices of the final (world*B_r, L) tensor: no concatenation pass
        big = x_local.new_empty((world * sizes[0],) + tuple(x_local.shape[1:])) if rank == dst else None
        out = list(big.split(sizes[0], 0)) if rank == dst else None
        src = x_local.contiguous()
        if async_op:
            return PendingGather(dist.gather(src, out, dst=dst, group=group, async_op=True), big, src)
        dist.gather(src, out, dst=dst, group=group)
        return big
    assert not async_op, "async gather needs equal shards"
    # ragged batch: point-to-point to the root
    if rank == dst:
        parts = []
        for r in range(world):
            if r == dst:
                parts.append(x_local)
            else:
                buf = x_local.new_empty((sizes[r],) + tuple(x_local.shape[1:]))
                dist.recv(buf, src=r, group=group)
                parts.append(buf)
        return torch.cat(parts, 0)
    dist.send(x_local.contiguous(), dst=dst, group=group)
    return None


def run_loop_global(plan, max_iter, eva_iter=10, tol=0.0, metric="sc", callback=None, group=None):
    """`_training_loop` (methods.py:153-190) with whole-batch semantics across ranks: every
    evaluation all-reduces (sum) the rank-local sums before the metric / stop rule."""
    assert eva_iter > 0 and max_iter > 0 and tol >= 0
    name = metric.upper()
    assert name in _lib.METRICS
    init_loss = None
    previous = None
    done = 0
    evals = []
    while done < max_iter:
        until = eva_iter - (done % eva_iter)
        if done + until > max_iter:
            plan.iterate(max_iter - done)
            done = max_iter
            break
        s = torch.tensor(plan.iterate(until, eval_last=True), dtype=torch.float64, device=plan.device)
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
        s = s.tolist()
        done += until
        m, loss = _from_sums(name, s), s[0] / s[3]
        evals.append((done - 1, m, loss))
        if callback is not None and callback(done - 1, m, loss):
            break
        if not init_loss:
            init_loss = loss
        elif (previous - loss) / init_loss < tol and previous > loss:
            break
        previous = loss
    return done, evals
