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

from tqdm import tqdm

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


def _host_staged(group, device) -> bool:
    """True when the group's backend cannot move device tensors itself (gloo with a GPU tensor): the payload is
    then staged through host memory.  RCCL ("nccl") groups exchange device buffers directly over xGMI."""
    return device.type == "cuda" and str(dist.get_backend(group)).lower() == "gloo"


def gather_waveforms(x_local: torch.Tensor, dst: int = 0, group=None, sizes=None, async_op: bool = False):
    """Gather (B_r, L) blocks of possibly different B_r to `dst`.  Returns the concatenated
    (sum B_r, L) tensor on `dst`, None elsewhere.  `sizes` (the per-rank batch sizes) may be
    passed when known, e.g. equal shards, to skip the size exchange.  With `async_op=True` (equal
    shards only) a `PendingGather` is returned instead and the transfer overlaps later work.
    `dst` is a rank OF `group` (like the `rank` it is compared with); the collectives get it as `group_dst`."""
    if not dist.is_initialized():
        return PendingGather(None, x_local, None) if async_op else x_local
    if _host_staged(group, x_local.device):
        assert not async_op, "asynchronous gathers need a backend that moves device buffers (nccl)"
        dev = x_local.device
        out = gather_waveforms(x_local.cpu(), dst=dst, group=group, sizes=sizes)
        return None if out is None else out.to(dev)
    # (a one-rank group still goes through the collective: the same code path at every world size)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
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
            return PendingGather(dist.gather(src, out, group_dst=dst, group=group, async_op=True), big, src)
        dist.gather(src, out, group_dst=dst, group=group)
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
                dist.recv(buf, group_src=r, group=group)
                parts.append(buf)
        return torch.cat(parts, 0)
    dist.send(x_local.contiguous(), group_dst=dst, group=group)
    return None


LAST_LOOP = {"device_sums": None}          # diagnostics of the last run_loop_global (tests: which path the evaluations took)


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
        if dist.is_initialized() and hasattr(plan, "iterate_dev") and not _host_staged(group, plan.device):
            # the rank's four sums stay on the device, are all-reduced there (RCCL, the plan's stream order) and read ONCE per
            # evaluation (round 5: a read of the local sums, then a second one of the reduced sums)
            s = torch.empty(4, dtype=torch.float64, device=plan.device)
            plan.iterate_dev(until, s)
            dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
            s = s.tolist()
            LAST_LOOP["device_sums"] = True
        else:
            s = plan.iterate(until, eval_last=True)      # this rank's four sums (host floats)
            LAST_LOOP["device_sums"] = False
            if dist.is_initialized():
                where = torch.device("cpu") if _host_staged(group, plan.device) else plan.device
                s = torch.tensor(s, dtype=torch.float64, device=where)
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


# ---- one-call sharded inversions ---------------------------------------------------------------------------
def _world(group):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def _device_plan(spec_local, stft_kwargs):
    """The HIP plan of this rank's shard on the current device."""
    from .plan import args_helper, get_plan, require_gpu
    args = args_helper(spec_local, **stft_kwargs)
    if args.complex_window:
        raise RuntimeError("complex windows are not supported (torch_specinv/methods.py:127 raises on them too)")
    device = require_gpu(spec_local.device if spec_local.device.type == "cuda" else None)
    rdtype = spec_local.real.dtype if spec_local.is_complex() else spec_local.dtype
    return get_plan(args, spec_local.shape[0], spec_local.shape[2], rdtype, device)


def _sharded(which, spec, coef, max_iter, tol, verbose, eva_iter, metric, dst, group, spec_is_local, plan_factory,
             stft_kwargs):
    assert eva_iter > 0 and max_iter > 0 and tol >= 0
    assert isinstance(metric, str) and metric.upper() in _lib.METRICS
    assert spec.dim() == 3, "sharded inversion takes a (B, F, T) batch"
    from .methods import _widen
    spec, stft_kwargs, narrow = _widen(spec, stft_kwargs)     # float16 / complex32 spectrograms are inverted in float32
    world, rank = _world(group)
    if spec_is_local:
        local = spec
    else:
        assert spec.shape[0] >= world, "fewer items than ranks"
        lo, hi = shard_bounds(spec.shape[0], world, rank)
        local = spec[lo:hi]
    plan = (plan_factory or _device_plan)(local, stft_kwargs)
    if local.device != plan.device:
        local = local.to(plan.device)
    init = getattr(plan, which + "_init")
    if local.is_complex():
        init(local, None, coef)              # warm start, target = |spec| (methods.py:110)
    else:
        init(None, local, coef)              # phase_init on the device (methods.py:106)
    watch = bool(verbose) or tol > 0
    if not (dist.is_available() and dist.is_initialized()) or not watch:
        # nothing couples the ranks before the gather: the library's own loop (evaluations stay on the device
        # when tol == 0 and nobody watches)
        if verbose:
            name = metric.upper()
            with tqdm(total=max_iter) as pbar:
                def cb(_i, m, loss):
                    pbar.set_postfix(**{name: m}, loss=loss)
                    pbar.update(eva_iter)
                    return 0
                done, evals = plan.run(max_iter, eva_iter, tol, metric, callback=cb)
        else:
            done, evals = plan.run(max_iter, eva_iter, tol, metric)
    else:
        # `_training_loop`'s metric and stop rule are whole-batch quantities (methods.py:181-190): every evaluation
        # all-reduces the three sums, every rank takes the same decision at the same iteration.  (`verbose` must be
        # the same on all ranks - the bar itself is only drawn by `dst`.)
        name = metric.upper()
        with tqdm(total=max_iter, disable=not (verbose and rank == dst)) as pbar:
            def cb(_i, m, loss):
                pbar.set_postfix(**{name: m}, loss=loss)
                pbar.update(eva_iter)
                return 0
            done, evals = run_loop_global(plan, max_iter, eva_iter, tol, metric, callback=cb, group=group)
    x = plan.wave()
    if narrow is not None:
        x = x.to(narrow)
    x = gather_waveforms(x, dst=dst, group=group)
    return x, done, evals


def griffin_lim_sharded(spec, max_iter=200, tol=1e-6, alpha=0.99, verbose=True, eva_iter=10, metric="sc", dst=0,
                        group=None, spec_is_local=False, return_info=False, _plan_factory=None, **stft_kwargs):
    r"""`griffin_lim` (torch_specinv/methods.py:193-270) over all ranks of `group`: the (B, F, T) batch is split into
    contiguous shards (`shard_bounds`), every rank inverts its shard on its own GPU, the whole-batch metric / early-stop
    rule of `_training_loop` (:181-190) is kept by all-reducing three sums per evaluation (only when `tol > 0` or
    progress is shown), and one RCCL gather brings the (B, L) waveforms to rank `dst`.  Call it on every rank with the
    same arguments; `spec` is the whole batch (or this rank's shard with `spec_is_local=True`).  Returns the waveforms
    on `dst`, None elsewhere (with `return_info`: also the iterations done and the evaluation trace)."""
    assert alpha >= 0
    x, done, evals = _sharded("gla", spec, alpha, max_iter, tol, verbose, eva_iter, metric, dst, group, spec_is_local,
                              _plan_factory, stft_kwargs)
    return (x, done, evals) if return_info else x


def ADMM_sharded(spec, max_iter=1000, tol=1e-6, rho=0.1, verbose=1, eva_iter=10, metric="sc", dst=0, group=None,
                 spec_is_local=False, return_info=False, _plan_factory=None, **stft_kwargs):
    r"""`ADMM` (torch_specinv/methods.py:415-506) sharded like `griffin_lim_sharded` - BASELINE.json's configs[3]."""
    x, done, evals = _sharded("admm", spec, rho, max_iter, tol, verbose, eva_iter, metric, dst, group, spec_is_local,
                              _plan_factory, stft_kwargs)
    return (x, done, evals) if return_info else x


def RTISI_LA_sharded(spec, look_ahead=-1, asymmetric_window=False, max_iter=25, alpha=0.99, dst=0, group=None,
                     spec_is_local=False, **stft_kwargs):
    r"""`RTISI_LA` (torch_specinv/methods.py:273-412) sharded over the batch: items are independent and there is no
    whole-batch rule, so the only exchange is the final gather."""
    from .methods import RTISI_LA
    assert spec.dim() == 3, "sharded inversion takes a (B, F, T) batch"
    world, rank = _world(group)
    local = spec
    if not spec_is_local:
        assert spec.shape[0] >= world, "fewer items than ranks"     # (an empty shard would leave the others waiting in the gather)
        lo, hi = shard_bounds(spec.shape[0], world, rank)
        local = spec[lo:hi]
    from .plan import require_gpu
    dev = require_gpu(local.device if local.device.type == "cuda" else None)
    x = RTISI_LA(local.to(dev), look_ahead, asymmetric_window, max_iter, alpha, verbose=False, **stft_kwargs)
    if x.dim() == 1:
        x = x.unsqueeze(0)
    return gather_waveforms(x, dst=dst, group=group)
