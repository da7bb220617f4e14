"""The multi-GPU host code on a real device: a one-rank `nccl` (= RCCL) process group on cuda:0 drives a real HIP plan
through `run_loop_global` (all-reduce of the evaluation sums, tol > 0), `gather_waveforms` (blocking and async) and the
one-call sharded entry points.  With one rank the collectives are local, but they are RCCL's: communicator set-up,
stream ordering between the plan's kernels and the collective, device tensors in and out.  The two-rank logic (ragged
shards, identical stop decision) is covered on CPU by tests/test_distributed_gloo.py.  Needs an MI355X: `-m gpu`."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

import oracle
from _util import hann, rel_l2

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda", 0)


@pytest.fixture
def nccl_group():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(DEV)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=DEV)
    try:
        yield
    finally:
        dist.destroy_process_group()


def test_rccl_loop_gather_and_sharded_entry_points(nccl_group):
    from spectrogram_inversion_amd.distributed import (ADMM_sharded, RTISI_LA_sharded, gather_waveforms, griffin_lim_sharded,
                                                       run_loop_global)
    from spectrogram_inversion_amd.plan import args_helper, get_plan
    import spectrogram_inversion_amd as si

    rng = np.random.default_rng(21)
    n_fft, hop, frames, batch = 1024, 256, 48, 6
    mag_np = rng.random((batch, n_fft // 2 + 1, frames), dtype=np.float32)
    mag = torch.from_numpy(mag_np).to(DEV)
    w = torch.from_numpy(hann(n_fft))
    kw = dict(hop_length=hop, window=w)

    # (1) run_loop_global on a real plan with a tolerance that fires: same stop iteration and trace as the library's own
    #     loop (the all-reduce over one rank is the identity), waveform equal bit for bit
    plan = get_plan(args_helper(mag, **kw), batch, frames, torch.float32, DEV)
    plan.gla_init(None, mag, 0.99)
    done_ref, evals_ref = plan.run(400, 5, 1e-3, "sc")
    x_ref = plan.wave()
    assert 5 < done_ref < 400
    plan.gla_init(None, mag, 0.99)
    done, evals = run_loop_global(plan, 400, eva_iter=5, tol=1e-3, metric="sc")
    from spectrogram_inversion_amd import distributed as D
    assert D.LAST_LOOP["device_sums"] is True          # the sums were reduced on the device: one read per evaluation
    assert done == done_ref and len(evals) == len(evals_ref)
    np.testing.assert_allclose(np.array(evals), np.array(evals_ref), rtol=1e-12)
    x = plan.wave()
    assert torch.equal(x, x_ref)

    # (2) gather: blocking and async (several in flight), results are the rank's own block
    g = gather_waveforms(x, dst=0)
    assert g.shape == x.shape and torch.equal(g, x)
    hs = [gather_waveforms(x + k, dst=0, sizes=[batch], async_op=True) for k in range(3)]
    for k, h in enumerate(hs):
        assert torch.equal(h.result(), x + k)

    # (3) the one-call entry points against the oracle
    y, done, evals = griffin_lim_sharded(mag, max_iter=12, tol=0.0, alpha=0.3, verbose=False, eva_iter=4, return_info=True, **kw)
    ref = oracle.griffin_lim(mag_np, max_iter=12, alpha=0.3, tol=0, hop_length=hop, window=hann(n_fft))
    assert done == 12 and len(evals) == 3 and rel_l2(y.cpu().numpy(), ref) < 1e-4
    y2 = griffin_lim_sharded(mag, max_iter=12, tol=1e-30, alpha=0.3, verbose=False, eva_iter=4, **kw)     # coupled branch
    assert torch.equal(y2, y)
    z = ADMM_sharded(mag, max_iter=3, tol=0.0, rho=0.5, verbose=False, **kw)
    ref = oracle.admm(mag_np, max_iter=3, rho=0.5, tol=0, hop_length=hop, window=hann(n_fft))
    assert rel_l2(z.cpu().numpy(), ref) < 1e-4
    r = RTISI_LA_sharded(mag, look_ahead=2, asymmetric_window=True, max_iter=3, alpha=0.5, **kw)
    r1 = si.RTISI_LA(mag, look_ahead=2, asymmetric_window=True, max_iter=3, alpha=0.5, verbose=False, **kw)
    assert torch.equal(r, r1)
    dist.barrier()
