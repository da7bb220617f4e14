"""World size 2 on the real device: two freshly spawned processes, both on cuda:0, drive REAL HIP plans through the sharded
entry points (`griffin_lim_sharded`, `ADMM_sharded`, `RTISI_LA_sharded` -> shard -> Plan -> `run_loop_global` -> gather).
One GPU cannot host a two-rank RCCL communicator, so the group is `gloo`: the three evaluation sums are all-reduced on host
copies and the waveforms are gathered through host memory (`distributed._host_staged`) - everything else (shard bounds,
plans, kernels, the agreed stop decision, batch order of the gather) is what an 8-GPU RCCL run executes.  xGMI itself stays
unmeasured here.  Needs an MI355X: `-m gpu`.  (reference coupling: torch_specinv/methods.py:181-190.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch

from _util import hann

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_FFT, HOP, FRAMES, ITEMS = 1024, 256, 40, 5              # 5 items over 2 ranks: ragged shards (3 + 2)
GLA = dict(max_iter=300, tol=1e-3, alpha=0.99, eva_iter=5)
ADMM = dict(max_iter=60, tol=2e-2, rho=1.0, eva_iter=2)
RTISI = dict(look_ahead=2, asymmetric_window=True, max_iter=3, alpha=0.5)


def _inputs():
    rng = np.random.default_rng(52)
    mag = rng.random((ITEMS, N_FFT // 2 + 1, FRAMES), dtype=np.float32)
    return mag, dict(hop_length=HOP, window=torch.from_numpy(hann(N_FFT)))


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from spectrogram_inversion_amd.distributed import ADMM_sharded, RTISI_LA_sharded, griffin_lim_sharded
    mag_np, kw = _inputs()
    mag = torch.from_numpy(mag_np).to(dev)
    x, done, evals = griffin_lim_sharded(mag, verbose=False, return_info=True, **GLA, **kw)
    y, done_a, evals_a = ADMM_sharded(mag, verbose=False, return_info=True, **ADMM, **kw)
    r = RTISI_LA_sharded(mag, **RTISI, **kw)
    # the no-coupling branch (tol == 0, nobody watches): the library's own loop per rank, then the gather
    f = griffin_lim_sharded(mag, max_iter=7, tol=0.0, alpha=0.3, verbose=False, eva_iter=3, **kw)
    with open("/proc/self/maps") as fh:
        assert "libspecinv.so" in fh.read(), "native HIP library not loaded"
    both = [None] * world
    dist.all_gather_object(both, (done, len(evals), done_a, len(evals_a)))
    assert both[0] == both[1], both                         # every rank stopped at the same iteration
    if rank == 0:
        assert all(t.device == dev for t in (x, y, r, f))
        np.savez(os.path.join(tmp, "out.npz"), x=x.cpu().numpy(), y=y.cpu().numpy(), r=r.cpu().numpy(), f=f.cpu().numpy(),
                 done=done, done_a=done_a, evals=np.array(evals), evals_a=np.array(evals_a))
    else:
        assert x is None and y is None and r is None and f is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_processes_on_one_device_match_the_whole_batch_run(tmp_path):
    import torch.multiprocessing as mp
    from spectrogram_inversion_amd.plan import args_helper, get_plan
    import spectrogram_inversion_amd as si
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    # fresh interpreters (spawn): nothing of this process's HIP state is inherited, nothing is re-exec'ed
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(tmp_path, "out.npz"))

    dev = torch.device("cuda", 0)
    mag_np, kw = _inputs()
    mag = torch.from_numpy(mag_np).to(dev)
    plan = get_plan(args_helper(mag, **kw), ITEMS, FRAMES, torch.float32, dev)
    # Griffin-Lim, the tolerance fires: same stop iteration, same trace (sums added in another order: 1e-12), and - the
    # items being independent and the frame kernel's sums having a fixed order - the same waveforms bit for bit
    plan.gla_init(None, mag, GLA["alpha"])
    done, evals = plan.run(GLA["max_iter"], GLA["eva_iter"], GLA["tol"], "sc")
    assert GLA["eva_iter"] < done < GLA["max_iter"], done   # (the rule really fired)
    assert int(got["done"]) == done
    np.testing.assert_allclose(got["evals"], np.array(evals), rtol=1e-10)
    assert np.array_equal(got["x"], plan.wave().cpu().numpy())
    plan.admm_init(None, mag, ADMM["rho"])
    done_a, evals_a = plan.run(ADMM["max_iter"], ADMM["eva_iter"], ADMM["tol"], "sc")
    assert ADMM["eva_iter"] < done_a < ADMM["max_iter"], done_a
    assert int(got["done_a"]) == done_a
    np.testing.assert_allclose(got["evals_a"], np.array(evals_a), rtol=1e-10)
    assert np.array_equal(got["y"], plan.wave().cpu().numpy())
    r = si.RTISI_LA(mag, verbose=False, **RTISI, **kw)
    assert np.array_equal(got["r"], r.cpu().numpy())
    f = si.griffin_lim(mag, max_iter=7, tol=0.0, alpha=0.3, verbose=False, eva_iter=3, **kw)
    assert np.array_equal(got["f"], f.cpu().numpy())


@pytest.mark.timeout(900)
@pytest.mark.parametrize("workload,extra", [("C1", []), ("C1", ["--no-overlap-gather", "--even-chunks"]), ("C5", ["--outer", "2"])])
def test_bench_world_size_two_rehearsal(workload, extra):
    """bench.py's N > 1 code path, started as a user would - plain `python3 bench.py --gpus 2`: with WORLD_SIZE unset the script
    spawns its own ranks under torch.distributed.run (a child process, before anything has touched the GPU) - with both ranks on
    cuda:0 and a gloo group (`SPECINV_BENCH_BACKEND=gloo`: a device cannot host two RCCL ranks): the shards, the barrier +
    max-over-ranks timing, the gather to rank 0, the per-rank diagnostics and the one JSON line with whole-job units.  The rate it
    prints is not a measurement; that the 8-GPU launch cannot die on a code path nobody has run is the point."""
    import json
    import subprocess
    env = dict(os.environ, SPECINV_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--workload", workload, "--no-cpu-baseline"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=800)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]              # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    mg = d["multi_gpu"]
    assert mg["ranks_seen"] == 2 and len(mg["per_rank_ms_per_step"]) == 2 and all(v > 0 for v in mg["per_rank_ms_per_step"])
    if workload == "C1":
        assert d["config"]["global_batch"] == 2 and d["check"]["ok"]
        assert mg["gather_alone_ms_rank0"] > 0 and mg["gather_bytes_per_rank"] == 4 * 130816
        assert mg["gather"].startswith("blocking" if "--no-overlap-gather" in extra else "overlapped")
    else:
        assert d["config"]["parallelism"].startswith("replicas x2")
