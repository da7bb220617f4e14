#!/usr/bin/env python3
"""Headline benchmark: Griffin-Lim iterations*frames/s (BASELINE.json metric).

Workload (BASELINE.json configs[1], "C2"): griffin_lim, batch 64 PER GPU (weak scaling),
n_fft=2048, hop=512, n_frames=1024, 100 iterations, alpha=0.3, periodic Hann window,
center/reflect, tol=0, eva_iter=10, metric 'sc'; magnitudes uniform[0,1) from
default_rng(1234 + rank).  One "step" = one complete inversion of the batch:
phase_init + initial ISTFT + 100 fused iterations (+ the RCCL gather of the waveforms to
rank 0 when N > 1).  Inputs are resident in HBM before the timed region; plan creation is
outside it.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 3 --warmup 1

Rank 0 prints ONE JSON line (see the driver contract) with `roofline` (dominant-kernel
achieved HBM GB/s from HIP events on the launch stream) and `cpu_baseline` (the NumPy oracle
timed on the host cores on a bounded sample of the same workload; N=1, rank 0 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

WORKLOADS = {
    # name: (batch per GPU, n_fft, hop, frames, iterations, alpha)
    "C2": (64, 2048, 512, 1024, 100, 0.3),
    "C1": (1, 1024, 256, 512, 50, 0.0),
}
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def hann(n):
    return (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n) / n)).astype(np.float32)


def algorithmic_bytes_per_unit(hop, n_freq, alpha):
    """one frame through one iteration, fp32: x read+write 8*hop, target 4F,
    pre_spec read+write 16F (alpha != 0); 8*hop + 4F when alpha == 0."""
    return 8 * hop + (20 if alpha != 0 else 4) * n_freq


def cpu_baseline(n_fft, hop, frames, alpha, budget_s=15.0):
    """The oracle (a port of the reference's algorithm) on the host cores, bounded sample."""
    import oracle
    from oracle import stftlib
    cores = os.cpu_count() or 1
    stftlib.WORKERS = cores
    b = 8
    rng = np.random.default_rng(99)
    mag = rng.random((b, n_fft // 2 + 1, frames), dtype=np.float32)
    w = hann(n_fft)
    init = oracle.phase_init(mag, hop_length=hop, window=w)
    oracle.griffin_lim(init, max_iter=1, alpha=alpha, tol=0, hop_length=hop, window=w)     # warm caches
    iters = 2
    while True:
        t0 = time.perf_counter()
        oracle.griffin_lim(init, max_iter=iters, alpha=alpha, tol=0, eva_iter=10, hop_length=hop, window=w)
        dt = time.perf_counter() - t0
        if dt >= budget_s / 2 or iters >= 400:
            break
        iters = min(400, max(iters * 2, int(iters * budget_s / max(dt, 1e-3))))
    stftlib.WORKERS = 1
    return {"value": iters * b * frames / dt, "unit": "iterations*frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle.griffin_lim batch={b} n_fft={n_fft} hop={hop} n_frames={frames} "
                      f"{iters} iterations alpha={alpha} ({dt:.1f} s, scipy.fft workers={cores})"}


def load_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the committed PMC summary, if any."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as fh:
            return json.load(fh).get(workload)
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="override the per-GPU batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--generic", action="store_true", help="force the generic (unfused) kernels")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import spectrogram_inversion_amd as si            # noqa: F401
    from spectrogram_inversion_amd.distributed import gather_waveforms
    from spectrogram_inversion_amd.plan import args_helper, get_plan

    batch, n_fft, hop, frames, iters, alpha = WORKLOADS[args.workload]
    if args.batch:
        batch = args.batch
    n_freq = n_fft // 2 + 1
    rng = np.random.default_rng(1234 + rank)
    mag = torch.from_numpy(rng.random((batch, n_freq, frames), dtype=np.float32)).to(dev)
    window = torch.from_numpy(hann(n_fft))
    a = args_helper(mag, hop_length=hop, window=window)
    plan = get_plan(a, batch, frames, torch.float32, dev)
    if args.generic:
        plan.force_generic(True)

    run_events = []
    pending = []

    def step():
        plan.gla_init(None, mag, alpha)                 # phase_init + initial ISTFT
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()                                     # HIP events on the stream the kernels are launched on
        done, _ = plan.run(iters, 10, 0.0, "sc")        # 100 iterations, evaluation every 10
        e1.record()
        run_events.append((e0, e1))
        x = plan.wave()
        if world > 1:
            # RCCL gather of the (B, L) waveforms to rank 0; it runs on RCCL's stream, so the next step's kernels
            # overlap it - every gather is completed (`result()`) inside the timed region
            if pending:
                pending.pop().result()
            pending.append(gather_waveforms(x, dst=0, sizes=[batch] * world, async_op=True))
        return done, x

    def fence():
        out = pending.pop().result() if pending else None
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        return out

    for _ in range(args.warmup):
        step()
    fence()
    run_events.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        done, x = step()
    gathered = fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        x = gathered
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert done == iters
    if rank == 0:
        assert x.shape == (batch * world, plan.length) and bool(torch.isfinite(x).all())

    # dominant kernel: the per-iteration launch.  Average duration over the timed region = HIP-event time of
    # each step's 100-iteration run / 100 (90 plain + 10 evaluating launches, back to back on one stream).
    launch_ms = sum(a.elapsed_time(b) for a, b in run_events) / (len(run_events) * iters)
    unit_bytes = algorithmic_bytes_per_unit(hop, n_freq, alpha)
    launch_bytes = unit_bytes * batch * frames
    achieved = launch_bytes / (launch_ms * 1e-3) / 1e9

    if rank == 0:
        units = args.steps * iters * batch * world * frames
        path = plan.path                       # "fused" | "frame" | "generic"
        fused = path == "fused"
        out = {
            "metric": "Griffin-Lim iterations*frames/sec at n_fft=2048 hop=512" if args.workload == "C2"
                      else f"Griffin-Lim iterations*frames/sec ({args.workload})",
            "value": units / elapsed,
            "unit": "iterations*frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: griffin_lim batch={batch}/GPU n_fft={n_fft} hop={hop} "
                                   f"n_frames={frames} maxiter={iters} alpha={alpha} hann center reflect tol=0 "
                                   f"eva_iter=10",
                       "global_batch": batch * world, "parallelism": f"batch-sharded x{world}, RCCL gather",
                       "kernel_path": {"fused": "fused wave-per-frame", "frame": "wave-level frame kernel + overlap-add",
                                       "generic": "generic (LDS FFT frame kernel + overlap-add)"}[path],
                       "step": "phase_init + ISTFT + iterations + gather"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": load_traffic(args.workload if fused else args.workload + "_generic"),
                         "kernel": {"fused": f"specinv::fast::k_fused4<{n_fft // 128}, GLA>", "frame": "k_semi+k_ola_f4",
                                    "generic": "k_iter_pair+k_ola"}[path],
                         "launch_ms": launch_ms, "algorithmic_bytes_per_launch": launch_bytes,
                         "bytes_per_unit": unit_bytes},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n_fft, hop, frames, alpha)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
