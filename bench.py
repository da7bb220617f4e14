#!/usr/bin/env python3
"""Benchmarks of the BASELINE.json configurations on MI355X.

Default (the headline, BASELINE.json configs[1], "C2"): griffin_lim, batch 64 PER GPU (weak scaling),
n_fft=2048, hop=512, n_frames=1024, 100 iterations, alpha=0.3, periodic Hann window, center/reflect, tol=0,
eva_iter=10, metric 'sc'; magnitudes uniform[0,1) from default_rng(1234 + rank).  One "step" = one
complete inversion of the batch: phase_init + initial ISTFT + 100 fused iterations (+ the RCCL gather of the
waveforms to rank 0 when N > 1).  Inputs are resident in HBM before the timed region; plan creation is outside.

Other workloads (`--workload`), same JSON contract:
  C1  griffin_lim B=1 n_fft=1024 hop=256 T=512 50 it alpha=0 (the reference's CPU-runnable case)
  C3  RTISI_LA B=32/GPU n_fft=2048 hop=512 T=1024 look_ahead=3 25 it (frame-serial per item; `--asym` for the
      asymmetric-window form)
  C4  ADMM B=32/GPU (256 over 8 GPUs) n_fft=1024 hop=256 T=2048 rho=0.1 200 it, batch-sharded + RCCL gather
  C5  L_BFGS from 80-bin log-mel B=16 n_fft=2048 hop=512 T=1024: a step = `--outer` optimizer.step calls of 20
      closure evaluations each (whole batch = one optimisation problem: replicas only for N > 1)

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus 8                       # spawns its own 8 ranks (torch.distributed.run) when WORLD_SIZE is unset
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 3 --warmup 1 [--workload C4]

Rank 0 prints ONE compact JSON line (< 6 KB: the driver keeps an 8 KB tail; `--verbose` prints the verbose record instead, which
is always written to gpurun_out/bench_detail.json).  Beside the contract's keys:
  roofline       the dominant kernel: `achieved` = the bytes the kernel HAS to move per launch (bytes_per_unit x
                 units_per_launch) / launch time, against the 8 TB/s HBM peak (`frac`); `traffic` = the HBM bytes the counters saw
                 per launch; `limiter` / `limiter_frac` = the resource that binds the kernel and how much of it is in use ("valu": share
                 of the chip's vector issue slots, "hbm": counter bytes / launch time / 8 TB/s, "latency": RTISI-LA, reported with
                 the dependent-step rate).  Launch time: HIP events on the launch stream inside the timed region.  Counters:
                 rocprofv3 --pmc passes over a child process of THIS run (N = 1; `--no-pmc` or a failed pass falls back to
                 profiles/traffic.json and says so).  `reference_bytes_frac`: SURVEY 8d's bytes of the REFERENCE ALGORITHM over the
                 same launch time (a speed-up over that algorithm: may exceed 1).
  value_incl_h2d the same step with the target magnitudes staged from pinned host memory each step (SURVEY 8d "report both")
  legs           short C4 / C3 / C5 / C5_mfma / C5_wolfe / C5_memory / C1 legs run in the same process after the timed region (default
                 C2 run at N = 1 only), one row each (`legs_cols`)
  cpu_baseline   the NumPy oracle timed on the host cores on a bounded sample of the same workload (N = 1, rank 0)
  check          an independent re-evaluation of the result outside the timed region
  legend         what the figures mean, every string once
"""
import argparse
import csv
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

WORKLOADS = {
    # name: (method, batch per GPU, n_fft, hop, frames, iterations, coefficient)
    "C2": ("griffin_lim", 64, 2048, 512, 1024, 100, 0.3),
    "C1": ("griffin_lim", 1, 1024, 256, 512, 50, 0.0),
    "C3": ("RTISI_LA", 32, 2048, 512, 1024, 25, 0.99),
    "C4": ("ADMM", 32, 1024, 256, 2048, 200, 0.1),
    "C5": ("L_BFGS", 16, 2048, 512, 1024, 20, None),
    # the coverage path (every dtype / size the reference's own tests sweep, test/test_griffin.py:9-32) - extra legs only
    "F64": ("griffin_lim", 16, 2048, 512, 1024, 100, 0.3),      # float64 at the headline frame size
    "S32": ("griffin_lim", 64, 256, 64, 4096, 100, 0.3),        # float32 at n_fft 256 (test/consts.py:1-3)
    "W400": ("griffin_lim", 64, 400, 160, 2048, 100, 0.3),      # float32 at 25 ms / 10 ms of 16 kHz speech: a size that is not a power of two
}
LEG_DTYPE = {"F64": torch.float64}
COVERAGE_LEGS = ("F64", "S32", "W400")
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
N_SIMD = 1024                  # 256 CUs x 4 SIMDs
N_MELS, SR, LOOK_AHEAD = 80, 22050, 3

# The dominant kernel of each workload at its BASELINE shape, as rocprofv3 names it (substring of Kernel_Name), and what binds it.
# tools/collect_workload_profiles.py reads the same table.
DOMINANT = {
    "C2": ("specinv::fast::k_fused4_td<16, false, false>", "valu"),
    "C4": ("specinv::fast::k_fused4<8, 1, false>", "hbm"),
    "C3": ("specinv::fast::k_rtisi_fast<16, 256, 4>", "latency"),
    "C5": ("specinv::fast::k_objective_walk<16, 4>", "valu"),
    "C1": ("specinv::fast::k_semi<8", "latency"),
    "F64": ("specinv::wave::k_wave_iter<double, 10", "hbm"),
    "S32": ("specinv::wave::k_wave_iter<float, 7", "hbm"),
    "W400": ("specinv::wave::k_wave_iter<float, 200", "hbm"),
}
PMC_PASSES = (
    ("fetch", ["FETCH_SIZE"]),
    ("write", ["WRITE_SIZE"]),
    ("sq", ["SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_ANY",
            "GRBM_GUI_ACTIVE"]),
)
VALU_FORMULA = ("valu_issue_frac = SQ_ACTIVE_INST_VALU * 4 / (GRBM_GUI_ACTIVE / 8 * 1024): SQ_ACTIVE_INST_VALU counts quad-cycles "
                "summed over the 1024 SIMDs, GRBM_GUI_ACTIVE busy cycles summed over the 8 XCDs (means per launch of the kernel)")
HBM_FORMULA = ("traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 B per launch (gfx950: FETCH_SIZE tallies a 16-B-per-lane streaming "
               "read at half its bytes, MI355X_MICROARCH.md HBM section; separate --pmc passes); frac = traffic / launch time / 8 TB/s")


def hann(n):
    return (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n) / n)).astype(np.float32)


def algorithmic_bytes_per_unit(method, hop, n_freq, coef):
    """SURVEY 8d, one frame through one iteration / evaluation of the REFERENCE ALGORITHM, fp32 (`roofline.reference_bytes_equiv`).
    griffin_lim: x read+write 8*hop, target 4F, pre_spec read+write 16F (8*hop + 4F when alpha == 0); ADMM: X and U read+write
    32F + target 4F; L_BFGS objective: x read + gradient write 8*hop, target 4*n_mels; RTISI_LA: target read 4F + committed frame
    4*hop per frame (state lives in LDS: not what bounds that kernel)."""
    if method == "griffin_lim":
        return 8 * hop + (20 if coef != 0 else 4) * n_freq
    if method == "ADMM":
        return 8 * hop + 36 * n_freq
    if method == "L_BFGS":
        return 8 * hop + 4 * N_MELS
    return 4 * n_freq + 4 * hop


def restated_bytes_per_unit(method, hop, n_freq, kernel):
    """What the shipped kernel has to move after the reformulations of DESIGN 3.2 / 3.3 (None: the reference algorithm's bytes).
    Griffin-Lim with the momentum carried as a (B, L) signal: z in 4h, target 4F, z out 4h (x out only when somebody reads it;
    the first iterations also read the starting spectrum) instead of 8h + 20F; ADMM on Y = X + U alone: 8h + 20F instead of
    8h + 36F (methods.py:467-468 only ever read X + U)."""
    if method == "griffin_lim" and kernel in ("k_fused4_td", "k_fused_td", "k_hop_td"):
        return 8 * hop + 4 * n_freq
    if method == "ADMM":
        return 8 * hop + 20 * n_freq
    if method == "L_BFGS" and kernel == "objective+step":      # x_old, g_prev read; x_new, g written; targets
        return 16 * hop + 4 * N_MELS
    return None


# ---------------------------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle; the only place this file touches oracle/)
def _timed_oracle(run, units_per_iter, budget_s, first=2, cap=400):
    iters = first
    while True:
        t0 = time.perf_counter()
        run(iters)
        dt = time.perf_counter() - t0
        if dt >= budget_s / 2 or iters >= cap:
            return iters, dt, iters * units_per_iter / dt
        iters = min(cap, max(iters * 2, int(iters * budget_s / max(dt, 1e-3))))


def cpu_baseline(method, n_fft, hop, frames, coef, budget_s=15.0):
    """The oracle (a port of the reference's algorithm) on the host cores, bounded sample of the same workload."""
    import oracle
    from oracle import stftlib
    cores = os.cpu_count() or 1
    stftlib.WORKERS = cores
    rng = np.random.default_rng(99)
    w = hann(n_fft)
    n_freq = n_fft // 2 + 1
    try:
        if method in ("griffin_lim", "ADMM"):
            b = 8 if method == "griffin_lim" else 4
            mag = rng.random((b, n_freq, frames), dtype=np.float32)
            init = oracle.phase_init(mag, hop_length=hop, window=w)
            if method == "griffin_lim":
                def run(n):
                    oracle.griffin_lim(init, max_iter=n, alpha=coef, tol=0, eva_iter=10, hop_length=hop, window=w)
            else:
                def run(n):
                    oracle.admm(init, max_iter=n, rho=coef, tol=0, eva_iter=10, hop_length=hop, window=w)
            run(1)                                                                     # warm caches
            iters, dt, rate = _timed_oracle(run, b * frames, budget_s)
            return {"value": rate, "unit": "iterations*frames/s", "cores": cores, "kind": "port",
                    "sample": f"oracle.{'griffin_lim' if method == 'griffin_lim' else 'admm'} batch={b} n_fft={n_fft} "
                              f"hop={hop} n_frames={frames} {iters} iterations coef={coef} ({dt:.1f} s, scipy.fft "
                              f"workers={cores})"}
        if method == "RTISI_LA":
            b, t, its = 2, 12, 25
            while True:                                                                # at least ~2 s of work (frame-serial)
                mag = rng.random((b, n_freq, t), dtype=np.float32)
                t0 = time.perf_counter()
                oracle.rtisi_la(mag, look_ahead=LOOK_AHEAD, asymmetric_window=False, max_iter=its, alpha=coef,
                                hop_length=hop, window=w)
                dt = time.perf_counter() - t0
                if dt >= 2.0 or t >= frames:
                    break
                t = min(frames, max(2 * t, int(t * 3.0 / max(dt, 1e-3))))
            return {"value": its * b * t / dt, "unit": "iterations*frames/s", "cores": 1, "kind": "port",
                    "sample": f"oracle.rtisi_la batch={b} n_fft={n_fft} hop={hop} n_frames={t} look_ahead={LOOK_AHEAD} "
                              f"{its} iterations ({dt:.1f} s, frame-serial NumPy loop)"}
        from oracle import lbfgs as olb
        from spectrogram_inversion_amd.mel import mel_filterbank
        b, t = 2, 256
        a = oracle.args_helper(n_freq, np.float32, hop_length=hop, window=w)
        tr = olb.LogMelStft(a, mel_filterbank(SR, n_fft, N_MELS).astype(np.float32))
        xs = (0.1 * rng.standard_normal((b, (t - 1) * hop))).astype(np.float32)
        target = tr.forward(xs)
        x0 = (1e-6 * rng.standard_normal(xs.shape)).astype(np.float32)
        tr.loss_grad(x0, target)

        def run(n):
            for _ in range(n):
                tr.loss_grad(x0, target)
        iters, dt, rate = _timed_oracle(run, b * t, budget_s, first=2, cap=200)
        return {"value": rate, "unit": "evaluations*frames/s", "cores": cores, "kind": "port",
                "sample": f"oracle LogMelStft.loss_grad batch={b} n_fft={n_fft} hop={hop} n_frames={t} n_mels={N_MELS} "
                          f"{iters} evaluations ({dt:.1f} s, scipy.fft workers={cores})"}
    finally:
        stftlib.WORKERS = 1


# ---------------------------------------------------------------------------------------------------------------------------
# Counters
def load_traffic():
    """Stored counter figures of the dominant kernels: profiles/traffic.json (tools/collect_workload_profiles.py from the
    committed rocprofv3 --pmc summaries) and the hash of the kernel sources they were measured on: the fall-back of `live_pmc`."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return {}


def summarise_counters(c):
    """Derived figures of one kernel from its mean counter values per launch (`c`: {counter: mean})."""
    out = {"counters": c}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        out["hbm_bytes_per_launch"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
    if c.get("GRBM_GUI_ACTIVE") and "SQ_ACTIVE_INST_VALU" in c:
        out["valu_issue_frac"] = c["SQ_ACTIVE_INST_VALU"] * 4 / (c["GRBM_GUI_ACTIVE"] / 8 * N_SIMD)
    if c.get("SQ_WAVE_CYCLES") and "SQ_ACTIVE_INST_VALU" in c:
        out["valu_share_of_wave_life"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"]
    if c.get("SQ_LDS_IDX_ACTIVE") and "SQ_LDS_BANK_CONFLICT" in c:
        out["lds_conflict_frac"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    return out


def under_profiler():
    env = os.environ
    return any(k.startswith("ROCPROF") or k.startswith("ROCPROFILER") for k in env) or "rocprofiler" in env.get("LD_PRELOAD", "")


def live_pmc(workloads, timeout_s=150, keep=None):
    """rocprofv3 --pmc passes (counters only: no trace domain beside them) around a CHILD process that runs one step of each of
    `workloads` on this GPU - started as a child (`subprocess`), never an exec of this process, which has initialised the GPU.
    One pass per counter group (FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots").
    Returns {workload: summarise_counters(...)} for the dominant kernel of each, {} on any failure.  `keep`: directory to
    copy the raw per-dispatch CSVs to (tools/profile_workloads.sh)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe) or under_profiler():
        return {}, "rocprofv3 not available (or this process is itself profiled)"
    tmp = tempfile.mkdtemp(prefix="specinv_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
    env.pop("SPECINV_BENCH_BACKEND", None)
    means = {w: {} for w in workloads}
    t0 = time.perf_counter()
    try:
        for tag, counters in PMC_PASSES:
            out_dir = os.path.join(tmp, tag)
            cmd = [exe, "--pmc", *counters, "--output-format", "csv", "-d", out_dir, "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--pmc-child", ",".join(workloads)]
            r = subprocess.run(cmd, cwd=tmp, env=env, capture_output=True, text=True, timeout=timeout_s)
            files = glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return {}, f"pass {tag}: rc {r.returncode}: {(r.stderr or r.stdout)[-300:]}"
            agg = {w: {} for w in workloads}
            for f in files:
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        for w in workloads:
                            if DOMINANT[w][0] in row["Kernel_Name"]:
                                agg[w].setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
                if keep:
                    os.makedirs(keep, exist_ok=True)
                    shutil.copy(f, os.path.join(keep, f"pmc_{tag}_counter_collection.csv"))
            for w in workloads:
                for k, v in agg[w].items():
                    means[w][k] = sum(v) / len(v)
                    means[w].setdefault("_launches", {})[k] = len(v)
        out = {}
        for w in workloads:
            n = means[w].pop("_launches", {})
            if means[w]:
                out[w] = summarise_counters(means[w])
                out[w]["launches_counted"] = min(n.values()) if n else 0
        return out, f"measured in this run: {len(PMC_PASSES)} rocprofv3 --pmc passes over a child process ({time.perf_counter() - t0:.0f} s)"
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError) as e:
        return {}, f"{type(e).__name__}: {e}"[:300]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def stored_pmc(workload, generic=False):
    tr = load_traffic()
    key = workload + ("_generic" if generic else "")
    out = {}
    if tr.get(key) is not None:
        out["hbm_bytes_per_launch"] = tr[key]
    for k in ("valu_issue_frac", "valu_share_of_wave_life", "lds_conflict_frac"):
        if tr.get(f"{key}_{k}") is not None:
            out[k] = tr[f"{key}_{k}"]
    from spectrogram_inversion_amd.build import sources_hash
    return out, tr.get("csrc_sha1"), tr.get("csrc_sha1") != sources_hash()


# ---------------------------------------------------------------------------------------------------------------------------
def sc_lin_f64(x_item, mag_item, hop, window, dev):
    """||  |STFT(x)| - m || / || m ||  of one item, evaluated by the float64 generic kernels."""
    from spectrogram_inversion_amd.plan import Plan, args_helper
    m = mag_item.to(torch.float64)[None]
    a = args_helper(m, hop_length=hop, window=window.double())
    p = Plan(a, 1, m.shape[2], torch.float64, dev)
    s = p.stft(x_item.to(torch.float64)[None])
    out = (torch.linalg.norm(s.abs() - m) / torch.linalg.norm(m)).item()
    return out, p


def ev():
    return torch.cuda.Event(enable_timing=True)


class Leg:
    """One workload set up on this rank's GPU: `step()` runs one complete inversion; HIP events on the launch stream bracket
    the dominant kernel's launches."""

    def __init__(self, workload, o, rank, world, dev, backend):
        import spectrogram_inversion_amd as si
        from spectrogram_inversion_amd.plan import args_helper, get_plan
        self.workload, self.o, self.rank, self.world, self.dev, self.backend = workload, o, rank, world, dev, backend
        self.method, self.batch, self.n_fft, self.hop, self.frames, self.iters, self.coef = WORKLOADS[workload]
        if o.batch:
            self.batch = o.batch
        self.n_freq = self.n_fft // 2 + 1
        rng = np.random.default_rng(1234 + rank)
        self.dtype = LEG_DTYPE.get(workload, torch.float32)
        self.window = torch.from_numpy(hann(self.n_fft)).to(self.dtype)
        self.events, self.pending, self.state = [], [], {}
        self.folded = {"ms": 0.0, "n": 0}
        self.counters = {"evals": 0}
        self.gather_wait_s = 0.0
        self.plan = self.mag = None
        m = self.method
        if m != "L_BFGS":
            mag_np = rng.random((self.batch, self.n_freq, self.frames), dtype=np.float32)
            self.mag = torch.from_numpy(mag_np).to(dev).to(self.dtype)
            self.plan = get_plan(args_helper(self.mag, hop_length=self.hop, window=self.window), self.batch, self.frames,
                                 self.dtype, dev)
            if o.generic:
                self.plan.force_generic(True)
            self.length = self.plan.length
            self.unit = "iterations*frames/s"
            self.units_per_step = self.iters * self.batch * self.frames
        else:
            self.length = (self.frames - 1) * self.hop
            fb = torch.from_numpy(si.mel_filterbank(SR, self.n_fft, N_MELS)).to(dev)
            self.tr = si.LogMelSTFT(fb, self.n_fft, hop_length=self.hop, window=self.window)
            gen = torch.Generator(device="cpu").manual_seed(1234 + rank)
            xs = (0.1 * torch.randn(self.batch, self.length, generator=gen)).to(dev)
            self.target = self.tr(xs)
            self.x_init = (1e-6 * torch.randn(self.batch, self.length, generator=gen)).to(dev)
            self.fwd, self.fg_raw = self.tr.bind(self.x_init, self.target)
            self.unit = "evaluations*frames/s"
            self.units_per_step = None                         # closure evaluations are counted
            self.opt_kw = {"baseline": {}, "main": dict(max_iter=50, history_size=10),
                           "wolfe": dict(line_search_fn="strong_wolfe"),
                           "memory": dict(line_search_fn="strong_wolfe", tolerance_grad=0.0, tolerance_change=0.0)}[o.c5_variant]

    # -- one step ---------------------------------------------------------------------------------------------------------
    def finish_step(self, x):
        if self.world > 1 and self.method != "L_BFGS":
            # RCCL gather of the (B, L) waveforms to rank 0; it runs on RCCL's stream, so the next step's kernels overlap it -
            # every gather is completed (`result()`) inside the timed region.  --no-overlap-gather waits for it right away.
            from spectrogram_inversion_amd.distributed import gather_waveforms
            t0 = time.perf_counter()
            if self.pending:
                self.pending.pop().result()
            if self.backend == "nccl":
                h = gather_waveforms(x, dst=0, sizes=[self.batch] * self.world, async_op=True)
                if self.o.no_overlap_gather:
                    h.result()
                    torch.cuda.current_stream().synchronize()
                self.pending.append(h)
            else:                                            # (rehearsal: a blocking gather through host memory)
                done = gather_waveforms(x, dst=0, sizes=[self.batch] * self.world)
                self.pending.append(type("Done", (), {"result": staticmethod(lambda done=done: done)})())
            self.gather_wait_s += time.perf_counter() - t0
        self.state["x"] = x

    def step(self, mag=None):
        m, plan = self.method, self.plan
        mag = self.mag if mag is None else mag
        if m in ("griffin_lim", "ADMM"):
            (plan.gla_init if m == "griffin_lim" else plan.admm_init)(None, mag, self.coef)   # phase_init + initial ISTFT
            e0, e1 = ev(), ev()
            e0.record()                                     # HIP events on the stream the kernels are launched on
            done, evals = plan.run(self.iters, 10, 0.0, "sc")    # evaluation every 10, sums stay on the device
            e1.record()
            assert done == self.iters
            self.state["evals"] = evals
            self.events.append((e0, e1, self.iters))
            self.finish_step(plan.wave())
        elif m == "RTISI_LA":
            e0, e1 = ev(), ev()
            e0.record()
            x = plan.rtisi(mag, LOOK_AHEAD, self.o.asym, self.iters, self.coef)
            e1.record()
            self.events.append((e0, e1, 1))
            self.finish_step(x)
        else:
            from spectrogram_inversion_amd.lbfgs import LBFGS
            x = self.x_init.clone()
            opt = LBFGS(x, device=self.dev, **self.opt_kw)    # torch.optim.LBFGS defaults: max_iter 20, history 100, lr 1
            # HIP events around every 8th objective evaluation (on the launch stream, recorded by the library: an event pair costs
            # ~10 us of the timeline, so bracketing every one of the 1000 evaluations of a step would lower the throughput measured)
            opt.time_objective = 0 if os.environ.get("SPECINV_BENCH_NO_EVENTS") else 8
            fg = self._timed_fg()
            for _ in range(self.o.outer):
                opt.step(fg)
            self.state["opt"] = opt
            if opt.objective_launches:                       # evaluations the device-resident optimiser ran (and timed a sample of)
                self.folded["ms"] += opt.objective_ms
                self.folded["n"] += opt.objective_timed
                self.counters["evals"] += opt.objective_launches
            self.finish_step(x)

    def _timed_fg(self):
        events, folded, counters, fg_raw = self.events, self.folded, self.counters, self.fg_raw

        def timed_eval(call):
            # finished event pairs are folded into a running sum as we go: hundreds of live HIP events slow every launch down
            # (measured: 20 steps x 40 evaluations with all pairs kept alive ran 20 % slower than 3 steps)
            while events and events[0][1].query():
                a, b, n = events.pop(0)
                folded["ms"] += a.elapsed_time(b)
                folded["n"] += n
            # an event pair costs ~10 us of the timeline - 4 % of a host-driven evaluation: every 4th evaluation is bracketed (the
            # device-resident optimiser samples every 8th of its own launches, LBFGS.time_objective)
            counters["evals"] += 1
            if os.environ.get("SPECINV_BENCH_NO_EVENTS") or counters["evals"] % 4 != 1:
                return call()
            e0, e1 = ev(), ev()
            e0.record()                                     # HIP events on the stream the objective is launched on
            out = call()
            e1.record()
            events.append((e0, e1, 1))
            return out

        def fg(v):
            return timed_eval(lambda: fg_raw(v))

        fg.dev = lambda v, loss_ptr: timed_eval(lambda: fg_raw.dev(v, loss_ptr))   # loss left on the device: no sync per evaluation
        # the device-resident optimiser (csrc/lbfgs_dev.h) enqueues a whole optimizer.step from C++: it times its own objective
        # launches with HIP events on the launch stream (LBFGS.time_objective) and counts the evaluations the device executed
        if hasattr(fg_raw, "device_objective"):
            fg.device_objective = fg_raw.device_objective
        if hasattr(fg_raw, "dev_stats"):              # (strong Wolfe, host-driven: objective + step statistics in one go)
            fg.dev_stats = lambda v, d, out_ptr: timed_eval(lambda: fg_raw.dev_stats(v, d, out_ptr))
        return fg

    def fence(self):
        import torch.distributed as dist
        t0 = time.perf_counter()
        out = self.pending.pop().result() if self.pending else None
        self.gather_wait_s += time.perf_counter() - t0
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        return out

    # -- the timed region -------------------------------------------------------------------------------------------------
    def run(self, steps, warmup, step=None):
        """W untimed steps, then EXACTLY `steps` timed ones between barrier + synchronize on both sides.  Returns the elapsed
        seconds of this rank and what the last gather delivered."""
        step = step or self.step
        for _ in range(warmup):
            step()
        self.fence()
        self.events.clear()
        self.counters["evals"] = 0
        self.folded["ms"], self.folded["n"] = 0.0, 0
        self.gather_wait_s = 0.0
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        gathered = self.fence()
        elapsed = time.perf_counter() - t0
        return elapsed, gathered

    def units(self, steps):
        if self.method == "L_BFGS":
            return self.counters["evals"] * self.batch * self.frames * self.world
        return steps * self.units_per_step * self.world

    def freeze(self):
        """Pin the figures of the timed region that just ended (later legs on this object reuse the event lists)."""
        self._frozen = None
        self._frozen = (self.launch_ms(), self.counters["evals"])

    def launch_ms(self):
        """dominant kernel: average duration over the timed region from the HIP events (launches back to back on one stream)"""
        if getattr(self, "_frozen", None):
            return self._frozen[0]
        n = sum(n for _, _, n in self.events) + self.folded["n"]
        ms = sum(a.elapsed_time(b) for a, b, _ in self.events) + self.folded["ms"]
        return ms / max(1, n), n

    def kernel_info(self):
        m = self.method
        if m in ("griffin_lim", "ADMM"):
            g = self.plan.launch_geometry
            r = self.n_fft // 128
            name = {"k_fused4": f"specinv::fast::k_fused4<{r}, {'GLA' if m == 'griffin_lim' else 'ADMM'}>",
                    "k_fused4_td": f"specinv::fast::k_fused4_td<{r}> (momentum carried as a signal; late and early (+c0) launches "
                                   f"and the evaluation kernel k_eval_td averaged in launch_ms)",
                    "k_fused": f"specinv::fast::k_fused<{r}, {self.n_fft // self.hop}>", "k_semi": "k_semi+k_ola_f4",
                    "k_fused_td": f"specinv::fast::k_fused_td<{r}, {self.n_fft // self.hop}>", "k_hop": "k_hop",
                    "k_hop_td": "k_hop_td", "k_iter_pair": "k_iter_pair+k_ola",
                    "k_wave_iter": "specinv::wave::k_wave_iter" + ("" if g["chunks"] < self.frames else "+k_ola")}[g["kernel"]]
            return self.plan.path, g, name
        if m == "RTISI_LA":
            fast = self.plan.fast_path
            return self.plan.path, {"kernel": "k_rtisi_fast" if fast else "k_rtisi"}, \
                (f"specinv::k_rtisi_fast<{self.n_fft // 128}>" if fast else "specinv::k_rtisi")
        # (the device-resident optimiser leaves the step x += t d of an iteration that accepts no pair to the next evaluation's
        # frame walk - DESIGN 3.7 (d): the launch then also reads g_prev and writes the new iterate)
        opt = (getattr(self, "state", None) or {}).get("opt")
        dev_opt = bool(getattr(opt, "_dev", False)) and "line_search_fn" not in (getattr(self, "opt_kw", None) or {})
        obj = getattr(getattr(self, "fg_raw", None), "device_objective", None)
        walk = bool(obj) and obj[0].objective_kind == "walk"
        deferred = dev_opt and walk and os.environ.get("SPECINV_LBFGS_DEFER", "1") != "0"
        return "fused", {"kernel": "objective+step" if deferred else "objective"}, \
            "L-BFGS objective (forward + loss + gradient in one launch" + (", the optimiser's step applied on the way)" if deferred else ")")

    def describe(self):
        o = self.o
        return {
            "griffin_lim": f"griffin_lim batch={self.batch}/GPU n_fft={self.n_fft} hop={self.hop} n_frames={self.frames} "
                           f"maxiter={self.iters} alpha={self.coef} hann center reflect tol=0 eva_iter=10",
            "ADMM": f"ADMM batch={self.batch}/GPU n_fft={self.n_fft} hop={self.hop} n_frames={self.frames} rho={self.coef} "
                    f"maxiter={self.iters} hann tol=0 eva_iter=10",
            "RTISI_LA": f"RTISI_LA batch={self.batch}/GPU n_fft={self.n_fft} hop={self.hop} n_frames={self.frames} "
                        f"look_ahead={LOOK_AHEAD} maxiter={self.iters} alpha={self.coef} asymmetric_window={o.asym} hann",
            "L_BFGS": f"L_BFGS log-mel-{N_MELS} batch={self.batch} n_fft={self.n_fft} hop={self.hop} n_frames={self.frames} "
                      f"maxiter={o.outer} (optimizer.step calls) x LBFGS("
                      f"{'defaults: max_iter 20, history 100, lr 1' if not getattr(self, 'opt_kw', None) else self.opt_kw})",
        }[self.method]

    # -- roofline ---------------------------------------------------------------------------------------------------------
    def roofline(self, pmc, pmc_source, stale=None, stored_sha=None):
        """The dominant kernel against the resource that binds it (module docstring)."""
        from spectrogram_inversion_amd.build import sources_hash
        path, geo, kernel = self.kernel_info()
        launch_ms, n_launch = self.launch_ms()
        secs = launch_ms * 1e-3
        kname = (geo or {}).get("kernel")
        default_shape = self.batch == WORKLOADS[self.workload][1] and not self.o.generic and (path != "generic" or self.workload in COVERAGE_LEGS)
        bound = DOMINANT[self.workload][1] if default_shape else ("valu" if path == "generic" else "hbm")
        units = self.batch * self.frames
        es = 2 if self.dtype == torch.float64 else 1              # (SURVEY 8d prices float32 elements)
        ref_unit = es * algorithmic_bytes_per_unit(self.method, self.hop, self.n_freq, self.coef)
        must_unit = restated_bytes_per_unit(self.method, self.hop, self.n_freq, kname) or ref_unit
        if kname == "k_wave_iter" and geo["chunks"] >= self.frames:   # frames buffer + k_ola: the frames' round trip on top
            must_unit += es * 8 * self.n_fft
        pmc = pmc or {}
        traffic = pmc.get("hbm_bytes_per_launch")
        vfrac = pmc.get("valu_issue_frac")
        hbm = {"peak": HBM_PEAK_GBS, "unit": "GB/s", "bytes_per_unit": must_unit, "algorithmic_bytes_per_launch": must_unit * units,
               "achieved_algorithmic": must_unit * units / secs / 1e9, "frac_algorithmic": must_unit * units / secs / 1e9 / HBM_PEAK_GBS,
               "what": "bytes this kernel has to move per frame-iteration (DESIGN 3) over the measured launch time; `achieved` / "
                       "`frac`: the counters' bytes over the same time", "formula": HBM_FORMULA}
        if traffic:
            hbm["achieved"] = traffic / secs / 1e9
            hbm["frac"] = hbm["achieved"] / HBM_PEAK_GBS
            hbm["traffic_over_algorithmic"] = traffic / (must_unit * units)
        roof = {"bound": "hbm", "limiter": bound, "kernel": kernel,
                "dominant_kernel_symbol": DOMINANT[self.workload][0] if default_shape else None,
                "launch_ms": launch_ms, "launches_timed": n_launch, "traffic": traffic, "traffic_source": pmc_source,
                "csrc_sha1": sources_hash()}
        if stale is not None:
            roof["traffic_stale"], roof["traffic_csrc_sha1"] = bool(stale), stored_sha
        # the contract's figures: ALGORITHMIC bytes the kernel has to move per launch (bytes per unit x units per launch)
        # over the launch time measured by HIP events, against the 8 TB/s HBM peak; `traffic` = what the counters saw per launch
        roof.update(achieved=hbm["achieved_algorithmic"], peak=HBM_PEAK_GBS, unit="GB/s", frac=hbm["frac_algorithmic"],
                    bytes_per_unit=must_unit, units_per_launch=units)
        if traffic:
            roof["traffic_frac"] = hbm["frac"]
        # ... and the resource that really limits the kernel (`limiter`): "valu" -> the share of the chip's vector issue slots in use,
        # "hbm" -> the counters' bytes over the launch time, "latency" -> dependent steps per second
        c = pmc.get("counters", {})
        if vfrac is not None:
            roof["valu_issue_frac"] = vfrac
        roof["limiter_frac"] = {"valu": vfrac, "hbm": hbm.get("frac", hbm["frac_algorithmic"]), "latency": vfrac}[bound]
        roof["hbm"] = hbm
        for k in ("valu_issue_frac", "valu_share_of_wave_life", "lds_conflict_frac", "launches_counted"):
            if pmc.get(k) is not None:
                roof[k] = pmc[k]
        if pmc.get("counters"):
            roof["counters_mean_per_launch"] = pmc["counters"]
        roof["reference_bytes_equiv"] = {
            "bytes_per_unit": ref_unit, "bytes_per_launch": ref_unit * units, "achieved": ref_unit * units / secs / 1e9, "unit": "GB/s",
            "over_hbm_peak": ref_unit * units / secs / 1e9 / HBM_PEAK_GBS,
            "what": "SURVEY 8d's bytes of the REFERENCE ALGORITHM per frame-iteration over this kernel's launch time: a speed-up over "
                    "that algorithm, not a fraction of anything (the kernel moves fewer bytes: may exceed 1)"}
        if self.method == "RTISI_LA":
            roof["dependent_steps_per_s"] = (self.frames + LOOK_AHEAD) * self.iters / secs
            roof["note"] = ("serial-latency-bound: (T + LA) * max_iter dependent steps per item, one workgroup per item (32 of 256 "
                            "CUs); frac = the chip's vector issue share, dependent_steps_per_s is the figure to watch")
        elif self.method == "griffin_lim" and must_unit != ref_unit:
            roof["note"] = ("the momentum is carried as a (B, L) signal (pre_t = STFT(z_t) + (-lr)^t c0, DESIGN 3.2): the kernel moves "
                            "8 hop + 4 F bytes per frame-iteration and is bound by the vector issue rate of its two FFTs per frame")
        elif self.method == "ADMM":
            roof["note"] = ("the kernel carries Y = X + U alone (methods.py:467-468 only read the sum, bit-identical): 8 hop + 20 F "
                            "bytes per frame-iteration instead of SURVEY's 8 hop + 36 F")
        elif self.method == "L_BFGS":
            roof["note"] = ("two FFTs per frame and the two mel contractions on the vector units (frame walk; SPECINV_OBJ_SPARSE=0: "
                            "contractions on the matrix cores), the spectrum never leaves the chip; HBM view in `hbm`")
            roof["evaluations_timed"] = self._frozen[1] if getattr(self, "_frozen", None) else self.counters["evals"]
        return roof

    def lbfgs_info(self):
        opt = self.state["opt"]
        return {"variant": self.o.c5_variant, "outer_steps": self.o.outer, "inner_iterations": opt.total_iters,
                "evaluations": opt.func_evals, "pairs_accepted": int(opt.pairs_accepted),
                "pairs_rejected": int(opt.pairs_rejected), "history_len": opt.history_len, "history_size": opt.history_size,
                "decisions": "on the device, one host synchronisation per optimizer.step" if opt._dev
                             else "on the host, one synchronisation per inner iteration"}

    # -- H2D-inclusive step ("excludes plan creation / H2D of inputs - report both") ---------------------------------
    def h2d_legs(self, steps, warmup):
        """The step with its input staged from pinned host memory: (a) serial - the copy on the launch stream in front of every
        step; (b) pipelined - the next step's target copied on a second stream into a second buffer while this step iterates."""
        if self.method == "L_BFGS":
            return None
        host = torch.empty(self.mag.shape, dtype=self.mag.dtype, pin_memory=True)
        host.copy_(self.mag)
        bufs = [torch.empty_like(self.mag), torch.empty_like(self.mag)]
        nbytes = self.mag.numel() * self.mag.element_size()

        def serial():
            bufs[0].copy_(host, non_blocking=True)
            self.step(bufs[0])
        el, _ = self.run(steps, warmup, serial)
        out = {"bytes_per_step": nbytes,
               "serial": {"ms_per_step": 1e3 * el / steps, "value": self.units(steps) / el,
                          "what": "hipMemcpyAsync of the target from pinned host memory on the launch stream, then the step"}}
        copy_stream = torch.cuda.Stream(self.dev)
        ready = [torch.cuda.Event(), torch.cuda.Event()]
        freed = [torch.cuda.Event(), torch.cuda.Event()]
        k = {"i": 0}

        def stage(slot):
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(freed[slot])
                bufs[slot].copy_(host, non_blocking=True)
                ready[slot].record(copy_stream)

        def pipelined():
            slot = k["i"] & 1
            k["i"] += 1
            torch.cuda.current_stream().wait_event(ready[slot])
            stage(slot ^ 1)                                   # the next step's input, under this step's iterations
            self.step(bufs[slot])
            freed[slot].record()
        for s in (0, 1):
            freed[s].record()
        stage(0)
        el, _ = self.run(steps, warmup, pipelined)
        torch.cuda.synchronize()
        out["pipelined"] = {"ms_per_step": 1e3 * el / steps, "value": self.units(steps) / el,
                            "what": "the next step's target copied on a second HIP stream into a second buffer while this step "
                                    "iterates (double-buffered)"}
        return out


# ---------------------------------------------------------------------------------------------------------------------------
def check(leg, x):
    """Independent re-evaluation of item 0 of the result, outside the timed region: the spectral convergence of the
    float32 result as the float64 generic kernels measure it, against a complete float64 re-run from the same
    starting spectrum (linear scale; the north-star bar for Griffin-Lim is 1e-5)."""
    method, dev, hop, window, iters, coef = leg.method, leg.dev, leg.hop, leg.window, leg.iters, leg.coef
    if method == "L_BFGS":
        # the loss the objective kernel reports against the loss recomputed from its own forward pass by the metric
        # kernel, and the directional derivative of the loss against g . d
        fwd, fg_raw, x0, target = leg.fwd, leg.fg_raw, leg.x_init, leg.target
        xr = (x0 + 1e-3 * torch.randn_like(x0)).contiguous()
        loss, g = fg_raw(xr)
        v = fwd(xr)
        mse = float(((v.double() - target.double()) ** 2).mean())
        d = g / g.norm()                                   # steepest direction: g . d = |g|
        eps = 1e-3 * abs(loss) / float(g.norm())            # the loss moves by ~0.1 % either way
        lp, _ = fg_raw((xr + eps * d).contiguous())
        lm, _ = fg_raw((xr - eps * d).contiguous())
        fd, gd = (lp - lm) / (2 * eps), float((g.double() * d.double()).sum())
        ok = abs(loss - mse) <= 1e-5 * abs(mse) and abs(fd - gd) <= 2e-2 * abs(gd)
        return {"what": "self-check (tripwire, not parity evidence): objective loss vs mse(forward, target); central difference vs g.d",
                "loss": loss, "mse": mse, "directional_fd": fd, "directional_g": gd, "ok": bool(ok)}
    mag = leg.mag
    from spectrogram_inversion_amd.plan import Plan, args_helper
    if leg.workload in COVERAGE_LEGS:
        # coverage legs: item 0 again on the workgroup-level kernels k_wave_iter replaced (the same update per bin on another
        # transform), from the same starting spectrum, in the leg's own dtype
        a = args_helper(mag[:1], hop_length=hop, window=window)
        p0 = Plan(a, 1, mag.shape[2], leg.dtype, dev)
        c0 = p0.phase_init(mag[:1])
        saved = os.environ.get("SPECINV_GENERIC_WAVE")
        os.environ["SPECINV_GENERIC_WAVE"] = "0"
        try:
            pr = Plan(a, 1, mag.shape[2], leg.dtype, dev)
            pr.force_generic(True)
        finally:
            if saved is None:
                os.environ.pop("SPECINV_GENERIC_WAVE", None)
            else:
                os.environ["SPECINV_GENERIC_WAVE"] = saved
        pr.gla_init(c0, None, coef)
        pr.iterate(iters)
        xr = pr.wave()[0].double()
        m64 = mag[0].double()[None]
        pe = Plan(args_helper(m64, hop_length=hop, window=window.double()), 1, mag.shape[2], torch.float64, dev)

        def sc(v):
            s_ = pe.stft(v[None])
            return (torch.linalg.norm(s_.abs() - m64) / torch.linalg.norm(m64)).item()
        sc_a, sc_b = sc(x[0].double()), sc(xr)
        tol = 1e-5
        return {"what": "self-check (tripwire): SC_lin of item 0 (float64 evaluation) vs the same iterations on the workgroup-level "
                        "coverage kernels (k_iter_pair)", "sc_lin": sc_a, "sc_lin_ref": sc_b, "abs_diff": abs(sc_a - sc_b), "tol": tol,
                "ok": bool(abs(sc_a - sc_b) <= tol)}
    sc32, p64 = sc_lin_f64(x[0], mag[0], hop, window, dev)
    if method == "RTISI_LA":
        a = args_helper(mag[:1], hop_length=hop, window=window)
        pg = Plan(a, 1, mag.shape[2], torch.float32, dev)
        pg.force_generic(True)
        xg = pg.rtisi(mag[:1], LOOK_AHEAD, leg.o.asym, iters, coef)
        scg, _ = sc_lin_f64(xg[0], mag[0], hop, window, dev)
        tol = 2e-3
        return {"what": "self-check (tripwire, not parity evidence): SC_lin of item 0 (float64 evaluation) vs the generic RTISI-LA kernel",
                "sc_lin": sc32, "sc_lin_ref": scg, "abs_diff": abs(sc32 - scg), "tol": tol, "ok": bool(abs(sc32 - scg) <= tol)}
    # float64 re-run of item 0 from the float32 phase_init
    a32 = args_helper(mag[:1], hop_length=hop, window=window)
    p32 = Plan(a32, 1, mag.shape[2], torch.float32, dev)
    c0 = p32.phase_init(mag[:1]).to(torch.complex128)
    if method == "griffin_lim":
        p64.gla_init(c0, None, coef)
    else:
        p64.admm_init(c0, None, coef)
    p64.iterate(iters)
    x64 = p64.wave()
    sc64, _ = sc_lin_f64(x64[0], mag[0], hop, window, dev)
    tol = 1e-5 if method == "griffin_lim" else 3e-3      # ADMM at rho = 0.1 is chaotic w.r.t. rounding
    out = {"what": "self-check (a tripwire, not parity evidence - parity is tests/ against the reference's fixtures): SC_lin of "
                   "item 0 (float64 evaluation) vs a float64 re-run of the same iterations on this library's generic kernels",
           "sc_lin": sc32, "sc_lin_ref": sc64, "abs_diff": abs(sc32 - sc64), "tol": tol, "ok": bool(abs(sc32 - sc64) <= tol)}
    ref = reference_trace_check(leg)
    if ref is not None:
        out["reference"] = ref
        out["ok"] = bool(out["ok"] and ref["ok"])
    return out


def reference_trace_check(leg):
    """C2 on rank 0's input is exactly what tests/golden/g16b_c2_headline.npz holds the UNMODIFIED REFERENCE's run of
    (torch_specinv/methods.py:193-270, B = 64, 100 iterations, alpha 0.3, eva_iter 10; tests/golden/make_golden.py:g16): the ten
    whole-batch evaluations of the last timed step against the reference's, |dSC_lin| <= 1e-5 (the north-star bar)."""
    if leg.workload != "C2" or leg.batch != 64 or leg.rank != 0 or leg.o.generic or "evals" not in leg.state:
        return None
    path = os.path.join(ROOT, "tests", "golden", "g16b_c2_headline.npz")
    if not os.path.exists(path):
        return None
    g = np.load(path)
    want, sc_ref = g["trace"], g["sc_db_from_loss"]
    got = np.array([[m, l] for _, m, l in leg.state["evals"]])
    if got.shape != want.shape:
        return None
    # The reference's loss column (F.mse_loss) is good to 1e-7; its SC column is not (torch's float32 `norm` over 6.7e7 elements is
    # 4e-3 off for ||target|| alone), so the spectral convergence is compared with what the reference's loss and the exact
    # ||target|| imply (tests/golden/make_golden.py:g16)
    d = np.abs(10.0 ** (got[:, 0] / 20.0) - 10.0 ** (sc_ref / 20.0))
    dl = np.abs(got[:, 1] / want[:, 1] - 1.0)
    return {"what": "the ten whole-batch evaluations of the last step vs the unmodified reference's run of this configuration and "
                    "input (tests/golden/g16b_c2_headline.npz): loss (F.mse_loss) relative difference, and |dSC_lin| against the "
                    "spectral convergence the reference's loss implies (its own SC column carries a 1.5e-3 float32-norm error)",
            "max_rel_dloss": float(dl.max()), "max_abs_dsc_lin": float(d.max()), "sc_db_final": float(got[-1, 0]),
            "sc_db_final_reference_from_loss": float(sc_ref[-1]), "sc_db_final_reference_reported": float(want[-1, 0]),
            "tol": 1e-5, "ok": bool(d.max() <= 1e-5 and dl.max() <= 1e-5)}


# ---------------------------------------------------------------------------------------------------------------------------
# The ONE line the driver keeps a tail of: everything a reader needs to recompute the claims, every string once (`legend`), the
# short legs as rows of one table; the verbose record (every `what` / `formula`, raw counters per leg) goes to a side file.
LEG_COLS = ["value", "unit", "ms_per_step", "steps", "launch_ms", "roofline_frac_hbm", "traffic_bytes", "limiter", "limiter_frac",
            "lds_conflict_frac", "check_ok", "note"]
LEGEND = {
    "roofline": "dominant kernel: achieved = bytes_per_unit x units_per_launch (the bytes the kernel HAS to move, DESIGN 3) / launch_ms "
                "(HIP events on the launch stream, timed region); peak 8 TB/s; traffic = (2 FETCH_SIZE + WRITE_SIZE) x 1024 B per "
                "launch (gfx950 correction, separate rocprofv3 --pmc passes over a child of this run); limiter = what binds the kernel: "
                "valu -> SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024) issue slots in use, hbm -> traffic / launch_ms / 8 TB/s, "
                "latency -> dependent steps/s; reference_bytes_frac = SURVEY 8d's bytes of the REFERENCE algorithm over the same time "
                "(a speed-up over that algorithm, may exceed 1)",
    "legs": "short legs of the other BASELINE workloads in this process after the headline's timed region, same timing rules; columns "
            "in legs_cols; note = evaluations | pairs accepted/rejected | who decides (C5), dependent steps/s (C3), objective form",
    "check": "tripwires outside the timed region (parity is tests/): SC_lin of item 0 vs a float64 re-run; reference = the ten "
             "whole-batch evaluations of the last step vs the unmodified reference's run (tests/golden/g16b_c2_headline.npz)",
    "h2d": "value_incl_h2d: the 268 MB target staged from pinned host memory every step, double-buffered on a second stream (serial: "
           "copy in front of every step)",
}


def _r(v, sig=6):
    if isinstance(v, float):
        return float(f"{v:.{sig}g}")
    if isinstance(v, dict):
        return {k: _r(x, sig) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_r(x, sig) for x in v]
    return v


def _short_workload(w):
    return (w.replace("batch=", "B").replace("n_frames=", "T=").replace("maxiter=", "it=").replace("look_ahead=", "LA=")
             .replace("asymmetric_window=", "asym=").replace("(optimizer.step calls) x ", ""))[:118]


def _compact_roof(r):
    keep = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_frac", "launch_ms", "launches_timed", "bytes_per_unit",
            "units_per_launch", "limiter", "limiter_frac", "valu_issue_frac", "valu_share_of_wave_life", "lds_conflict_frac",
            "launches_counted", "dependent_steps_per_s", "csrc_sha1", "traffic_stale")
    out = {k: r[k] for k in keep if r.get(k) is not None}
    out["kernel"] = r.get("dominant_kernel_symbol") or r.get("kernel")
    src = r.get("traffic_source") or ""
    out["traffic_source"] = "live" if src.startswith("measured in this run") else src[:80]
    if r.get("hbm", {}).get("traffic_over_algorithmic") is not None:
        out["traffic_over_algorithmic"] = r["hbm"]["traffic_over_algorithmic"]
    if r.get("reference_bytes_equiv"):
        out["reference_bytes_frac"] = r["reference_bytes_equiv"]["over_hbm_peak"]
    if r.get("counters_mean_per_launch"):
        out["counters"] = r["counters_mean_per_launch"]
    return out


def _leg_row(e):
    if "error" in e:
        return {"error": e["error"][:160]}
    r = e.get("roofline", {})
    lb = e.get("lbfgs")
    note = None
    if lb:
        note = (f"{lb['evaluations']} ev | {lb['pairs_accepted']}/{lb['pairs_rejected']} pairs | hist {lb['history_len']} | "
                f"{'device' if lb['decisions'].startswith('on the device') else 'host'}-decided | objective "
                f"{e.get('objective_kind')}")
    elif r.get("dependent_steps_per_s"):
        note = f"{r['dependent_steps_per_s']:.0f} dependent steps/s"
    ck = e.get("check") or {}
    return [e["value"], e["unit"], e["ms_per_step"], e["steps"], e.get("launch_ms"), r.get("frac"), r.get("traffic"), r.get("limiter"),
            r.get("limiter_frac"), r.get("lds_conflict_frac"), ck.get("ok"), note]


def compact_line(full):
    out = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data") if k in full}
    cfg = dict(full["config"])
    cfg["workload"] = _short_workload(cfg["workload"])
    geo = cfg.pop("launch_geometry", None)
    if geo:
        cfg["kernel"] = geo.get("kernel")
    out["config"] = cfg
    for k in ("value_incl_h2d", "ms_per_step_incl_h2d"):
        if k in full:
            out[k] = full[k]
    if full.get("h2d"):
        out["h2d_serial"] = [full["h2d"]["serial"]["value"], full["h2d"]["serial"]["ms_per_step"]]
    out["roofline"] = _compact_roof(full["roofline"])
    if full.get("cpu_baseline"):
        out["cpu_baseline"] = full["cpu_baseline"]
    if full.get("multi_gpu"):
        out["multi_gpu"] = full["multi_gpu"]
    ck = full.get("check")
    if ck:
        out["check"] = {k: ck[k] for k in ("ok", "sc_lin", "sc_lin_ref", "abs_diff", "tol", "loss", "mse") if k in ck}
        if ck.get("reference"):
            out["check"]["reference"] = {k: v for k, v in ck["reference"].items() if k != "what"}
    legs = (full.get("extra") or {}).get("workloads")
    if legs:
        out["legs_cols"] = LEG_COLS
        out["legs"] = {k: _leg_row(e) for k, e in legs.items()}
        out["legs_workloads"] = {k: _short_workload(e["workload"]) for k, e in legs.items() if "workload" in e}
    out["legend"] = LEGEND
    for k in ("bench_seconds", "detail"):
        if k in full:
            out[k] = full[k]
    return _r(out)


def write_detail(full):
    """The verbose record beside the line: gpurun_out/bench_detail.json (or $SPECINV_BENCH_DETAIL); best effort."""
    path = os.environ.get("SPECINV_BENCH_DETAIL") or os.path.join(ROOT, "gpurun_out", "bench_detail.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as fh:
            json.dump(full, fh)
        return os.path.relpath(path, ROOT)
    except OSError:
        return None


# ---------------------------------------------------------------------------------------------------------------------------
EXTRA_LEGS = (          # (key, workload, steps, warmup, option overrides, environment) - short legs after the headline's timed region
    ("C4", "C4", 5, 1, {}, {}),
    ("C3", "C3", 2, 1, {}, {}),
    ("C5", "C5", 2, 1, {"outer": 50, "c5_variant": "baseline"}, {}),
    ("C5_mfma", "C5", 2, 1, {"outer": 50, "c5_variant": "baseline"}, {"SPECINV_OBJ_SPARSE": "0"}),   # the filterbank on the matrix cores
    ("C5_wolfe", "C5", 2, 1, {"outer": 50, "c5_variant": "wolfe"}, {}),
    ("C5_memory", "C5", 1, 1, {"outer": 8, "c5_variant": "memory"}, {}),                             # the memory fills: the recursion carries weight
    ("C1", "C1", 50, 5, {}, {}),
    ("F64", "F64", 3, 1, {}, {}),                                                                    # coverage path, float64 (k_wave_iter + k_ola)
    ("S32", "S32", 3, 1, {}, {}),                                                                    # coverage path, n_fft 256 (k_wave_iter, register overlap-add)
    ("W400", "W400", 3, 1, {}, {}),                                                                  # coverage path, n_fft 400 / hop 160 (k_wave_iter, LDS ring)
)


def leg_options(args, **over):
    o = argparse.Namespace(**vars(args))
    o.batch, o.generic = None, False
    for k, v in over.items():
        setattr(o, k, v)
    return o


def run_extra(args, dev, pmc_all, pmc_source):
    from spectrogram_inversion_amd.plan import clear_plan_cache
    out = {}
    for key, workload, steps, warmup, over, env in EXTRA_LEGS:
        t0 = time.perf_counter()
        saved = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            leg = Leg(workload, leg_options(args, **over), 0, 1, dev, "nccl")
            el, _ = leg.run(steps, warmup)
            leg.freeze()
            entry = {"workload": f"{workload}: {leg.describe()}", "value": leg.units(steps) / el, "unit": leg.unit, "steps": steps,
                     "warmup": warmup, "ms_per_step": 1e3 * el / steps}
            pmc = pmc_all.get(workload)
            src = pmc_source
            stale = sha = None
            if key != workload:                               # a variant of the workload: the counted kernel is not the one that ran
                pmc, src = {}, "not counted (variant leg)"
            elif not pmc:
                pmc, sha, stale = stored_pmc(workload)
                src = "stored: profiles/traffic.json (rocprofv3 --pmc summary), not measured in this run"
            roof = leg.roofline(pmc, src, stale, sha)
            entry["launch_ms"] = roof["launch_ms"]
            entry["frac"] = roof.get("frac")
            entry["roofline"] = {k: roof[k] for k in ("bound", "limiter", "limiter_frac", "kernel", "frac", "achieved", "peak", "unit",
                                                       "traffic", "traffic_frac", "traffic_source", "hbm", "valu_issue_frac",
                                                       "lds_conflict_frac", "dependent_steps_per_s", "evaluations_timed",
                                                       "bytes_per_unit", "units_per_launch") if k in roof}
            obj = getattr(getattr(leg, "fg_raw", None), "device_objective", None)
            entry["objective_kind"] = obj[0].objective_kind if obj else None
            if leg.method == "L_BFGS":
                entry["lbfgs"] = leg.lbfgs_info()
            if not args.no_check:
                entry["check"] = check(leg, leg.state["x"])
            entry["leg_seconds"] = time.perf_counter() - t0
            out[key] = entry
            del leg
        except Exception as e:                                # an extra leg must never take the headline line down with it
            out[key] = {"error": f"{type(e).__name__}: {e}"[:400]}
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        clear_plan_cache()
        torch.cuda.empty_cache()
    return out


def pmc_child(workloads):
    """The process rocprofv3 wraps (`live_pmc`): one warm-up + one counted step of every workload, nothing printed."""
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ap = build_parser()
    base = ap.parse_args([])
    for w in workloads:
        o = leg_options(base, outer=2) if w == "C5" else leg_options(base)
        leg = Leg(w, o, 0, 1, dev, "nccl")
        leg.run(1, 1)
        del leg
        from spectrogram_inversion_amd.plan import clear_plan_cache
        clear_plan_cache()
    torch.cuda.synchronize()


def spawn_ranks(args):
    """`python3 bench.py --gpus N` without a launcher: start N ranks under torch.distributed.run as a CHILD process - before this
    process has made any GPU call - and leave with its exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="override the per-GPU batch")
    ap.add_argument("--asym", action="store_true", help="C3: asymmetric_window=True")
    ap.add_argument("--outer", type=int, default=50, help="C5: optimizer.step calls per bench step (BASELINE configs[4]: maxiter=50)")
    ap.add_argument("--c5-variant", default="baseline", choices=["baseline", "main", "wolfe", "memory"],
                    help="C5 optimiser options: torch.optim.LBFGS defaults (BASELINE); the reference demo's (main.py:43: max_iter 50, "
                         "history 10); defaults + line_search_fn='strong_wolfe' (steps long enough for the curvature pairs to pass "
                         "y.s > 1e-10; its default tolerances end most steps after one iteration on this input); 'memory': strong "
                         "Wolfe with both tolerances at 0, so that every step runs its 20 iterations, the memory fills to "
                         "history_size = 100 and the recursion's passes over 200 vectors carry weight - use --outer 8 or more")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the C4 / C3 / C5 legs that follow the default C2 run")
    ap.add_argument("--no-pmc", action="store_true", help="no live rocprofv3 --pmc passes: counters from profiles/traffic.json")
    ap.add_argument("--no-h2d", action="store_true", help="skip the H2D-inclusive legs")
    ap.add_argument("--no-overlap-gather", action="store_true",
                    help="N > 1: wait for each step's RCCL gather before the next step starts (default: it overlaps the next step)")
    ap.add_argument("--gather-kernel-budget", type=int, default=None, metavar="K",
                    help="N > 1: plan the iteration launches for 256 - K compute units (SPECINV_CU_BUDGET) so that the RCCL kernels of "
                         "a gather that overlaps the next step find free CUs instead of pushing iteration workgroups into a second "
                         "round; default 16 over RCCL (NCCL_MAX_NCHANNELS is capped to 8 with it), 0 at N = 1 and in the gloo rehearsal")
    ap.add_argument("--even-chunks", action="store_true",
                    help="even chunks of frames per wave instead of the skewed ones (DESIGN 3.2 (7)): to try when RCCL's kernels share "
                         "the chip with an exactly-full iteration launch")
    ap.add_argument("--keep-pmc", default=None, help="directory to keep the raw per-dispatch counter CSVs of the live passes in")
    ap.add_argument("--generic", action="store_true", help="force the generic (unfused) kernels")
    ap.add_argument("--verbose", action="store_true", help="print the verbose record (every what / formula string, raw counters per "
                                                           "leg) instead of the compact line; it is always written to "
                                                           "gpurun_out/bench_detail.json")
    ap.add_argument("--pmc-child", default=None, help=argparse.SUPPRESS)
    return ap


def main():
    args = build_parser().parse_args()
    if args.even_chunks:
        os.environ["SPECINV_TD_SKEW"] = "0"
        os.environ["SPECINV_K4_SKEW"] = "0,0"
    if args.pmc_child:
        return pmc_child(args.pmc_child.split(","))
    # N > 1 over RCCL: leave compute units to the gather's kernels (see --gather-kernel-budget); decided before anything is planned
    # and before the communicator exists (NCCL_MAX_NCHANNELS is read at its creation)
    n_ranks = int(os.environ.get("WORLD_SIZE", str(args.gpus)))
    budget = args.gather_kernel_budget
    if budget is None:
        budget = 16 if (n_ranks > 1 and os.environ.get("SPECINV_BENCH_BACKEND", "nccl") == "nccl") else 0
    budget = max(0, min(128, budget)) if n_ranks > 1 else 0
    if budget:
        os.environ["SPECINV_CU_BUDGET"] = str(budget)
        os.environ.setdefault("NCCL_MAX_NCHANNELS", str(max(1, budget // 2)))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # SPECINV_BENCH_BACKEND=gloo (development only): rehearse the N > 1 code path of this script - sharding, barrier, gather,
    # max over ranks - with several ranks on the ONE GPU of a test box (a device cannot host two ranks of an RCCL communicator);
    # the gather is then staged through host memory, so the line it prints is not a measurement
    backend = os.environ.get("SPECINV_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    t_begin = time.perf_counter()
    leg = Leg(args.workload, args, rank, world, dev, backend)
    method, batch = leg.method, leg.batch
    elapsed_local, gathered = leg.run(args.steps, args.warmup)
    elapsed = elapsed_local
    x = gathered if (world > 1 and method != "L_BFGS") else leg.state["x"]
    diag = None
    if world > 1:
        where = dev if backend == "nccl" else torch.device("cpu")
        tt = torch.tensor([elapsed_local], dtype=torch.float64, device=where)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        # per-rank diagnostics: every rank's own wall time and the host time it spent waiting on gathers; ranks RCCL really saw
        mine = torch.tensor([elapsed_local, leg.gather_wait_s, 1.0], dtype=torch.float64, device=where)
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        # one blocking gather after the timed region, nothing else on the chip: what the exchange itself costs
        gather_ms = None
        if method != "L_BFGS":
            from spectrogram_inversion_amd.distributed import gather_waveforms
            xl = leg.state["x"]
            gather_waveforms(xl, dst=0, sizes=[batch] * world)
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            gather_waveforms(xl, dst=0, sizes=[batch] * world)
            torch.cuda.synchronize()
            gather_ms = 1e3 * (time.perf_counter() - t0)
        diag = {"per_rank_ms_per_step": [1e3 * float(t[0]) / args.steps for t in every],
                "per_rank_gather_wait_ms_per_step": [1e3 * float(t[1]) / args.steps for t in every],
                "ranks_seen": int(sum(float(t[2]) for t in every)), "backend": backend,
                "gather": "overlapped with the next step (RCCL stream)" if not args.no_overlap_gather else "blocking after each step",
                "gather_alone_ms_rank0": gather_ms,
                "gather_kernel_budget_cus": budget, "nccl_max_nchannels": os.environ.get("NCCL_MAX_NCHANNELS"),
                "gather_bytes_per_rank": None if method == "L_BFGS" else batch * leg.length * 4}
    if rank == 0:
        want = (batch * world if method != "L_BFGS" else batch, leg.length)
        assert tuple(x.shape) == want, (tuple(x.shape), want)

    if rank == 0:
        path, geo, kernel = leg.kernel_info()
        out = {
            "metric": "Griffin-Lim iterations*frames/sec at n_fft=2048 hop=512" if args.workload == "C2"
                      else f"{method} {leg.unit.split('/')[0]}/sec ({args.workload})",
            "value": leg.units(args.steps) / elapsed,
            "unit": leg.unit,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {leg.describe()}",
                       "global_batch": batch * world,
                       "parallelism": (f"batch-sharded x{world}, RCCL gather" if method != "L_BFGS"
                                       else f"replicas x{world} (one optimisation problem per GPU)"),
                       "kernel_path": path, "launch_geometry": geo,
                       "step": {"griffin_lim": "phase_init + ISTFT + iterations + gather",
                                "ADMM": "phase_init + ISTFT + iterations + gather",
                                "RTISI_LA": "persistent RTISI-LA launch + overlap-add + gather",
                                "L_BFGS": "optimizer.step calls (objective evaluations + two-loop recursion)"}[method]},
            "roofline": None,
        }
        if diag:
            out["multi_gpu"] = diag
        if method == "L_BFGS":
            out["config"]["lbfgs"] = leg.lbfgs_info()
        leg.freeze()                       # the launch time belongs to the timed region: later legs reuse the event lists
        primary_state = dict(leg.state)
        solo = world == 1 and args.batch is None and not args.generic
        # ---- counters: live passes over a child process (N = 1), else the stored summary
        want_extra = solo and args.workload == "C2" and not args.no_extra
        pmc_all, pmc_source = {}, None
        if solo and not args.no_pmc and args.workload in DOMINANT:
            wl = [args.workload] + ([w for w in ("C4", "C3", "C5") if want_extra])
            pmc_all, pmc_source = live_pmc(wl, keep=args.keep_pmc)
        pmc, stale, sha = pmc_all.get(args.workload), None, None
        if not pmc:
            why = pmc_source
            pmc, sha, stale = stored_pmc(args.workload, generic=(path == "generic"))
            pmc_source = "stored: profiles/traffic.json (rocprofv3 --pmc summary), not measured in this run" + \
                         (f" [live passes: {why}]" if why else "")
        out["roofline"] = leg.roofline(pmc, pmc_source, stale, sha)
        if not args.no_check:
            out["check"] = check(leg, x)
        # ---- the same step with its input staged over PCIe
        if world == 1 and not args.no_h2d and method != "L_BFGS":
            h = leg.h2d_legs(max(3, min(args.steps, 10)), 1)
            if h:
                out["value_incl_h2d"] = h["pipelined"]["value"]
                out["ms_per_step_incl_h2d"] = h["pipelined"]["ms_per_step"]
                out["h2d"] = h
        leg.state.update(primary_state)
        if want_extra:
            del leg
            from spectrogram_inversion_amd.plan import clear_plan_cache
            clear_plan_cache()
            torch.cuda.empty_cache()
            out["extra"] = {"workloads": run_extra(args, dev, pmc_all, pmc_source),
                            "what": "short legs of the other BASELINE workloads, run in this process after the headline's timed "
                                    "region (same timing rules: warm-up, barrier + synchronize on both sides)"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(method, WORKLOADS[args.workload][2], WORKLOADS[args.workload][3],
                                               WORKLOADS[args.workload][4],
                                               WORKLOADS[args.workload][6] if WORKLOADS[args.workload][6] is not None else 0.0)
        out["bench_seconds"] = time.perf_counter() - t_begin
        detail = write_detail(out)
        if detail:
            out["detail"] = detail
        print(json.dumps(out if args.verbose else compact_line(out)), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
