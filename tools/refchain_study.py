#!/usr/bin/env python3
"""The reference's operation chain in the projection (fast_core.h: ref_rcp_abs2) against the default approximations: accuracy on the
g2 / g14 / g16a fixtures (strict gate min(1e-4, 6 x the reference's own float32-vs-float64 distance)) on every float32 kernel
path, and the cost on the headline shapes.  Each (library, exact) arm in its own process, timing arms interleaved.

    python tools/refchain_study.py [variant.so ...]      # the shipped library is always included; writes gpurun_out/r04_refchain.txt
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def child():
    import numpy as np
    import torch
    import spectrogram_inversion_amd as si                      # noqa: F401
    from _util import load_golden, rel_l2, segment_errors
    from spectrogram_inversion_amd.plan import Plan, args_helper, clear_plan_cache
    dev = torch.device("cuda", 0)
    exact = os.environ.get("STUDY_EXACT") == "1"
    print(f"== {os.environ.get('STUDY_TAG')}")
    if os.environ.get("STUDY_ACC") == "1":
        npass = ntot = 0
        worst = 0.0
        for path in ("frame", "fused", "fused_prespec"):
            os.environ["SPECINV_SMALL_FRAMES"] = "6144" if path == "frame" else "0"
            clear_plan_cache()
            for fx, key_fmt, its in (("g2_gla", "a{a}_it{it}", (10, 100)), ("g14_wellcond", "a{a}", (100,)), ("g15_wellcond_1024", "a{a}", (100,)), ("g16a_wellcond_2048", "a{a}", (100,))):
                try:
                    g = load_golden(fx)
                except OSError:
                    continue
                hop, w = int(g["hop"]), torch.from_numpy(g["window"])
                init = torch.from_numpy(g["init"]).to(dev)
                for alpha in (0.0, 0.3, 0.99):
                    for it in its:
                        key = key_fmt.format(a=alpha, it=it)
                        if "wave_" + key not in g.files:
                            continue
                        p = Plan(args_helper(init, hop_length=hop, window=w), init.shape[0], init.shape[2], torch.float32, dev)
                        p.set_exact(exact)
                        p.keep_state(path == "fused_prespec")
                        p.gla_init(init, None, alpha)
                        kern = p.launch_geometry["kernel"]
                        p.run(it, 10, 0.0, "sc")
                        y = p.wave().cpu().numpy()
                        ref, ref64 = g["wave_" + key], g["wave64_" + key]
                        noise = rel_l2(ref, ref64)
                        gate = min(1e-4, max(6 * noise, 3e-6))
                        err = rel_l2(y, ref)
                        seg = segment_errors(y, ref, hop)
                        ok = err < gate
                        npass += ok
                        ntot += 1
                        worst = max(worst, err / gate)
                        print(f"   {fx:20s} {path:14s} {kern:12s} a={alpha:4.2f} it={it:3d}  vs ref32 {err:.2e}  vs ref64 {rel_l2(y, ref64):.2e}  "
                              f"ref32-ref64 {noise:.2e}  gate {gate:.1e} {'pass' if ok else 'FAIL'}  q75 seg {np.quantile(seg, 0.75):.2e} max {seg.max():.2e}")
        print(f"   strict gate: {npass} of {ntot} pass; worst error / gate {worst:.2f}")
        os.environ.pop("SPECINV_SMALL_FRAMES", None)
        clear_plan_cache()
        return
    # cost on the headline shapes
    rng = np.random.default_rng(1234)
    for tag, (b, n_fft, hop, frames, admm) in {"C2 griffin_lim 2048/512": (64, 2048, 512, 1024, False),
                                               "C4 ADMM 1024/256": (32, 1024, 256, 2048, True)}.items():
        mag = torch.from_numpy(rng.random((b, n_fft // 2 + 1, frames), dtype=np.float32)).to(dev)
        win = torch.from_numpy((0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)).astype(np.float32))
        p = Plan(args_helper(mag, hop_length=hop, window=win), b, frames, torch.float32, dev)
        p.set_exact(exact)
        (p.admm_init if admm else p.gla_init)(None, mag, 0.1 if admm else 0.3)
        p.iterate(30)                                          # past the early (c0) launches
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            p.iterate(50)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 50)
        # a whole step as bench.py runs it
        steps = []
        for _ in range(4):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            (p.admm_init if admm else p.gla_init)(None, mag, 0.1 if admm else 0.3)
            p.run(200 if admm else 100, 10, 0.0, "sc")
            p.wave()
            e1.record()
            torch.cuda.synchronize()
            steps.append(e0.elapsed_time(e1))
        print(f"   {tag}: {best:.4f} ms per late launch (best of 5 x 50); step {min(steps):.2f} ms (best of 4)  [{p.launch_geometry['kernel']}]")
        del p


def main():
    if os.environ.get("STUDY_TAG"):
        return child()
    libs = [None] + sys.argv[1:]
    out = []

    def run(tag, lib, exact, acc):
        env = dict(os.environ, STUDY_TAG=tag, STUDY_EXACT="1" if exact else "0", STUDY_ACC="1" if acc else "0")
        if lib:
            env["SPECINV_LIB"] = os.path.abspath(lib)
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
        out.append(r.stdout + (r.stderr[-2000:] if r.returncode else ""))
        print(out[-1], flush=True)
    for lib in libs:
        for exact in (False, True):
            if lib and not exact:
                continue                                     # (a variant only differs in its reference-chain kernels)
            run(f"accuracy: {os.path.basename(lib) if lib else 'shipped library'}, set_exact({exact})", lib, exact, True)
    for rnd in range(3):
        for lib in libs:
            for exact in (False, True):
                if lib and not exact:
                    continue
                run(f"cost, round {rnd}: {os.path.basename(lib) if lib else 'shipped library'}, set_exact({exact})", lib, exact, False)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r04_refchain.txt"), "w") as fh:
        fh.write("\n".join(out))


if __name__ == "__main__":
    main()
