#!/usr/bin/env python3
"""Does IEEE arithmetic in the projection restore the strict 100-iteration waveform gate, and what does it cost?

Runs the default build (v_rcp_f32 / v_sqrt_f32 approximations, multiplication by 1/envelope) and the -DSPECINV_IEEE=1
build (correctly rounded sqrt and divisions, true division by the envelope: the reference's operations,
torch_specinv/methods.py:132,246-247) of libspecinv on the same box, each in its own process (SPECINV_LIB), and prints
for both: the g2 fixture (random, inconsistent magnitudes) and the g14 fixture (well-conditioned) against the
reference's float32 and float64 waveforms at 1 / 10 / 100 iterations, the strict gate min(1e-4, 6 x noise), and the
time per launch of the headline kernel (BASELINE C2).

    python tools/ieee_study.py            # parent: runs both builds, writes gpurun_out/r02_ieee_study.txt
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
    from spectrogram_inversion_amd.plan import Plan, args_helper
    dev = torch.device("cuda", 0)
    print(f"== build: {os.environ.get('SPECINV_TAG')}  ({os.environ.get('SPECINV_LIB', 'default library')})")
    for paths in ("frame kernel (k_semi)", "fused kernel (k_fused<4,4>)"):
        if paths.startswith("fused"):
            os.environ["SPECINV_SMALL_FRAMES"] = "0"
        print(f"-- {paths}")
        g = load_golden("g2_gla")
        hop, w = int(g["hop"]), torch.from_numpy(g["window"])
        init = torch.from_numpy(g["init"]).to(dev)
        print("   g2 (random magnitudes):   alpha  it   vs ref32    vs ref64    ref32-ref64  strict gate  q75 segment  max segment")
        for alpha in (0.0, 0.3, 0.99):
            for it in (1, 10, 100):
                p = Plan(args_helper(init, hop_length=hop, window=w), init.shape[0], init.shape[2], torch.float32, dev)
                p.gla_init(init, None, alpha)
                p.run(it, 10, 0.0, "sc")
                y = p.wave().cpu().numpy()
                ref, ref64 = g[f"wave_a{alpha}_it{it}"], g[f"wave64_a{alpha}_it{it}"]
                noise = rel_l2(ref, ref64)
                gate = min(1e-4, max(6 * noise, 3e-6))
                seg = segment_errors(y, ref, hop)
                print(f"                              {alpha:4.2f} {it:4d}   {rel_l2(y, ref):.2e}   {rel_l2(y, ref64):.2e}   {noise:.2e}     "
                      f"{'pass' if rel_l2(y, ref) < gate else 'FAIL'} ({gate:.1e})  {np.quantile(seg, 0.75):.2e}    {seg.max():.2e}")
        g = load_golden("g14_wellcond")
        hop, w = int(g["hop"]), torch.from_numpy(g["window"])
        init = torch.from_numpy(g["init"]).to(dev)
        print("   g14 (well-conditioned), 100 iterations:")
        for alpha in (0.0, 0.3, 0.99):
            p = Plan(args_helper(init, hop_length=hop, window=w), init.shape[0], init.shape[2], torch.float32, dev)
            p.gla_init(init, None, alpha)
            p.run(100, 10, 0.0, "sc")
            y = p.wave().cpu().numpy()
            ref, ref64 = g[f"wave_a{alpha}"], g[f"wave64_a{alpha}"]
            noise = rel_l2(ref, ref64)
            gate = min(1e-4, max(6 * noise, 3e-6))
            print(f"                              {alpha:4.2f}  100   {rel_l2(y, ref):.2e}   {rel_l2(y, ref64):.2e}   {noise:.2e}     "
                  f"{'pass' if rel_l2(y, ref) < gate else 'FAIL'} ({gate:.1e})")
    os.environ.pop("SPECINV_SMALL_FRAMES", None)
    # cost on the headline shape
    rng = np.random.default_rng(1234)
    for tag, (b, n_fft, hop, frames, admm) in {"C2 k_fused4<16,GLA>": (64, 2048, 512, 1024, False),
                                               "C4 k_fused4<8,ADMM>": (32, 1024, 256, 2048, True)}.items():
        mag = torch.from_numpy(rng.random((b, n_fft // 2 + 1, frames), dtype=np.float32)).to(dev)
        win = torch.from_numpy((0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)).astype(np.float32))
        p = Plan(args_helper(mag, hop_length=hop, window=win), b, frames, torch.float32, dev)
        (p.admm_init if admm else p.gla_init)(None, mag, 0.1 if admm else 0.3)
        p.iterate(20)
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            p.iterate(50)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 50)
        print(f"   {tag}: {best:.4f} ms per launch (best of 5 x 50)")
        del p


def main():
    if os.environ.get("SPECINV_TAG"):
        return child()
    out = []
    variants = [("default", None), ("IEEE (-DSPECINV_IEEE=1)", os.path.join(ROOT, "spectrogram_inversion_amd", "variants", "libspecinv_ieee.so"))]
    for rnd in range(2):                                    # the timing part twice, interleaved (same box, same process order)
        for tag, lib in variants:
            env = dict(os.environ, SPECINV_TAG=f"{tag}, round {rnd}")
            if lib:
                if not os.path.exists(lib):
                    raise SystemExit(f"{lib} missing: python -c \"from spectrogram_inversion_amd.build import build_lib; "
                                     f"build_lib(extra_flags=['-DSPECINV_IEEE=1'], out='{lib}')\"")
                env["SPECINV_LIB"] = lib
            r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
            out.append(r.stdout + (r.stderr[-2000:] if r.returncode else ""))
            print(out[-1], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r02_ieee_study.txt"), "w") as fh:
        fh.write("\n".join(out))


if __name__ == "__main__":
    main()
