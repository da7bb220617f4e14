"""k_fused4_td (momentum carried as a signal) against k_fused4 (pre_spec) and float64, on a noise-signal spectrogram.
Prints rel-L2 distances after n iterations: which float32 kernel is closer to the float64 iteration?"""
import sys, os
os.environ["SPECINV_SMALL_FRAMES"] = "0"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectrogram_inversion_amd.plan import Plan, args_helper

dev = torch.device("cuda", 0)
def rel(a, b): return float(np.linalg.norm(a - b) / np.linalg.norm(b))
def hann(n): return (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n) / n)).astype(np.float32)

for n_fft, batch, frames in [(1024, 5, 333), (2048, 8, 512)]:
    hop = n_fft // 4
    rng = np.random.default_rng(n_fft + frames)
    sig = torch.from_numpy(rng.standard_normal((batch, (frames - 1) * hop)).astype(np.float32)).to(dev)
    w = torch.from_numpy(hann(n_fft))
    probe = torch.empty((1, n_fft // 2 + 1, 1))
    for alpha in (0.0, 0.3, 0.99):
        for its in (10, 20, 40, 100):
            ys = {}
            for name in ("td", "spec", "f64"):
                dt = torch.float64 if name == "f64" else torch.float32
                p = Plan(args_helper(probe, hop_length=hop, window=w.to(dt)), batch, frames, dt, dev)
                if name == "td":
                    mag = p.stft(sig).abs()
                    c0 = p.phase_init(mag)
                p.keep_state(name == "spec")
                p.gla_init(c0.to(torch.complex128 if name == "f64" else torch.complex64), None, alpha)
                if name != "f64":
                    assert p.launch_geometry["kernel"] == ("k_fused4_td" if name == "td" else "k_fused4")
                p.iterate(its)
                ys[name] = p.wave().double().cpu().numpy()
                del p
            print(f"n_fft {n_fft} alpha {alpha} its {its:3d}: td-spec {rel(ys['td'], ys['spec']):.2e}  td-f64 {rel(ys['td'], ys['f64']):.2e}  "
                  f"spec-f64 {rel(ys['spec'], ys['f64']):.2e}", flush=True)
