"""Long runs on the well-conditioned fixture g15 (n_fft 1024 / hop 256): does carrying the momentum as a signal drift over
hundreds of iterations?  rel-L2 of the float32 kernels against the float64 generic kernels from the same start."""
import sys, os
os.environ["SPECINV_SMALL_FRAMES"] = "0"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectrogram_inversion_amd.plan import Plan, args_helper
dev = torch.device("cuda", 0)
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g15_wellcond_1024.npz"))
init = torch.from_numpy(g["init"]).to(dev)
hop, w = int(g["hop"]), torch.from_numpy(g["window"])
def rel(a, b): return float(np.linalg.norm(a - b) / np.linalg.norm(b))
for alpha in (0.3, 0.99):
    for its in (100, 300, 1000):
        ys = {}
        for name in ("td", "spec", "f64"):
            dt = torch.float64 if name == "f64" else torch.float32
            p = Plan(args_helper(init, hop_length=hop, window=w.to(dt)), init.shape[0], init.shape[2], dt, dev)
            p.keep_state(name == "spec")
            p.gla_init(init.to(torch.complex128 if name == "f64" else torch.complex64), None, alpha)
            p.iterate(its)
            ys[name] = p.wave().double().cpu().numpy()
        print(f"alpha {alpha} its {its:4d}: td-f64 {rel(ys['td'], ys['f64']):.2e}  spec-f64 {rel(ys['spec'], ys['f64']):.2e}  td-spec {rel(ys['td'], ys['spec']):.2e}", flush=True)
