import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import spectrogram_inversion_amd as si
from _util import load_golden, rel_l2
g = load_golden("g2_gla")
kw = dict(hop_length=int(g["hop"]), window=torch.from_numpy(g["window"]))
init = torch.from_numpy(g["init"]).cuda()
print("lib", os.environ.get("SPECINV_LIB", "default"), "n_fft", (init.shape[1] - 1) * 2, "hop", int(g["hop"]), "T", init.shape[2])
for alpha in (0.0, 0.3, 0.99):
    for it in (1, 10, 100):
        key = f"a{alpha}_it{it}"
        y = si.griffin_lim(init, max_iter=it, alpha=alpha, tol=0, verbose=False, eva_iter=10, **kw).cpu().numpy()
        ref, ref64 = g["wave_" + key], g["wave64_" + key]
        print(f"  alpha {alpha} it {it:3d}: vs ref32 {rel_l2(y, ref):.2e}  vs ref64 {rel_l2(y, ref64):.2e}  ref32 vs ref64 {rel_l2(ref, ref64):.2e}")
print("--- trajectory of the alpha=0.3 case vs the float64 GPU path")
from spectrogram_inversion_amd.plan import Plan, args_helper
dev = torch.device("cuda", 0)
a32 = args_helper(init, **kw)
p32 = Plan(a32, init.shape[0], init.shape[2], torch.float32, dev)
kw64 = dict(hop_length=int(g["hop"]), window=torch.from_numpy(g["window"]).double())
p64 = Plan(args_helper(init.to(torch.complex128), **kw64), init.shape[0], init.shape[2], torch.float64, dev)
p32.gla_init(init, None, 0.3); p64.gla_init(init.to(torch.complex128), None, 0.3)
hop = int(g["hop"])
for it in range(1, 131):
    p32.iterate(1); p64.iterate(1)
    if it % 10 == 0 or it in (95, 98, 99, 101, 102, 105):
        y, z = p32.wave().double(), p64.wave()
        e = (y - z)
        seg = e[:, : (e.shape[1] // hop) * hop].reshape(e.shape[0], -1, hop).pow(2).sum(-1)
        top = torch.topk(seg.flatten(), 3)
        print(f"  it {it:3d} rel {float(e.norm() / z.norm()):.2e}  top-3 hop segments hold {float(top.values.sum() / seg.sum()):.2f} of the error energy at {top.indices.tolist()}")
