#!/usr/bin/env python3
"""What does the reference's projection `spec * target / (spec.abs() + 1e-16)` (torch_specinv/methods.py:246-247) execute, operation
by operation, on the CPU path (torch, this container) - and how close do the candidate device formulas come to it?

Findings (torch 2.10.0 CPU, AVX512 build; 2^22 random bins):
  * spec.abs() is hypotf: correctly rounded sqrt(x^2 + y^2) of the EXACT sum of squares;
  * complex / real is a multiplication by the correctly rounded reciprocal of the divisor, not a division:
    out = (s * m) * RN(1 / (|s| + 1e-16))   [bit-identical on every sample; a true division matches 74 %]
  * y / envelope (methods.py:132, real tensors) is an IEEE division.
Candidates (emulated in float64 -> float32 here; the kernels' forms are in csrc/fast_core.h):
  A  r02/r03 "exact" build:  RN(x m / (RN sqrt(fma(x,x,y y)) + 1e-16))             (IEEE sqrt + division of the product)
  B  SPECINV_REFCHAIN=1:     (x m) * RN(1 / (RN sqrt(t) + 1e-16)),  t = fma(x,x,fl(y y))
  C  SPECINV_REFCHAIN=2:     (x m) * RN(t^-1/2)                       one rounding of the exact inverse root
  D  default approximations:  x * (m * rsq(t)),  rsq good to 1 ulp (emulated as the correctly rounded value: a best case)
"""
import numpy as np
import torch

rng = np.random.default_rng(0)
n = 1 << 22
f32 = np.float32
x = (rng.standard_normal(n) * rng.random(n)).astype(f32)
y = rng.standard_normal(n).astype(f32)
m = rng.random(n).astype(f32)

s = torch.from_numpy(x + 1j * y.astype(np.complex64)).to(torch.complex64)
out_t = (s * torch.from_numpy(m) / (s.abs() + 1e-16)).numpy()
ref = out_t.real.copy()

hyp = np.hypot(x, y)
r_ref = (f32(1) / (hyp + f32(1e-16))).astype(f32)
chain = ((x * m).astype(f32) * r_ref).astype(f32)
print(f"torch == (x m) * RN(1 / (hypotf + 1e-16)): {(chain == ref).mean():.4f}   |s|: hypotf == torch.abs {(hyp == s.abs().numpy()).mean():.4f}")
print(f"torch == RN(x m / (hypotf + 1e-16)) (true division): {(((x * m).astype(f32) / (hyp + f32(1e-16))).astype(f32) == ref).mean():.4f}")

t = (x.astype(np.float64) * x.astype(np.float64) + (y * y).astype(f32).astype(np.float64)).astype(f32)
sq = np.sqrt(t)                                   # correctly rounded sqrt of the rounded sum
A = ((x * m).astype(f32) / (sq + f32(1e-16))).astype(f32)
B = ((x * m).astype(f32) * (f32(1) / (sq + f32(1e-16))).astype(f32)).astype(f32)
C = ((x * m).astype(f32) * (1.0 / np.sqrt(t.astype(np.float64))).astype(f32)).astype(f32)
D = (x * (m * (1.0 / np.sqrt(t.astype(np.float64))).astype(f32)).astype(f32)).astype(f32)


def ulp(a, b):
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


exact = x.astype(np.float64) * m.astype(np.float64) / np.sqrt(x.astype(np.float64) ** 2 + y.astype(np.float64) ** 2)
print(f"{'':28s} bit-identical   mean ulp   rms rel. deviation from the reference's value | from the exact value")
for name, o in (("reference chain itself", ref), ("A  IEEE sqrt + division", A), ("B  REFCHAIN=1", B), ("C  REFCHAIN=2", C), ("D  default (ideal rsq)", D)):
    dev = np.sqrt(np.mean(((o.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1e-30)) ** 2))
    dex = np.sqrt(np.mean(((o.astype(np.float64) - exact) / np.maximum(np.abs(exact), 1e-30)) ** 2))
    print(f"{name:28s} {(o == ref).mean():10.4f} {ulp(o, ref).mean():10.3f}   {dev:.2e} | {dex:.2e}")
