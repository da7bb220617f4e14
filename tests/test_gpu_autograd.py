"""Gradients of griffin_lim w.r.t. the input spectrogram: hand-written adjoints against the gradients torch
autograd gives for the reference (tests/golden/g10_autograd.npz), plus adjoint identities.  Needs an MI355X."""
import numpy as np
import pytest
import torch

from _util import hann, load_golden, rel_l2

pytestmark = pytest.mark.gpu

import spectrogram_inversion_amd as si                               # noqa: E402
from spectrogram_inversion_amd.plan import Plan, args_helper          # noqa: E402

DEV = torch.device("cuda", 0)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def N(t):
    return t.detach().cpu().numpy()


CASES = [("f32_hann", 128, 32, True, {}, 2e-4), ("f64_hann", 128, 32, True, {}, 1e-9),
         ("f64_rect_default", 128, None, False, {}, 1e-9), ("f64_const_pad", 64, 16, True, dict(pad_mode="constant"), 1e-9),
         ("f64_normalized", 64, 16, True, dict(normalized=True), 1e-9)]


@pytest.mark.parametrize("tag,n_fft,hop,use_hann,extra,tol", CASES)
def test_gradient_matches_reference_autograd(tag, n_fft, hop, use_hann, extra, tol):
    g = load_golden("g10_autograd")
    mag = g[f"mag_{tag}"]
    kw = dict(extra)
    if hop:
        kw["hop_length"] = hop
    if use_hann:
        kw["window"] = torch.from_numpy(hann(n_fft, mag.dtype.type))
    spec = T(mag).requires_grad_(True)
    y = si.griffin_lim(spec, max_iter=3, alpha=0.5, tol=0, verbose=False, **kw)
    assert y.requires_grad
    assert rel_l2(N(y), g[f"y_{tag}"]) < max(tol, 1e-5 if mag.dtype == np.float32 else 1e-11)
    (y * T(g[f"w_{tag}"])).sum().backward()
    assert spec.grad is not None and spec.grad.shape == spec.shape
    assert rel_l2(N(spec.grad), g[f"grad_{tag}"]) < tol, rel_l2(N(spec.grad), g[f"grad_{tag}"])


def test_gradient_complex_warm_start():
    g = load_golden("g10_autograd")
    spec = T(g["c_complex"]).requires_grad_(True)
    y = si.griffin_lim(spec, max_iter=2, alpha=0.3, tol=0, verbose=False, hop_length=32,
                       window=torch.from_numpy(hann(128, np.float64)))
    (y * T(g["w_complex"])).sum().backward()
    assert rel_l2(N(spec.grad), g["grad_complex"]) < 1e-9


def test_reference_test_pattern_backward():
    """test/test_griffin.py:53-66: spec.requires_grad, mse against the original signal, backward()."""
    g = load_golden("g10_autograd")
    x = T(g["x_ref_test"])
    sp = torch.stft(x, 256, return_complex=True).abs().requires_grad_(True)
    y = si.griffin_lim(sp, max_iter=2, verbose=False)
    torch.nn.functional.mse_loss(x[:y.shape[0]], y).backward()
    assert hasattr(sp, "grad") and rel_l2(N(sp.grad), g["grad_ref_test"]) < 5e-4


def test_no_grad_requested_keeps_the_fused_path():
    mag = torch.rand(1, 513, 16, device=DEV)
    y = si.griffin_lim(mag, max_iter=2, verbose=False, hop_length=256, window=torch.hann_window(1024))
    assert not y.requires_grad
    with torch.no_grad():
        y2 = si.griffin_lim(mag.clone().requires_grad_(True), max_iter=2, verbose=False, hop_length=256,
                            window=torch.hann_window(1024))
    assert not y2.requires_grad and torch.equal(y, y2)


@pytest.mark.parametrize("n_fft,hop,frames,dtype,kw", [(128, 32, 9, torch.float64, {}), (1024, 256, 12, torch.float32, {}),
                                                       (64, 20, 11, torch.float64, dict(pad_mode="circular")),
                                                       (64, 16, 9, torch.float64, dict(center=False)),
                                                       (64, 16, 9, torch.float64, dict(onesided=False, normalized=True))])
def test_adjoint_identities(n_fft, hop, frames, dtype, kw):
    """<A x, Y> = <x, A^T Y> for the STFT and <B Q, g> = <Q, B^T g> for the ISTFT (real inner products)."""
    npdt = np.float32 if dtype == torch.float32 else np.float64
    w = torch.from_numpy(hann(n_fft, npdt))
    n_freq = n_fft // 2 + 1 if kw.get("onesided", True) else n_fft
    a = args_helper(torch.empty(1, n_freq, 1, dtype=dtype), hop_length=hop, window=w, **kw)
    plan = Plan(a, 2, frames, dtype, DEV)
    torch.manual_seed(0)
    cd = torch.complex64 if dtype == torch.float32 else torch.complex128
    x = torch.randn(2, plan.length, dtype=dtype, device=DEV)
    Y = torch.randn(2, n_freq, frames, dtype=cd, device=DEV)
    tol = 2e-5 if dtype == torch.float32 else 1e-12

    def rdot(u, v):
        return float((u.conj() * v).real.sum()) if u.is_complex() else float((u * v).sum())

    lhs, rhs = rdot(plan.stft(x), Y), rdot(x, plan.stft_adjoint(Y, plan.length))
    assert abs(lhs - rhs) < tol * max(1.0, abs(lhs)), (lhs, rhs)
    if kw.get("center", True):                          # with center=False the Hann envelope has zeros (1/0)
        gq = torch.randn(2, plan.length, dtype=dtype, device=DEV)
        lhs, rhs = rdot(plan.istft(Y), gq), rdot(Y, plan.istft_adjoint(gq))
        if kw.get("onesided", True):
            # irfft ignores the imaginary parts of DC / Nyquist: compare against Y with those removed
            Y0 = Y.clone()
            Y0[:, 0].imag.zero_()
            Y0[:, -1].imag.zero_()
            rhs = rdot(Y0, plan.istft_adjoint(gq))
        assert abs(lhs - rhs) < tol * max(1.0, abs(lhs)), (lhs, rhs)


ADMM_CASES = [("f32_hann", 128, 32, True, 0.1, {}, 3e-4), ("f64_hann", 128, 32, True, 0.1, {}, 1e-9),
              ("f64_rect_default", 64, None, False, 0.5, {}, 1e-9),
              ("f64_twosided", 64, 16, True, 0.2, dict(onesided=False, pad_mode="constant"), 1e-9)]


@pytest.mark.parametrize("tag,n_fft,hop,use_hann,rho,extra,tol", ADMM_CASES)
def test_admm_gradient_matches_reference_autograd(tag, n_fft, hop, use_hann, rho, extra, tol):
    g = load_golden("g11_autograd_admm")
    mag = g[f"mag_{tag}"]
    kw = dict(extra)
    if hop:
        kw["hop_length"] = hop
    if use_hann:
        kw["window"] = torch.from_numpy(hann(n_fft, mag.dtype.type))
    spec = T(mag).requires_grad_(True)
    y = si.ADMM(spec, max_iter=3, rho=rho, tol=0, verbose=False, **kw)
    assert y.requires_grad
    assert rel_l2(N(y), g[f"y_{tag}"]) < max(tol, 1e-5 if mag.dtype == np.float32 else 1e-11)
    (y * T(g[f"w_{tag}"])).sum().backward()
    assert rel_l2(N(spec.grad), g[f"grad_{tag}"]) < tol, rel_l2(N(spec.grad), g[f"grad_{tag}"])


def test_admm_gradient_complex_warm_start():
    g = load_golden("g11_autograd_admm")
    spec = T(g["c_complex"]).requires_grad_(True)
    y = si.ADMM(spec, max_iter=2, rho=0.3, tol=0, verbose=False, hop_length=32,
                window=torch.from_numpy(hann(128, np.float64)))
    (y * T(g["w_complex"])).sum().backward()
    assert rel_l2(N(spec.grad), g["grad_complex"]) < 1e-9


def test_admm_differentiable_forward_equals_fused_forward():
    mag = torch.rand(2, 513, 24, device=DEV, dtype=torch.float64) + 0.05
    kw = dict(hop_length=256, window=torch.hann_window(1024, dtype=torch.float64))
    y0 = si.ADMM(mag, max_iter=5, tol=0, verbose=False, **kw)
    y1 = si.ADMM(mag.clone().requires_grad_(True), max_iter=5, tol=0, verbose=False, **kw)
    assert rel_l2(N(y1), N(y0)) < 1e-10


def _rtisi_cases():
    return [str(m) for m in load_golden("g12_autograd_rtisi")["meta"]]


@pytest.mark.parametrize("meta", _rtisi_cases())
def test_rtisi_gradient_matches_reference_autograd(meta):
    g = load_golden("g12_autograd_rtisi")
    tag, n_fft, hop, la, asym, alpha, iters, onesided, normalized = meta.split("|")
    n_fft, hop, la, iters = int(n_fft), int(hop), int(la), int(iters)
    mag = g[f"mag_{tag}"]
    kw = {}
    if hop:
        kw["hop_length"] = hop
    if tag != "f64_rect_default":
        kw["window"] = torch.from_numpy(hann(n_fft, mag.dtype.type))
    if not int(onesided):
        kw["onesided"] = False
    if int(normalized):
        kw["normalized"] = True
    f32 = mag.dtype == np.float32
    spec = T(mag).requires_grad_(True)
    y = si.RTISI_LA(spec, look_ahead=la, asymmetric_window=bool(int(asym)), max_iter=iters, alpha=float(alpha),
                    verbose=False, **kw)
    assert y.requires_grad
    assert rel_l2(N(y), g[f"y_{tag}"]) < (2e-4 if f32 else 1e-9)
    (y * T(g[f"w_{tag}"])).sum().backward()
    assert spec.grad.shape == spec.shape
    err = rel_l2(N(spec.grad), g[f"grad_{tag}"])
    assert err < (2e-3 if f32 else 1e-8), err


def test_rtisi_reference_test_pattern_backward():
    """test/test_rtisila.py:46-70: spec.requires_grad, mse against the original signal, backward().  The defaults
    (symmetric window, alpha 0.99) amplify rounding noise, so the values are compared in float64 and the float32
    run is only required to produce a finite gradient, which is what the reference's test asserts."""
    g = load_golden("g12_autograd_rtisi")
    x = T(g["x_ref_test"])
    sp = torch.stft(x, 256, return_complex=True).abs().requires_grad_(True)
    y = si.RTISI_LA(sp, max_iter=2, verbose=False)
    torch.nn.functional.mse_loss(x[:y.shape[0]], y).backward()
    assert hasattr(sp, "grad") and sp.grad.shape == sp.shape and torch.isfinite(sp.grad).all()
    x64 = x.double()
    sp = torch.stft(x64, 256, return_complex=True).abs().requires_grad_(True)
    y = si.RTISI_LA(sp, max_iter=2, verbose=False)
    torch.nn.functional.mse_loss(x64[:y.shape[0]], y).backward()
    assert rel_l2(N(sp.grad), g["grad_ref_test64"]) < 1e-6, rel_l2(N(sp.grad), g["grad_ref_test64"])


def test_rtisi_gradient_from_cpu_leaf():
    """A CPU leaf tensor gets its gradient back on the CPU."""
    sp = (torch.rand(65, 8, dtype=torch.float64) + 0.05).requires_grad_(True)
    y = si.RTISI_LA(sp, max_iter=2, look_ahead=1, verbose=False, hop_length=32, window=torch.hann_window(128, dtype=torch.float64))
    assert y.device.type == "cpu"
    y.square().sum().backward()
    assert sp.grad is not None and sp.grad.device.type == "cpu" and torch.isfinite(sp.grad).all()


# ---- the reference's kwarg sweep with backward() (test/test_griffin.py:24-68, test_admm.py, test_rtisila.py:20-70) -----
def _sweep():
    for wl, win in ((None, None), (300, None), (300, "hann")):
        for hop in (None, 128):
            for center in (True, False):
                for normalized in (False, True):
                    for onesided in (False, True):
                        yield dict(win_length=wl, window=win, hop_length=hop, center=center, normalized=normalized,
                                   onesided=onesided)


@pytest.mark.parametrize("pad_mode", ["reflect", "constant", "replicate", "circular"])
@pytest.mark.parametrize("method", ["griffin_lim", "ADMM", "RTISI_LA"])
def test_reference_kwarg_sweep_is_differentiable(method, pad_mode):
    """Every stft-kwarg combination of the reference's `test_stft_args` (48 per pad mode; RTISI_LA also look_ahead in
    {-1, 2} x asymmetric_window): magnitudes of torch.stft with `requires_grad`, 2 iterations, mse against the signal,
    `backward()`.  Asserts what the reference asserts (rank, length, a gradient exists) plus its shape, and that it is
    finite whenever the waveform is (no centring + a window that vanishes at the ends gives the reference's 0/0)."""
    torch.manual_seed(5)
    x = torch.randn(4410, device=DEV)
    fn = getattr(si, method)
    n_done = 0
    for kw in _sweep():
        kw = dict(kw, pad_mode=pad_mode)
        if kw["window"] == "hann":
            kw["window"] = torch.hann_window(300, device=DEV)
        extras = [dict()] if method != "RTISI_LA" else [dict(look_ahead=la, asymmetric_window=a) for la in (-1, 2)
                                                        for a in (True, False)]
        for ex in extras:
            sp = torch.stft(x, 512, return_complex=True, **kw).abs().requires_grad_(True)
            y = fn(sp, max_iter=2, verbose=False, return_complex=False, **ex, **kw)
            assert y.dim() == 1 and 0 < y.shape[0] <= x.shape[0] + 512
            n = min(y.shape[0], x.shape[0])
            torch.nn.functional.mse_loss(x[:n], y[:n]).backward()
            assert sp.grad is not None and sp.grad.shape == sp.shape
            if bool(torch.isfinite(y).all()):
                assert bool(torch.isfinite(sp.grad).all()), (method, kw, ex)
            n_done += 1
    assert n_done == (48 if method != "RTISI_LA" else 192)


@pytest.mark.parametrize("fn,kw", [(si.griffin_lim, dict(alpha=0.5)), (si.ADMM, dict(rho=0.5))])
def test_backward_twice_with_retain_graph(fn, kw):
    """The reference's result is an ordinary autograd graph: a second backward under retain_graph=True works and gives
    the same gradient; without it the graph is released and autograd says so."""
    spec = torch.rand(2, 65, 20, dtype=torch.float64, device=DEV, requires_grad=True)
    y = fn(spec, max_iter=3, tol=0, verbose=False, hop_length=32, **kw)
    (g1,) = torch.autograd.grad(y.square().sum(), spec, retain_graph=True)
    (g2,) = torch.autograd.grad(y.square().sum(), spec)
    assert torch.equal(g1, g2) and g1.abs().sum() > 0
    with pytest.raises(RuntimeError, match="second time|already been freed"):
        torch.autograd.grad(y.square().sum(), spec)
