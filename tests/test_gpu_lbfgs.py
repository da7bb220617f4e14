"""L_BFGS path on the device: fused transform forward / loss+gradient (incl. the MFMA mel contractions),
vector kernels, optimiser trajectories.  Needs an MI355X: `-m gpu`."""
import numpy as np
import pytest
import torch

import oracle
from oracle.lbfgs import LogMelStft, MagStft
from _util import hann, load_golden, rel_l2

pytestmark = pytest.mark.gpu

import spectrogram_inversion_amd as si                                  # noqa: E402
from spectrogram_inversion_amd.lbfgs import LBFGS, HipVecOps            # noqa: E402
from spectrogram_inversion_amd.transforms import LogMelSTFT, MagSTFT    # noqa: E402


def dev():
    return torch.device("cuda", 0)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def N(t):
    return t.detach().cpu().numpy()


def test_vector_kernels():
    ops = HipVecOps(torch.float32, dev())
    rng = np.random.default_rng(0)
    a, b = rng.standard_normal(100003).astype(np.float32), rng.standard_normal(100003).astype(np.float32)
    ta, tb = T(a), T(b)
    assert abs(ops.dot(ta, tb) - float(np.dot(a.astype(np.float64), b.astype(np.float64)))) < 1e-6 * 100003 ** 0.5
    mx, sm = ops.absmax_abssum(ta)
    assert mx == float(np.abs(a).max()) and abs(sm - float(np.abs(a.astype(np.float64)).sum())) < 1e-8 * sm
    y = tb.clone()
    ops.axpy(0.37, ta, y)
    np.testing.assert_allclose(N(y), b + np.float32(0.37) * a, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(N(ops.scaled(-2.5, ta)), -2.5 * a, rtol=1e-7)


def test_mag_transform_golden():
    g = load_golden("g6_lbfgs")
    tr = MagSTFT(256)
    x0, spec = T(g["mag_x0"]), T(g["mag_spec"])
    v = tr(x0)
    ref = oracle.stft(g["mag_x0"], oracle.args_helper(129, np.float32))
    assert rel_l2(N(v), np.abs(ref)) < 2e-6
    _, fg = tr.bind(x0, spec)
    loss, grad = fg(x0)
    assert abs(loss - float(g["mag_loss0"])) < 1e-5 * float(g["mag_loss0"])
    assert rel_l2(N(grad), g["mag_grad0"]) < 1e-5


def test_logmel_transform_golden():
    """BASELINE config 5 transform: log1p(mel @ |STFT|), mel GEMMs on the exact-fp32 MFMA."""
    g = load_golden("g6_lbfgs")
    fb = si.mel_filterbank(22050, 2048, 80)
    tr = LogMelSTFT(torch.from_numpy(fb), 2048, hop_length=512, window=torch.from_numpy(hann(2048)))
    x, tgt = T(g["mel_x"]), T(g["mel_target"])
    v = tr(x)
    assert v.shape == tuple(g["mel_fwd"].shape)
    assert rel_l2(N(v), g["mel_fwd"]) < 1e-5
    _, fg = tr.bind(x, tgt)
    loss, grad = fg(x)
    assert abs(loss - float(g["mel_loss"])) < 1e-5 * float(g["mel_loss"])
    assert rel_l2(N(grad), g["mel_grad"]) < 1e-5


@pytest.mark.parametrize("kw", [dict(center=True, pad_mode="reflect"), dict(center=True, pad_mode="constant"),
                                dict(center=True, pad_mode="replicate"), dict(center=True, pad_mode="circular"),
                                dict(center=False), dict(center=True, normalized=True, onesided=False),
                                dict(center=True, hop_length=100, win_length=300)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_gradient_vs_autograd(kw, dtype):
    """Analytic STFT-magnitude gradient against torch autograd (torch.stft on the same device) for every
    pad mode / sidedness the reference's tests sweep (test/test_griffin.py:24-32)."""
    torch.manual_seed(3)
    x = torch.randn(2, 3000, dtype=dtype, device=dev())
    n_fft = 512
    skw = dict(kw)
    win = torch.hann_window(skw.get("win_length", n_fft), dtype=dtype, device=dev())
    xt = x.clone().requires_grad_(True)
    spec = torch.stft(xt, n_fft, window=win, return_complex=True, **skw).abs()
    target = torch.rand_like(spec)
    loss_ref = torch.nn.functional.mse_loss(spec, target)
    (g_ref,) = torch.autograd.grad(loss_ref, xt)
    tr = MagSTFT(n_fft, window=win, **skw)
    _, fg = tr.bind(x, target)
    loss, grad = fg(x)
    tol = 2e-5 if dtype == torch.float32 else 1e-11
    assert abs(loss - loss_ref.item()) < tol * loss_ref.item()
    assert rel_l2(N(grad), N(g_ref)) < tol


def test_logmel_gradient_vs_autograd_f64():
    torch.manual_seed(4)
    x = 0.1 * torch.randn(2, 4096, dtype=torch.float64, device=dev())
    fb = torch.from_numpy(si.mel_filterbank(22050, 1024, 40, dtype=np.float64)).to(dev())
    win = torch.hann_window(1024, dtype=torch.float64, device=dev())

    def fn(v):
        return torch.log1p(torch.matmul(fb, torch.stft(v, 1024, hop_length=256, window=win, return_complex=True).abs()))

    target = fn(x + 0.05 * torch.randn_like(x))
    xt = x.clone().requires_grad_(True)
    loss_ref = torch.nn.functional.mse_loss(fn(xt), target)
    (g_ref,) = torch.autograd.grad(loss_ref, xt)
    tr = LogMelSTFT(fb, 1024, hop_length=256, window=win)
    assert rel_l2(N(tr(x)), N(fn(x))) < 1e-12
    _, fg = tr.bind(x, target)
    loss, grad = fg(x)
    assert abs(loss - loss_ref.item()) < 1e-11 * loss_ref.item()
    assert rel_l2(N(grad), N(g_ref)) < 1e-10


@pytest.mark.parametrize("tag,kw", [("wolfe", dict(max_iter=40, history_size=5, line_search_fn="strong_wolfe")),
                                    ("wolfe_h100", dict(max_iter=25, line_search_fn="strong_wolfe")),
                                    ("fixed", dict(max_iter=30, lr=1e-3, history_size=4))])
def test_optimizer_retraces_torch_lbfgs(tag, kw):
    """float64 Rosenbrock: the optimiser (host control flow + HIP vector kernels) retraces torch.optim.LBFGS."""
    g = load_golden("g9_lbfgs_rosen")
    x = T(g["x0"].copy())
    losses = []

    def fg(v):
        a, b = v[1:] - v[:-1] ** 2, 1.0 - v[:-1]
        f = float((100.0 * a * a + b * b).sum())
        gr = torch.zeros_like(v)
        gr[1:] += 200.0 * a
        gr[:-1] += -400.0 * a * v[:-1] - 2.0 * b
        losses.append(f)
        return f, gr

    opt = LBFGS(x, **kw)
    for _ in range(2):
        opt.step(fg)
    ref = g[f"losses_{tag}"]
    assert len(losses) == len(ref)
    np.testing.assert_allclose(losses, ref, rtol=5e-4, atol=1e-7)
    np.testing.assert_allclose(N(x), g[f"x_{tag}"], rtol=1e-4, atol=1e-6)


def test_l_bfgs_first_steps_golden():
    g = load_golden("g6_lbfgs")
    x = si.L_BFGS(T(g["mag_spec"]), MagSTFT(256), init_x0=T(g["mag_x0"]), outer_max_iter=1, tol=0, eva_iter=1,
                  verbose=False, max_iter=3)
    assert rel_l2(N(x), g["mag_x_3inner"]) < 1e-4
    # one outer step of 10 inner iterations: loss after the step as recorded from the reference
    from spectrogram_inversion_amd.metrics import _sums
    x = si.L_BFGS(T(g["mag_spec"]), MagSTFT(256), init_x0=T(g["mag_x0"]), outer_max_iter=1, tol=0, eva_iter=1,
                  verbose=False, max_iter=10)
    s = _sums(MagSTFT(256)(x), T(g["mag_spec"]))
    assert abs(s[0] / s[3] - g["mag_trace_plain"][0, 1]) < 2e-3 * g["mag_trace_plain"][0, 1]


def test_l_bfgs_logmel_first_outer_step():
    g = load_golden("g6_lbfgs")
    fb = si.mel_filterbank(22050, 2048, 80)
    tr = LogMelSTFT(torch.from_numpy(fb), 2048, hop_length=512, window=torch.from_numpy(hann(2048)))
    x = si.L_BFGS(T(g["mel_target"]), tr, init_x0=T(g["mel_x"]), outer_max_iter=1, tol=0, eva_iter=1, verbose=False)
    assert rel_l2(N(x), g["mel_x_plain"]) < 2e-2
    from spectrogram_inversion_amd.metrics import _sums
    s = _sums(tr(x), T(g["mel_target"]))
    assert abs(s[0] / s[3] - g["mel_trace_plain"][0, 1]) < 5e-3 * g["mel_trace_plain"][0, 1]


def test_l_bfgs_generic_callable_and_shapes():
    """Any differentiable callable is accepted like in the reference (test/test_lbfgs.py:17-22)."""
    torch.manual_seed(0)
    for shape in [(4410,), (2, 4410), (1, 4410)]:
        x = torch.randn(*shape, device=dev())

        def trsfn(v):
            return torch.stft(v, 256, return_complex=True).abs()

        spec = trsfn(x)
        for metric in ("sc", "snr", "ser"):
            y = si.L_BFGS(spec, trsfn, samples=x.shape, max_iter=10, metric=metric, eva_iter=3, outer_max_iter=3,
                          verbose=False)
            assert y.shape == x.shape and bool(torch.isfinite(y).all())
    # the device transform gives the same first step as the autograd closure
    x0 = 1e-2 * torch.randn(2, 4410, device=dev())
    spec = trsfn(torch.randn(2, 4410, device=dev()))
    ya = si.L_BFGS(spec, trsfn, init_x0=x0.clone(), outer_max_iter=1, max_iter=4, verbose=False, tol=0)
    yb = si.L_BFGS(spec, MagSTFT(256), init_x0=x0.clone(), outer_max_iter=1, max_iter=4, verbose=False, tol=0)
    assert rel_l2(N(yb), N(ya)) < 1e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("k,n", [(1, 1000), (7, 100003), (64, 50000), (150, 20011)])
def test_many_vector_passes(k, n, dtype):
    """g . v_j for k vectors in one pass, and one linear combination of them (more than 64 vectors: several launches)."""
    ops = HipVecOps(dtype, dev())
    rng = np.random.default_rng(k + n)
    npdt = np.float32 if dtype == torch.float32 else np.float64
    g = rng.standard_normal(n).astype(npdt)
    vs = [rng.standard_normal(n).astype(npdt) for _ in range(k)]
    tv = [T(v) for v in vs]
    dots = ops.multi_dot(T(g), tv)
    want = [float(np.dot(g.astype(np.float64), v.astype(np.float64))) for v in vs]
    assert np.allclose(dots, want, rtol=0, atol=(1e-9 if dtype == torch.float64 else 1e-6) * n ** 0.5)
    coef = rng.standard_normal(k)
    out = N(ops.lincomb(tv, coef))
    ref = sum(c * v.astype(np.float64) for c, v in zip(coef, vs))
    assert rel_l2(out, ref) < (1e-15 if dtype == torch.float64 else 1e-7)


@pytest.mark.parametrize("history", [3, 6])
def test_gram_direction_follows_the_two_loop_recursion(history):
    """The recursion on Gram matrices (2 passes over the memory) against the two-loop recursion with one dot / axpy
    per pair on the same memory, at every iteration of an optimisation (including those that drop the oldest pair)."""
    rng = np.random.default_rng(history)
    n = 4096
    A = rng.standard_normal((n, n // 8)).astype(np.float32)
    q = T(A @ A.T / n + 1e-3 * np.eye(n, dtype=np.float32))          # an ill-conditioned SPD quadratic
    b = T(rng.standard_normal(n).astype(np.float32))

    def fg(x):
        qx = q @ x
        return float(0.5 * torch.dot(x, qx) - torch.dot(b, x)), qx - b

    x = torch.zeros(n, device=dev())
    opt = LBFGS(x, max_iter=40, history_size=history, line_search_fn="strong_wolfe", tolerance_change=1e-12, tolerance_grad=0)
    assert opt.gram
    errs, sizes = [], []
    gram_direction = opt._direction

    def both(g):
        d = gram_direction(g)
        ref = opt.ops.direction(g, opt.ss, opt.ys, opt.rho, opt.h_diag)
        errs.append(rel_l2(N(d), N(ref)))
        sizes.append(len(opt.ss))
        return d

    opt._direction = both
    f_start = fg(x)[0]
    opt.step(fg)
    assert len(errs) > history + 3 and max(sizes) == history and sizes.count(history) > 3      # pairs were dropped
    assert max(errs) < 2e-5, errs
    assert fg(x)[0] < f_start - 1.0


# ---- the one-launch log-mel objective (kernels_objective.h) -------------------------------------------------------------
@pytest.mark.parametrize("n_fft,hop,frames,batch,n_mels,kw", [
    (2048, 512, 40, 3, 80, {}),                                   # BASELINE config 5's frame size
    (2048, 512, 37, 2, 80, dict(pad_mode="constant")),            # a tile of 5 frames at the end
    (2048, 512, 16, 1, 80, dict(pad_mode="replicate")),           # exactly one tile
    (2048, 512, 33, 2, 80, dict(pad_mode="circular")),
    (2048, 1024, 24, 2, 80, {}),                                  # hop = n_fft/2
    (2048, 700, 21, 2, 64, {}),                                   # hop not dividing n_fft (even offsets)
    (2048, 333, 48, 2, 40, {}),                                   # odd hop: scalar overlap-add
    (2048, 1500, 20, 2, 80, {}),                                  # hardly any overlap
    (1024, 256, 50, 3, 40, {}),
    (1024, 128, 64, 2, 128, {}),                                  # 8 mel tiles (+ an empty ninth at n_fft 1024), hop = n_fft/8
    (1024, 256, 30, 2, 100, {}),                                  # 7 mel tiles, the last one partial, on the nine-tile kernel
    (2048, 512, 24, 2, 128, {}),                                  # 8 mel tiles at n_fft 2048
    (1024, 256, 40, 2, 20, dict(normalized=True)),                # padded mel tile, ortho scaling
    (2048, 512, 20, 2, 80, dict(center=False)),                   # no padding: signal of (T-1) hop + n_fft samples
    (2048, 512, 100, 24, 80, {}),                                 # the frame walk on 288 chunks of 8 / 9 frames, skewed pairs
    (1024, 256, 259, 5, 64, dict(pad_mode="replicate")),          # ... chunks of 8 ... 9 frames at n_fft 1024, an odd frame count
    (2048, 512, 9, 1, 80, {}),                                    # ... ONE chunk
    (2048, 256, 200, 12, 80, {}),                                 # ... hop = n_fft/8: chunks of 16 / 17 frames, seven-block seams
    (1024, 512, 300, 9, 40, dict(pad_mode="circular")),           # ... hop = n_fft/2
])
def test_one_launch_objective_vs_chain_and_oracle(monkeypatch, n_fft, hop, frames, batch, n_mels, kw):
    """`specinv_transform_loss_grad` for the log-mel transform as ONE kernel (spectrum kept on the chip) - the filterbank as bands on
    the vector units (what a mel filterbank gets) and as 16 x 16 blocks on the matrix cores (any matrix) - against the same
    objective as a chain of kernels (spectrum through HBM) and against the float64 oracle: loss and gradient."""
    rng = np.random.default_rng(n_fft + hop + frames)
    center = kw.get("center", True)
    length = (frames - 1) * hop + (0 if center else n_fft)
    fb = si.mel_filterbank(22050, n_fft, n_mels)
    w = hann(n_fft)
    xs = (0.1 * rng.standard_normal((batch, length))).astype(np.float32)
    x0 = (0.05 * rng.standard_normal((batch, length))).astype(np.float32)
    out = {}
    # the frame walk (kernels_objective_walk.h) serves hop = n_fft / 2, / 4, / 8, centred; where it does not apply the default is the band form
    walks = n_fft // hop in (2, 4, 8) and n_fft % hop == 0 and center and n_mels <= 128 and frames >= (16 if 8 * hop == n_fft else 8)
    for mode in ("walk", "bands", "matrix", "chain"):
        monkeypatch.setenv("SPECINV_DISABLE_FUSED_OBJECTIVE", "1" if mode == "chain" else "0")
        monkeypatch.setenv("SPECINV_REQUIRE_FUSED_OBJECTIVE", "0" if mode == "chain" else "1")
        monkeypatch.setenv("SPECINV_OBJ_SPARSE", "0" if mode == "matrix" else "1")
        monkeypatch.setenv("SPECINV_OBJ_WALK", "1" if mode == "walk" else "0")
        tr = LogMelSTFT(T(fb), n_fft, hop_length=hop, window=torch.from_numpy(w), **kw)
        target = tr(T(xs))
        _, fg = tr.bind(T(x0), target)
        loss, grad = fg(T(x0))
        assert fg.device_objective[0].objective_kind == (mode if mode != "walk" or walks else "bands")
        loss2, grad2 = fg(T(x0))
        assert loss == loss2 and torch.equal(grad, grad2)           # fixed summation order: bitwise reproducible
        out[mode] = (loss, N(grad), N(target))
    a = oracle.args_helper(n_fft // 2 + 1, np.float64, hop_length=hop, window=w.astype(np.float64), **kw)
    ref = LogMelStft(a, fb.astype(np.float64))
    lo, go = ref.loss_grad(x0.astype(np.float64), ref.forward(xs.astype(np.float64)))
    lc, gc, tc = out["chain"]
    for mode in ("walk", "bands", "matrix"):
        lf, gf, tf = out[mode]
        assert np.array_equal(tf, tc)
        assert abs(lf - lc) < 2e-6 * abs(lc), (mode, lf, lc)
        assert rel_l2(gf, gc) < 3e-6, (mode, rel_l2(gf, gc))
        assert abs(lf - lo) < 1e-5 * abs(lo), (mode, lf, lo)
        assert rel_l2(gf, go) < 1e-5, (mode, rel_l2(gf, go))


@pytest.mark.parametrize("n_fft,hop,n_mels", [(2048, 512, 140), (1024, 256, 137)])
def test_one_launch_objective_beyond_128_bands(monkeypatch, n_fft, hop, n_mels):
    """The matrix-core kernels stop at 128 bands (8 tiles of 16 rows); the band form takes a mel filterbank of up to 140."""
    monkeypatch.setenv("SPECINV_REQUIRE_FUSED_OBJECTIVE", "1")
    rng = np.random.default_rng(n_mels)
    frames, batch = 35, 2
    fb = si.mel_filterbank(22050, n_fft, n_mels)
    w = hann(n_fft)
    xs = (0.1 * rng.standard_normal((batch, (frames - 1) * hop))).astype(np.float32)
    x0 = (0.05 * rng.standard_normal(xs.shape)).astype(np.float32)
    tr = LogMelSTFT(T(fb), n_fft, hop_length=hop, window=torch.from_numpy(w))
    _, fg = tr.bind(T(x0), tr(T(xs)))
    loss, grad = fg(T(x0))
    assert fg.device_objective[0].objective_kind == "bands"
    a = oracle.args_helper(n_fft // 2 + 1, np.float64, hop_length=hop, window=w.astype(np.float64))
    ref = LogMelStft(a, fb.astype(np.float64))
    lo, go = ref.loss_grad(x0.astype(np.float64), ref.forward(xs.astype(np.float64)))
    assert abs(loss - lo) < 1e-5 * abs(lo) and rel_l2(N(grad), go) < 1e-5


def test_one_launch_objective_falls_back_where_it_does_not_fit(monkeypatch):
    """Configurations the one-launch kernel does not cover (n_fft other than 1024 / 2048; a dense matrix of more than 128 rows, a
    mel filterbank of more than 140 bands; float64; two-sided spectra) run the kernel chain - same results as ever."""
    fb = si.mel_filterbank(22050, 512, 40)
    x = 0.1 * torch.randn(2, 30 * 128, device=dev())
    tr = LogMelSTFT(T(fb), 512, hop_length=128, window=torch.from_numpy(hann(512)))
    _, fg = tr.bind(x, tr(x + 0.01))
    monkeypatch.setenv("SPECINV_REQUIRE_FUSED_OBJECTIVE", "1")
    with pytest.raises(NotImplementedError, match="one-launch objective"):
        fg(x)
    monkeypatch.setenv("SPECINV_REQUIRE_FUSED_OBJECTIVE", "0")
    loss, g = fg(x)
    assert loss > 0 and torch.isfinite(g).all()
    # float64 stays on the chain as well
    x64 = 0.1 * torch.randn(2, 30 * 256, device=dev(), dtype=torch.float64)
    fb = si.mel_filterbank(22050, 1024, 40).astype(np.float64)
    tr = LogMelSTFT(T(fb), 1024, hop_length=256, window=torch.from_numpy(hann(1024, np.float64)))
    _, fg = tr.bind(x64, tr(x64 + 0.01))
    monkeypatch.setenv("SPECINV_REQUIRE_FUSED_OBJECTIVE", "1")
    with pytest.raises(NotImplementedError, match="one-launch objective"):
        fg(x64)


@pytest.mark.parametrize("center", [True, False])
def test_one_launch_objective_signal_longer_than_its_frames(monkeypatch, center):
    """torch.stft drops what is left of the signal after the last whole hop (the reference's demo signal, main.py:16-43,
    has such a remainder).  Centred: the remainder is still inside the last frames, only padded positions stay uncovered;
    not centred: the last samples take no part in the objective and get a zero gradient.  Against the float64 oracle."""
    monkeypatch.setenv("SPECINV_REQUIRE_FUSED_OBJECTIVE", "1")
    rng = np.random.default_rng(3)
    n_fft, hop, frames = 1024, 128, 41
    length = (frames - 1) * hop + (0 if center else n_fft) + 97
    w = hann(n_fft)
    xs = (0.1 * rng.standard_normal((2, length))).astype(np.float32)
    x0 = (0.05 * rng.standard_normal((2, length))).astype(np.float32)
    a = oracle.args_helper(n_fft // 2 + 1, np.float64, hop_length=hop, window=w.astype(np.float64), center=center)
    fb = si.mel_filterbank(22050, n_fft, 40)
    kw = dict(hop_length=hop, window=torch.from_numpy(w), center=center)
    for tr, ref in ((MagSTFT(n_fft, **kw), MagStft(a)), (LogMelSTFT(T(fb), n_fft, **kw), LogMelStft(a, fb.astype(np.float64)))):
        target = tr(T(xs))
        assert target.shape[-1] == frames
        _, fg = tr.bind(T(x0), target)
        loss, grad = fg(T(x0))
        lo, go = ref.loss_grad(x0.astype(np.float64), ref.forward(xs.astype(np.float64)))
        assert abs(loss - lo) < 1e-5 * abs(lo) and rel_l2(N(grad), go) < 1e-5
        if not center:
            assert not N(grad)[:, -97:].any() and not go[:, -97:].any()


@pytest.mark.parametrize("n_fft,hop,rows,per_bin", [(2048, 512, 60, 4), (1024, 256, 90, 3), (2048, 1024, 24, 1), (1024, 128, 40, 5)])
def test_one_launch_objective_with_a_banded_matrix(monkeypatch, n_fft, hop, rows, per_bin):
    """A filterbank whose rows are overlapping random bands, `per_bin` rows meeting every bin: up to four per bin run as bands (the
    kernel has the two forms 'two rows per bin' and 'four', shorter columns padded with zero weights), five stay on the matrix
    cores.  Against the float64 oracle."""
    monkeypatch.setenv("SPECINV_REQUIRE_FUSED_OBJECTIVE", "1")
    monkeypatch.setenv("SPECINV_OBJ_WALK", "0")                  # (the tile kernel's band forms are what this test is about)
    rng = np.random.default_rng(rows + per_bin)
    F, frames, batch = n_fft // 2 + 1, 37, 2
    fb = np.zeros((rows, F), np.float32)
    step = F / rows
    for m in range(rows):                       # row m covers bins [m step, (m + per_bin) step): per_bin rows over every bin
        lo, hi = int(m * step), min(F, int((m + per_bin) * step))
        fb[m, lo:hi] = 0.01 + 0.02 * rng.random(hi - lo)
    assert np.count_nonzero(fb, axis=0).max() == per_bin
    w = hann(n_fft)
    xs = (0.1 * rng.standard_normal((batch, (frames - 1) * hop))).astype(np.float32)
    x0 = (0.05 * rng.standard_normal(xs.shape)).astype(np.float32)
    tr = LogMelSTFT(T(fb), n_fft, hop_length=hop, window=torch.from_numpy(w))
    _, fg = tr.bind(T(x0), tr(T(xs)))
    loss, grad = fg(T(x0))
    assert fg.device_objective[0].objective_kind == ("bands" if per_bin <= 4 else "matrix")
    a = oracle.args_helper(F, np.float64, hop_length=hop, window=w.astype(np.float64))
    ref = LogMelStft(a, fb.astype(np.float64))
    lo_, go = ref.loss_grad(x0.astype(np.float64), ref.forward(xs.astype(np.float64)))
    assert abs(loss - lo_) < 1e-5 * abs(lo_), (loss, lo_)
    assert rel_l2(N(grad), go) < 1e-5, rel_l2(N(grad), go)


@pytest.mark.parametrize("n_mels", [80, 33])
def test_one_launch_objective_with_a_dense_matrix(monkeypatch, n_mels):
    """The block list leaves out all-zero 16 x 16 blocks of the filterbank; a dense (random, signed-free) matrix keeps
    every block (and stays on the matrix cores), a matrix with one non-zero entry is one band - both against the float64 oracle."""
    n_fft, hop, frames, batch = 1024, 256, 40, 2
    rng = np.random.default_rng(n_mels)
    w = hann(n_fft)
    xs = (0.1 * rng.standard_normal((batch, (frames - 1) * hop))).astype(np.float32)
    x0 = (0.05 * rng.standard_normal(xs.shape)).astype(np.float32)
    dense = (0.02 * rng.random((n_mels, n_fft // 2 + 1))).astype(np.float32)
    single = np.zeros_like(dense)
    single[n_mels - 1, 300] = 0.5
    monkeypatch.setenv("SPECINV_REQUIRE_FUSED_OBJECTIVE", "1")
    monkeypatch.setenv("SPECINV_OBJ_WALK", "0")
    for fb in (dense, single, np.zeros_like(dense)):
        tr = LogMelSTFT(T(fb), n_fft, hop_length=hop, window=torch.from_numpy(w))
        _, fg = tr.bind(T(x0), tr(T(xs)))
        loss, grad = fg(T(x0))
        # (a dense matrix has no band form: every bin meets every row; one entry, or none, is the sparsest band there is)
        assert fg.device_objective[0].objective_kind == ("matrix" if fb is dense else "bands")
        a = oracle.args_helper(n_fft // 2 + 1, np.float64, hop_length=hop, window=w.astype(np.float64))
        ref = LogMelStft(a, fb.astype(np.float64))
        lo, go = ref.loss_grad(x0.astype(np.float64), ref.forward(xs.astype(np.float64)))
        if not fb.any():
            assert loss == 0.0 and not N(grad).any()
            continue
        assert abs(loss - lo) < 1e-5 * abs(lo), (loss, lo)
        assert rel_l2(N(grad), go) < 1e-5, rel_l2(N(grad), go)


@pytest.mark.parametrize("n_fft,hop,frames,batch,kw", [
    (2048, 512, 40, 2, {}), (2048, 333, 37, 2, dict(pad_mode="constant")), (1024, 256, 50, 3, dict(normalized=True)),
    (1024, 128, 64, 2, dict(pad_mode="circular")), (2048, 1024, 20, 1, dict(center=False)), (1024, 300, 33, 2, dict(pad_mode="replicate")),
])
def test_one_launch_magnitude_objective(monkeypatch, n_fft, hop, frames, batch, kw):
    """mean((|STFT(x)| - target)^2) - `MagSTFT`, the transform of the reference's own test and demo (test/test_lbfgs.py:17-18,
    main.py:21-43) - on the one-launch kernel (the log-mel kernel without the contractions) against the kernel chain and
    the float64 oracle."""
    rng = np.random.default_rng(n_fft + hop)
    center = kw.get("center", True)
    length = (frames - 1) * hop + (0 if center else n_fft)
    w = hann(n_fft)
    xs = (0.1 * rng.standard_normal((batch, length))).astype(np.float32)
    x0 = (0.05 * rng.standard_normal((batch, length))).astype(np.float32)
    out = {}
    for mode in ("fused", "chain"):
        monkeypatch.setenv("SPECINV_DISABLE_FUSED_OBJECTIVE", "1" if mode == "chain" else "0")
        monkeypatch.setenv("SPECINV_REQUIRE_FUSED_OBJECTIVE", "1" if mode == "fused" else "0")
        tr = MagSTFT(n_fft, hop_length=hop, window=torch.from_numpy(w), **kw)
        target = tr(T(xs))
        _, fg = tr.bind(T(x0), target)
        loss, grad = fg(T(x0))
        loss2, grad2 = fg(T(x0))
        assert loss == loss2 and torch.equal(grad, grad2)
        out[mode] = (loss, N(grad))
    (lf, gf), (lc, gc) = out["fused"], out["chain"]
    assert abs(lf - lc) < 2e-6 * abs(lc), (lf, lc)
    assert rel_l2(gf, gc) < 3e-6, rel_l2(gf, gc)
    a = oracle.args_helper(n_fft // 2 + 1, np.float64, hop_length=hop, window=w.astype(np.float64), **kw)
    ref = MagStft(a)
    lo, go = ref.loss_grad(x0.astype(np.float64), ref.forward(xs.astype(np.float64)))
    assert abs(lf - lo) < 1e-5 * abs(lo), (lf, lo)
    assert rel_l2(gf, go) < 1e-5, rel_l2(gf, go)


@pytest.mark.parametrize("n", [1000, 100003, 2048 * 1024 + 5, 8388608])
def test_pair_statistics_pass(n):
    """`k_lbfgs_pair_stats`: y, s and the eight scalars of an L-BFGS iteration in one pass over g, g_prev, d (+ a one-workgroup
    reduction of the per-block sums in block order).  Against float64 NumPy, against the separate pair pass, and 50 launches in a
    row bit for bit.  (Letting the last workgroup to finish do that reduction - no second launch - was tried: the agent-scope
    release / acquire it needs costs 25 us per launch in L2 write-backs, against the 7 us of the second kernel.)"""
    ops = HipVecOps(torch.float32, dev())
    rng = np.random.default_rng(n)
    g, gp, d = (T(rng.standard_normal(n).astype(np.float32)) for _ in range(3))
    t = 0.37
    board = torch.zeros(16, dtype=torch.float64, device=dev())
    first = None
    for rep in range(50):
        y, s = ops.plan.lbfgs_pair_stats_dev(g, gp, d, t, board.data_ptr())
        vals = np.array(ops.plan.read_doubles(board.data_ptr(), 8))
        if first is None:
            first = (vals, y.clone(), s.clone())
        else:
            assert np.array_equal(vals, first[0]), rep
    assert torch.equal(y, first[1]) and torch.equal(s, first[2])
    g64, gp64, d64 = (N(v).astype(np.float64) for v in (g, gp, d))
    y64, s64 = N(y).astype(np.float64), N(s).astype(np.float64)
    want = [g64 @ d64, np.abs(g64).sum(), np.abs(g64).max(), np.abs(d64).max(), y64 @ s64, y64 @ y64, g64 @ g64, g64 @ gp64]
    np.testing.assert_allclose(first[0], want, rtol=1e-12, atol=1e-9)
    y2, s2, ys, yy = ops.pair(g, gp, d, t)
    assert torch.equal(y, y2) and torch.equal(s, s2)
    np.testing.assert_allclose([ys, yy], first[0][4:6], rtol=1e-12)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("k,n", [(1, 1003), (5, 100003), (70, 50000)])
def test_direction_and_step_in_one_pass(k, n, dtype):
    """`specinv_vec_lincomb_step`: d = sum c_j v_j and x += t d in the pass that completes the sum (more than 64 vectors: the last
    launch): the same d and the same x, bit for bit, as `lincomb` followed by `axpy`."""
    ops = HipVecOps(dtype, dev())
    rng = np.random.default_rng(k + n)
    npdt = np.float32 if dtype == torch.float32 else np.float64
    tv = [T(rng.standard_normal(n).astype(npdt)) for _ in range(k)]
    coef = rng.standard_normal(k)
    x0 = T(rng.standard_normal(n).astype(npdt))
    t = 0.8125
    d_ref = ops.lincomb(tv, coef)
    x_ref = x0.clone()
    ops.axpy(t, d_ref, x_ref)
    x = x0.clone()
    d = ops.lincomb_step(tv, coef, t, x)
    assert torch.equal(d, d_ref) and torch.equal(x, x_ref)


def test_host_board_receives_device_results():
    """`specinv_board_alloc`: the scalar results of the optimiser passes are written by the kernels into pinned host memory;
    `specinv_stream_wait` makes them readable - the same values as through a device buffer and `specinv_read_doubles`."""
    ops = HipVecOps(torch.float32, dev())
    rng = np.random.default_rng(3)
    n = 300007
    g, gp, d = (T(rng.standard_normal(n).astype(np.float32)) for _ in range(3))
    hb = ops.board(16)
    db = torch.zeros(16, dtype=torch.float64, device=dev())
    for rep in range(20):
        t = 0.1 + 0.05 * rep
        ops.plan.lbfgs_pair_stats_dev(g, gp, d, t, hb.data_ptr() + 8)         # slots 1..8
        ops.plan.lbfgs_pair_stats_dev(g, gp, d, t, db.data_ptr() + 8)
        hb.put(0, float(rep))
        got = ops.read(hb, 9)
        want = ops.plan.read_doubles(db.data_ptr(), 9)
        assert got[0] == float(rep) and list(got[1:]) == list(want[1:]), rep


def test_one_pinned_board_per_plan():
    """Every optimiser on a device / dtype shares the plan `get_plan` caches, and with it ONE pinned result board: a service that
    calls `L_BFGS` in a loop must not pin another host region per call (the plan frees its boards only when it is destroyed)."""
    from spectrogram_inversion_amd.lbfgs import LBFGS
    x = torch.zeros(4096, device=dev())
    a, b = LBFGS(x), LBFGS(x.clone(), history_size=7)
    assert a.ops.plan is b.ops.plan
    assert a.ops.board(23) is b.ops.board(9 + 2 * 100)
    big = a.ops.board(1000)                                  # grows once, then serves everybody
    assert big.numel() >= 1000 and b.ops.board(50) is big


# ---- the device-resident optimiser (csrc/lbfgs_dev.h): decisions in k_lbd_decide, one host synchronisation per step ----------
def _device_problem(kind, seed=7, shape=None):
    rng = np.random.default_rng(seed)
    if kind == "logmel":
        n_fft, hop, frames, batch = shape or (2048, 512, 24, 2)
        tr = LogMelSTFT(torch.from_numpy(si.mel_filterbank(22050, n_fft, 80)).to(dev()), n_fft, hop_length=hop,
                        window=torch.from_numpy(hann(n_fft)))
    else:
        n_fft, hop, frames, batch = 1024, 256, 30, 3
        tr = MagSTFT(n_fft, hop_length=hop, window=torch.from_numpy(hann(n_fft)))
    length = (frames - 1) * hop
    xs = torch.from_numpy((0.1 * rng.standard_normal((batch, length))).astype(np.float32)).to(dev())
    x0 = torch.from_numpy((1e-2 * rng.standard_normal((batch, length))).astype(np.float32)).to(dev())
    return tr, tr(xs), x0


def _run_steps(monkeypatch, device_path, tr, target, x0, steps, **kw):
    monkeypatch.setenv("SPECINV_LBFGS_DEVICE", "1" if device_path else "0")
    x = x0.clone()
    _, fg = tr.bind(x, target)
    opt = LBFGS(x, device=dev(), **kw)
    losses, snaps = [], []
    for _ in range(steps):
        losses.append(opt.step(fg))
        snaps.append((x.clone(), opt.total_iters, opt.func_evals, int(opt.pairs_accepted), int(opt.pairs_rejected), opt.history_len))
    assert bool(opt._dev) == device_path
    return losses, snaps


@pytest.mark.parametrize("kind,kw,steps", [
    ("logmel", dict(), 3),                                       # torch.optim.LBFGS defaults: max_iter 20, history 100
    ("logmel", dict(max_iter=12, history_size=3), 4),            # the memory ring wraps (drop-oldest on the device)
    ("mag", dict(max_iter=10), 3),
    ("mag", dict(max_iter=50, history_size=10), 2),              # the reference demo's options (main.py:43)
    ("mag", dict(max_iter=20, tolerance_change=1e-4), 3),        # a tolerance that breaks steps early: the rest is no-ops
    ("logmel", dict(max_iter=6, max_eval=4), 3),                 # max_eval ends the inner loop
])
def test_device_resident_optimiser_retraces_the_host_driven_loop(monkeypatch, kind, kw, steps):
    """One `optimizer.step` enqueued as a whole with the decisions of torch.optim.LBFGS.step taken by `k_lbd_decide` on the
    device (curvature guard, memory ring, Gram recursion, tolerance tests) against the same step driven from the host with one
    read-back per inner iteration (`_step_packed`, itself retraced against torch.optim.LBFGS in test_host_logic.py and on the g6 /
    g9 fixtures): the same counters after every step, the same losses and iterates up to the order of the float64 sums."""
    tr, target, x0 = _device_problem(kind)
    la, sa = _run_steps(monkeypatch, False, tr, target, x0, steps, **kw)
    lb, sb = _run_steps(monkeypatch, True, tr, target, x0, steps, **kw)
    for i, (a, b) in enumerate(zip(sa, sb)):
        assert a[1:] == b[1:], (i, a[1:], b[1:])                  # total_iters, func_evals, accepted, rejected, history
        # (bit-identical while the memory is short; with dozens of pairs the float64 sums of the recursion are taken in another
        # order - wave reductions against numpy's dot - and the last digits of a nearly converged iterate move)
        tol = 1e-6 if a[5] <= 20 else 2e-3
        assert rel_l2(b[0].cpu().numpy(), a[0].cpu().numpy()) < tol, (i, rel_l2(b[0].cpu().numpy(), a[0].cpu().numpy()))
    np.testing.assert_allclose(lb, la, rtol=1e-4)
    assert sa[-1][3] > 0                                          # (pairs were accepted: the memory path ran)


def test_device_resident_optimiser_at_the_optimum_and_reuse(monkeypatch):
    """At a stationary point max|g| <= tolerance_grad ends the step at its entry evaluation (nothing moves, one evaluation);
    two optimisers on one plan keep their states apart; a dropped optimiser frees its state (its parameter-sized vectors go back to
    the plan's pool for the next optimiser)."""
    monkeypatch.setenv("SPECINV_LBFGS_DEVICE", "1")
    tr, target, x0 = _device_problem("mag")
    xs = torch.from_numpy((0.1 * np.random.default_rng(7).standard_normal(tuple(x0.shape))).astype(np.float32)).to(dev())
    x = xs.clone()
    _, fg = tr.bind(x, tr(xs))
    opt = LBFGS(x, device=dev(), tolerance_grad=1e-3)
    loss = opt.step(fg)
    assert opt._dev and loss < 1e-10 and opt.func_evals == 1 and opt.total_iters == 0 and torch.equal(x, xs)
    a, b = x0.clone(), x0.clone()
    _, fga = tr.bind(a, target)
    _, fgb = tr.bind(b, target)
    oa, ob = LBFGS(a, device=dev(), max_iter=5), LBFGS(b, device=dev(), max_iter=5)
    for _ in range(3):
        oa.step(fga)
    for _ in range(3):
        ob.step(fgb)
    assert oa._dev and ob._dev and oa._dev[1] != ob._dev[1] and torch.equal(a, b)
    plan = oa._dev[0]
    before = plan.device_bytes
    del oa, ob, opt
    import gc
    gc.collect()
    assert plan.device_bytes < before


# ---- strong Wolfe with one read-back per evaluation (lbfgs.py:_step_wolfe_packed) -------------------------------------------
def _run_wolfe(monkeypatch, packed, tr, target, x0, steps, **kw):
    monkeypatch.setenv("SPECINV_LBFGS_PACKED", "1" if packed else "0")
    x = x0.clone()
    _, fg = tr.bind(x, target)
    opt = LBFGS(x, device=dev(), line_search_fn="strong_wolfe", **kw)
    taken = []
    orig = opt._step_wolfe_packed
    opt._step_wolfe_packed = lambda f: (taken.append(1), orig(f))[1]
    losses, snaps = [], []
    for _ in range(steps):
        losses.append(opt.step(fg))
        snaps.append((x.clone(), opt.total_iters, opt.func_evals, int(opt.pairs_accepted), int(opt.pairs_rejected), opt.history_len))
    assert bool(taken) == packed
    return losses, snaps


@pytest.mark.parametrize("kind,kw,steps", [
    ("logmel", dict(), 3),                                       # torch.optim.LBFGS defaults + strong Wolfe: max_iter 20, history 100
    ("logmel", dict(max_iter=12, history_size=3), 3),            # the memory wraps
    ("mag", dict(max_iter=10), 3),
    ("mag", dict(max_iter=20, max_eval=8), 3),                   # max_eval ends line searches early (max_ls = max_eval - evaluations)
])
def test_packed_wolfe_step_retraces_the_general_step(monkeypatch, kind, kw, steps):
    """`line_search_fn='strong_wolfe'` on a device objective takes `_step_wolfe_packed` - the objective, the step statistics and
    g . d of every trial point enqueued together and fetched with one read, the new pair's products with the gradient by linearity -
    against the general `step` (three scalar read-backs around every evaluation; retraced against torch.optim.LBFGS on the g9
    fixture and in test_host_logic.py): the same counters after every step, losses and iterates equal to the rounding of the
    Gram products."""
    tr, target, x0 = _device_problem(kind)
    la, sa = _run_wolfe(monkeypatch, False, tr, target, x0, steps, **kw)
    lb, sb = _run_wolfe(monkeypatch, True, tr, target, x0, steps, **kw)
    for i, (a, b) in enumerate(zip(sa, sb)):
        assert a[1:] == b[1:], (i, a[1:], b[1:])                  # total_iters, func_evals, accepted, rejected, history
        # (the new pair's products with the gradient come by linearity here and from a pass over the vectors there: with dozens of
        # pairs in the memory the last digits of the Gram recursion move a nearly converged iterate)
        tol = 2e-4 if a[5] <= 12 else 5e-3
        assert rel_l2(b[0].cpu().numpy(), a[0].cpu().numpy()) < tol, (i, a[5], rel_l2(b[0].cpu().numpy(), a[0].cpu().numpy()))
    np.testing.assert_allclose(lb, la, rtol=1e-3)
    assert sa[-1][3] > 0 and la[-1] < 0.5 * la[0]                # pairs were accepted, the loss fell


@pytest.mark.parametrize("fused", [True, False])
def test_packed_wolfe_with_and_without_fused_statistics(monkeypatch, fused):
    """The trial points of the host-driven search take their statistics from the objective's own launches
    (specinv_transform_loss_grad_stats_dev) or from a pass of their own (SPECINV_LBFGS_FUSED_STATS=0): the same counters, iterates
    equal to the order of the float64 sums."""
    tr, target, x0 = _device_problem("logmel")
    monkeypatch.setenv("SPECINV_LBFGS_FUSED_STATS", "0")
    la, sa = _run_wolfe(monkeypatch, True, tr, target, x0, 3)
    monkeypatch.setenv("SPECINV_LBFGS_FUSED_STATS", "1" if fused else "0")
    lb, sb = _run_wolfe(monkeypatch, True, tr, target, x0, 3)
    for i, (a, b) in enumerate(zip(sa, sb)):
        assert a[1:] == b[1:], (i, a[1:], b[1:])
        assert rel_l2(b[0].cpu().numpy(), a[0].cpu().numpy()) < (1e-6 if a[5] <= 12 else 5e-3), i
    np.testing.assert_allclose(lb, la, rtol=1e-6)


# ---- the statistics of the gradient taken inside the objective's launches (kernels_objective.h: ObjArgs::st_*) -------------------
@pytest.mark.parametrize("kind,n_fft,hop,frames,batch,kw", [
    ("logmel", 2048, 512, 40, 2, {}),                                # seams between three tiles, reflect margins folded back
    ("logmel", 2048, 512, 16, 1, {}),                                # ONE tile: no seam
    ("logmel", 1024, 256, 70, 3, dict(pad_mode="constant")),         # margins dropped
    ("logmel", 1024, 64, 80, 2, dict(pad_mode="circular")),          # a fold stretch that reaches into the second and third tile
    ("logmel", 2048, 333, 37, 2, {}),                                # a hop that is no multiple of 4: scalar seams, scalar gather
    ("logmel", 1024, 256, 33, 2, dict(center=False)),
    ("mag", 1024, 256, 45, 2, dict(pad_mode="replicate")),
    ("mag", 2048, 1024, 20, 2, {}),
])
@pytest.mark.parametrize("with_d", [True, False])
def test_objective_statistics_ride_on_the_objective(monkeypatch, kind, n_fft, hop, frames, batch, kw, with_d):
    """{loss, g.d, sum|g|, max|g|, max|d|} from the objective's own launches - every gradient sample counted once, where it becomes
    final: by the epilogue's pass over the gradient, which finishes the seams on its way, or by the thread that folds a margin onto it - against the objective followed by k_lbfgs_stats' pass over g and d:
    the same gradient bit for bit, the maxima exactly, the sums to the order of their float64 additions."""
    monkeypatch.setenv("SPECINV_REQUIRE_FUSED_OBJECTIVE", "1")
    rng = np.random.default_rng(5)
    win = torch.from_numpy(hann(n_fft))
    if kind == "logmel":
        tr = LogMelSTFT(torch.from_numpy(si.mel_filterbank(22050, n_fft, 80)).to(dev()), n_fft, hop_length=hop, window=win, **kw)
    else:
        tr = MagSTFT(n_fft, hop_length=hop, window=win, **kw)
    length = (frames - 1) * hop + (0 if kw.get("center", True) else n_fft)
    xs = torch.from_numpy((0.1 * rng.standard_normal((batch, length))).astype(np.float32)).to(dev())
    x = torch.from_numpy((0.05 * rng.standard_normal((batch, length))).astype(np.float32)).to(dev())
    d = torch.from_numpy(rng.standard_normal((batch, length)).astype(np.float32)).to(dev()) if with_d else None
    _, fg = tr.bind(x, tr(xs))
    opt = LBFGS(x.clone(), device=dev())
    bd = opt.ops.board(9 + 2 * 100)
    g1 = fg.dev_stats(x, d, bd.data_ptr())
    got = np.array(opt.ops.read(bd, 5), dtype=np.float64).copy()
    g2 = opt.ops.eval_into(fg, x, bd, 0)
    opt.ops.stats_into(g2, g2 if d is None else d, bd, 1)
    want = np.array(opt.ops.read(bd, 5), dtype=np.float64).copy()
    assert torch.equal(g1, g2)
    assert got[3] == want[3] and got[4] == want[4]                   # maxima
    g64 = g1.double()
    scale = float((g64.abs() * (g64 if d is None else d.double()).abs()).sum())
    assert abs(got[0] - want[0]) <= 1e-12 * abs(want[0])             # loss
    assert abs(got[1] - want[1]) <= 1e-12 * scale                    # g . d
    assert abs(got[2] - want[2]) <= 1e-12 * want[2]                  # sum |g|


# ---- the lean iteration (csrc/lbfgs_dev.h: three launches while the memory is empty) against the full form -----------------------
@pytest.mark.parametrize("kind,kw,steps", [
    ("logmel", dict(), 3),                                       # pairs are accepted: the first chain is suspended and resumed in the full form
    ("mag", dict(max_iter=10), 3),
    ("mag", dict(max_iter=20, tolerance_change=1e-4), 3),        # a tolerance that ends steps inside the lean chain
    ("logmel", dict(max_iter=6, max_eval=4), 3),
])
def test_lean_iteration_retraces_the_full_form(monkeypatch, kind, kw, steps):
    """`k_lbd_direction_lean` - every workgroup finishes the evaluation's rows, takes the iteration's decisions and forms its share
    of the direction - against the full form from the first iteration on (SPECINV_LBFGS_LEAN=0: memory products, k_lbd_decide,
    k_lbd_lincomb_step): the same counters after every step and the same iterates (the same float operations on the same sums;
    the one accepted pair of a lean chain goes through the same scalar recursion)."""
    tr, target, x0 = _device_problem(kind)
    monkeypatch.setenv("SPECINV_LBFGS_LEAN", "0")
    la, sa = _run_steps(monkeypatch, True, tr, target, x0, steps, **kw)
    monkeypatch.setenv("SPECINV_LBFGS_LEAN", "1")
    lb, sb = _run_steps(monkeypatch, True, tr, target, x0, steps, **kw)
    for i, (a, b) in enumerate(zip(sa, sb)):
        assert a[1:] == b[1:], (i, a[1:], b[1:])
        assert rel_l2(b[0].cpu().numpy(), a[0].cpu().numpy()) < 1e-6, (i, rel_l2(b[0].cpu().numpy(), a[0].cpu().numpy()))
    np.testing.assert_allclose(lb, la, rtol=1e-9)


@pytest.mark.parametrize("kw,steps,shape", [
    (dict(), 2, None),                                           # pairs are accepted: every form hands over to the full one
    (dict(lr=1e-7, max_iter=8, tolerance_change=0.0, tolerance_grad=0.0), 3, None),   # every pair rejected (BASELINE C5's regime): lean throughout
    (dict(lr=1e-7, max_iter=5, tolerance_change=0.0, tolerance_grad=0.0, max_eval=4), 2, None),
    # the walk's other instantiations (n_fft 1024; hop = n_fft/2, /8), several chunks per item (seams), a batch of one
    (dict(lr=1e-7, max_iter=6, tolerance_change=0.0, tolerance_grad=0.0), 2, (1024, 256, 40, 3)),
    (dict(lr=1e-7, max_iter=6, tolerance_change=0.0, tolerance_grad=0.0), 2, (2048, 1024, 33, 1)),
    (dict(lr=1e-7, max_iter=6, tolerance_change=0.0, tolerance_grad=0.0), 2, (1024, 128, 48, 2)),
    (dict(), 2, (1024, 512, 17, 2)),
])
def test_deferred_step_and_two_launch_iteration_are_bit_identical(monkeypatch, kw, steps, shape):
    """Where the frame walk serves the objective, a lean iteration that accepts no pair leaves x += t d to the NEXT evaluation's
    walk (x_new = fma(t, (float)(c0 (double)g), x_old) formed while the samples are loaded, written to the iterate's other
    buffer), and its decisions are taken by the last workgroup of the evaluation's epilogue - two launches per iteration.  Against
    the three-launch iteration with the step deferred (SPECINV_LBFGS_LEAN2=0) and with the step streamed by
    k_lbd_direction_lean (SPECINV_LBFGS_DEFER=0): the same float operations on the same sums - identical iterates, bit for bit,
    after every step (the step still pending at the end of one is applied by k_lbd_settle_x)."""
    tr, target, x0 = _device_problem("logmel", shape=shape)
    runs = []
    for env in (dict(), dict(SPECINV_LBFGS_LEAN2="0"), dict(SPECINV_LBFGS_DEFER="0")):
        for name in ("SPECINV_LBFGS_LEAN2", "SPECINV_LBFGS_DEFER"):
            monkeypatch.delenv(name, raising=False)
        for name, v in env.items():
            monkeypatch.setenv(name, v)
        runs.append(_run_steps(monkeypatch, True, tr, target, x0, steps, **kw))
    (l0, s0) = runs[0]
    assert s0[-1][1] > steps                                                # (more than one iteration per step ran)
    for l1, s1 in runs[1:]:
        assert l1 == l0
        for a, b in zip(s0, s1):
            assert a[1:] == b[1:]
            assert torch.equal(a[0], b[0])


def test_lean_chain_is_suspended_once_and_resumed(monkeypatch):
    """Which form ran: a fresh optimiser starts lean; the iteration after its first accepted pair suspends the chain and the step
    continues in the full form (one extra synchronisation, once per optimisation); with every pair rejected - BASELINE C5's
    tiny gradients - the chain stays lean for good."""
    monkeypatch.setenv("SPECINV_LBFGS_DEVICE", "1")
    tr, target, x0 = _device_problem("logmel")
    x = x0.clone()
    _, fg = tr.bind(x, target)
    opt = LBFGS(x, device=dev())
    opt.step(fg)
    lean, full, susp = opt.dev_iterations
    assert opt._dev and lean >= 2 and full >= 1 and susp == 1 and opt.pairs_accepted > 0
    opt.step(fg)
    assert opt.dev_iterations[0] == lean and opt.dev_iterations[2] == 1       # the memory is not empty: full form from the start
    x = x0.clone()                                                         # steps of 1e-7: y.s never passes 1e-10
    _, fg = tr.bind(x, target)
    opt = LBFGS(x, device=dev(), lr=1e-7)
    for _ in range(3):
        opt.step(fg)
    lean, full, susp = opt.dev_iterations
    # (iterations ENQUEUED in each form: a step enqueues up to max_iter of them, what follows a break runs as no-ops)
    assert lean >= opt.total_iters > 0 and full == 0 and susp == 0 and opt.pairs_accepted == 0 and opt.pairs_rejected > 0


def test_l_bfgs_strong_wolfe_golden():
    """The reference's L_BFGS with line_search_fn='strong_wolfe' on the g6 fixtures (torch_specinv/methods.py:509-569 + torch.optim.LBFGS,
    run by tests/golden/make_golden.py): the log-mel problem's first outer step - iterate within 2e-2, loss within 5e-3 (the
    tolerances the oracle itself is held to, test_oracle_golden.py) - and the magnitude problem's first recorded loss (its
    trajectory cannot be pinned: the cubic interpolation's discriminant flips on 1-ulp loss differences in float32)."""
    from spectrogram_inversion_amd.metrics import _sums
    g = load_golden("g6_lbfgs")
    fb = si.mel_filterbank(22050, 2048, 80)
    tr = LogMelSTFT(torch.from_numpy(fb), 2048, hop_length=512, window=torch.from_numpy(hann(2048)))
    x = si.L_BFGS(T(g["mel_target"]), tr, init_x0=T(g["mel_x"]), outer_max_iter=1, tol=0, eva_iter=1, verbose=False,
                  line_search_fn="strong_wolfe")
    assert rel_l2(N(x), g["mel_x_wolfe"]) < 2e-2
    s = _sums(tr(x), T(g["mel_target"]))
    assert abs(s[0] / s[3] - g["mel_trace_wolfe"][0, 1]) < 5e-3 * g["mel_trace_wolfe"][0, 1]
    x = si.L_BFGS(T(g["mag_spec"]), MagSTFT(256), init_x0=T(g["mag_x0"]), outer_max_iter=1, tol=0, eva_iter=1, verbose=False,
                  max_iter=10, line_search_fn="strong_wolfe")
    s = _sums(MagSTFT(256)(x), T(g["mag_spec"]))
    assert abs(s[0] / s[3] - g["mag_trace_wolfe"][0, 1]) < 0.1 * g["mag_trace_wolfe"][0, 1]
