"""L-BFGS path of the oracle (test infrastructure only).

The reference's `L_BFGS` (torch_specinv/methods.py:509-569) minimises
`mean((transform_fn(x) - spec)**2)` over the waveform with the THIRD-PARTY
optimiser `torch.optim.LBFGS` (imported at methods.py:6, constructed at :543).
That optimiser is not part of the reference checkout; its version is not
pinned by the reference (`requirements.txt:1` says torch>=1.6.0) - the copy
installed beside the reference in the build container is torch 2.10.0
(`torch/optim/lbfgs.py`).  This file restates the published algorithm
(limited-memory BFGS two-loop recursion with the minFunc conventions torch
documents: first step length min(1, 1/|g|_1)*lr, curvature guard y.s > 1e-10,
initial Hessian scale y.s / y.y, optional strong-Wolfe line search with cubic
interpolation) and is pinned against trajectories recorded from the real
optimiser (tests/golden/lbfgs_*.npz).

Because NumPy has no autograd, `transform_fn` is replaced here by objects that
provide `forward(x)` and `loss_grad(x, target)` analytically:
`MagStft` (|STFT|, the transform of test/test_lbfgs.py:17-18) and
`LogMelStft` (log1p(M @ |STFT|), BASELINE config 5).  The analytic backward is
the formula of SURVEY 8a (verified against autograd to 3e-16).
"""
from __future__ import annotations

import numpy as np
import scipy.fft as sfft

from . import metrics as _metrics
from . import stftlib as _stft
from .methods import training_loop
from .stftlib import StftArgs


# --------------------------------------------------------------------------- #
# transforms with analytic gradients                                           #
# --------------------------------------------------------------------------- #
def _stft_adjoint(g_spec: np.ndarray, a: StftArgs, length: int) -> np.ndarray:
    """Adjoint of `stft` w.r.t. the real signal.

    g_spec: (B, F, T) complex cotangent dL/dRe + i dL/dIm of the onesided
    spectrum.  Returns dL/dx of shape (B, length).
    """
    n = a.n_fft
    g = np.swapaxes(g_spec, 1, 2)                       # (B, T, F)
    if a.onesided:
        # frame_grad[j] = Re sum_{k=0}^{N/2} G[k] e^{+2 pi i k j / N} = N * irfft(H)
        # with the interior bins halved (DC and Nyquist kept); n_fft is even
        # whenever onesided (args_helper derives it as 2*(F-1)).
        h = g.copy()
        h[..., 1:n // 2] *= 0.5
        fr = sfft.irfft(h, n=n, axis=-1) * n
    else:
        fr = sfft.ifft(g, n=n, axis=-1).real * n
    if a.normalized:
        fr = fr / np.sqrt(n)
    fr = (fr * a.window).astype(g_spec.real.dtype)
    gp = _stft.overlap_add(fr, a.hop_length, 0)         # gradient w.r.t. padded signal
    if not a.center:
        out = np.zeros((g.shape[0], length), dtype=fr.dtype)
        out[:, :gp.shape[1]] = gp[:, :length]
        return out
    p = n // 2
    lp = length + 2 * p
    full = np.zeros((g.shape[0], lp), dtype=fr.dtype)
    full[:, :gp.shape[1]] = gp[:, :lp]
    out = full[:, p:p + length].copy()
    i = np.arange(p)
    if a.pad_mode == "reflect":
        # left margin sample i mirrors x[p - i]; right margin sample p+L+i mirrors x[L-2-i]
        np.add.at(out, (slice(None), p - i), full[:, i])
        np.add.at(out, (slice(None), length - 2 - i), full[:, p + length + i])
    elif a.pad_mode == "replicate":
        out[:, 0] += full[:, :p].sum(1)
        out[:, -1] += full[:, p + length:].sum(1)
    elif a.pad_mode == "circular":
        np.add.at(out, (slice(None), (i - p) % length), full[:, i])
        np.add.at(out, (slice(None), i % length), full[:, p + length + i])
    elif a.pad_mode == "constant":
        pass
    else:
        raise ValueError(a.pad_mode)
    return out


class MagStft:
    """V = |STFT(x)|, x: (B, L) -> (B, F, T).  (test/test_lbfgs.py:17-18)"""

    def __init__(self, a: StftArgs):
        self.a = a

    def forward(self, x):
        return np.abs(_stft.stft(x, self.a))

    def loss_grad(self, x, target):
        s = _stft.stft(x, self.a)
        v = np.abs(s)
        d = v - target
        # `float(loss)` of a tensor in the parameter dtype: round to that dtype
        loss = float(x.dtype.type(np.sum(d.astype(np.float64) ** 2) / d.size))
        dv = (2.0 / d.size) * d
        with np.errstate(divide="ignore", invalid="ignore"):
            unit = np.where(v > 0, s / v, 0)
        g = _stft_adjoint((dv * unit).astype(s.dtype), self.a, x.shape[-1])
        return loss, g.astype(x.dtype)


class LogMelStft:
    """V = log1p(M @ |STFT(x)|) with M (n_mels, F).  BASELINE config 5."""

    def __init__(self, a: StftArgs, mel_fb: np.ndarray):
        self.a = a
        self.m = mel_fb

    def forward(self, x):
        mag = np.abs(_stft.stft(x, self.a))
        return np.log1p(np.einsum("mf,bft->bmt", self.m, mag).astype(x.dtype))

    def loss_grad(self, x, target):
        s = _stft.stft(x, self.a)
        mag = np.abs(s)
        mm = np.einsum("mf,bft->bmt", self.m, mag).astype(x.dtype)
        v = np.log1p(mm)
        d = v - target
        loss = float(x.dtype.type(np.sum(d.astype(np.float64) ** 2) / d.size))
        dm = (2.0 / d.size) * d / (1 + mm)
        da = np.einsum("mf,bmt->bft", self.m, dm).astype(x.dtype)
        with np.errstate(divide="ignore", invalid="ignore"):
            unit = np.where(mag > 0, s / mag, 0)
        g = _stft_adjoint((da * unit).astype(s.dtype), self.a, x.shape[-1])
        return loss, g.astype(x.dtype)


# --------------------------------------------------------------------------- #
# strong-Wolfe line search                                                     #
# --------------------------------------------------------------------------- #
def _cubic_min(xa, fa, ga, xb, fb, gb, bounds=None):
    """Minimiser of the cubic through (xa, fa, ga), (xb, fb, gb), clipped."""
    lo, hi = bounds if bounds is not None else ((xa, xb) if xa <= xb else (xb, xa))
    d1 = ga + gb - 3 * (fa - fb) / (xa - xb)
    disc = d1 * d1 - ga * gb
    if disc < 0:
        return (lo + hi) / 2.0
    d2 = np.sqrt(disc)
    if xa <= xb:
        pos = xb - (xb - xa) * ((gb + d2 - d1) / (gb - ga + 2 * d2))
    else:
        pos = xa - (xa - xb) * ((ga + d2 - d1) / (ga - gb + 2 * d2))
    return min(max(pos, lo), hi)


def _strong_wolfe(phi, t, d, f0, g0, gtd0, c1=1e-4, c2=0.9, tol_change=1e-9, max_ls=25):
    """Bracketing + zoom line search.  `phi(t)` returns (loss, grad) at x + t d.

    Scalar types follow the optimiser being restated: losses are Python floats
    (double), while step lengths and directional derivatives are scalars of the
    parameter dtype (0-dim tensors there, NumPy scalars here), so the cubic
    interpolation runs in float32 for float32 problems - its cancellation makes
    that visible in the trial steps."""
    dt = d.dtype.type
    d_norm = np.abs(d).max()
    f_new, g_new = phi(t)
    evals = 1
    gtd_new = dt(g_new @ d)
    t_prev, f_prev, g_prev, gtd_prev = 0, f0, g0, gtd0
    done = False
    it = 0
    br = None
    while it < max_ls:
        if f_new > f0 + c1 * t * gtd0 or (it > 1 and f_new >= f_prev):
            br = [[t_prev, f_prev, g_prev, gtd_prev], [t, f_new, g_new, gtd_new]]
            break
        if abs(gtd_new) <= -c2 * gtd0:
            br = [[t, f_new, g_new, gtd_new]]
            done = True
            break
        if gtd_new >= 0:
            br = [[t_prev, f_prev, g_prev, gtd_prev], [t, f_new, g_new, gtd_new]]
            break
        lo_step = t + 0.01 * (t - t_prev)
        hi_step = t * 10
        t_next = _cubic_min(t_prev, f_prev, gtd_prev, t, f_new, gtd_new, (lo_step, hi_step))
        t_prev, f_prev, g_prev, gtd_prev = t, f_new, g_new, gtd_new
        t = t_next
        f_new, g_new = phi(t)
        evals += 1
        gtd_new = dt(g_new @ d)
        it += 1
    if it == max_ls:
        br = [[0, f0, g0, gtd0], [t, f_new, g_new, gtd_new]]

    stalled = False
    lo, hi = (0, 1) if br[0][1] <= br[-1][1] else (1, 0)
    while not done and it < max_ls:
        if abs(br[1][0] - br[0][0]) * d_norm < tol_change:
            break
        t = _cubic_min(br[0][0], br[0][1], br[0][3], br[1][0], br[1][1], br[1][3])
        bmax, bmin = max(br[0][0], br[1][0]), min(br[0][0], br[1][0])
        eps = 0.1 * (bmax - bmin)
        if min(bmax - t, t - bmin) < eps:
            if stalled or t >= bmax or t <= bmin:
                t = bmax - eps if abs(t - bmax) < abs(t - bmin) else bmin + eps
                stalled = False
            else:
                stalled = True
        else:
            stalled = False
        f_new, g_new = phi(t)
        evals += 1
        gtd_new = dt(g_new @ d)
        it += 1
        if f_new > f0 + c1 * t * gtd0 or f_new >= br[lo][1]:
            br[hi] = [t, f_new, g_new, gtd_new]
            lo, hi = (0, 1) if br[0][1] <= br[1][1] else (1, 0)
        else:
            if abs(gtd_new) <= -c2 * gtd0:
                done = True
            elif gtd_new * (br[hi][0] - br[lo][0]) >= 0:
                br[hi] = list(br[lo])
            br[lo] = [t, f_new, g_new, gtd_new]
    if len(br) == 1:
        lo = 0
    return br[lo][1], br[lo][2], br[lo][0], evals


# --------------------------------------------------------------------------- #
# the optimiser                                                                #
# --------------------------------------------------------------------------- #
class LbfgsState:
    def __init__(self):
        self.n_iter = 0
        self.func_evals = 0
        self.d = None
        self.t = None
        self.ys_hist = []      # y vectors ("old_dirs")
        self.s_hist = []       # s vectors ("old_stps")
        self.rho = []
        self.h_diag = 1.0
        self.prev_grad = None
        self.prev_loss = None


def lbfgs_minimize(fg, x, state: LbfgsState, lr=1.0, max_iter=20, max_eval=None,
                   tolerance_grad=1e-7, tolerance_change=1e-9, history_size=100,
                   line_search_fn=None):
    """One `optimizer.step(closure)` (methods.py:553).  `fg(x_flat)` returns
    (loss, grad_flat); `x` is a flat float array updated in place.  Returns the
    loss of the first evaluation, like torch's `step`."""
    if max_eval is None:
        max_eval = max_iter * 5 // 4
    dt = x.dtype.type
    loss, g = fg(x)
    first_loss = loss
    evals = 1
    state.func_evals += 1
    if np.abs(g).max() <= tolerance_grad:
        return first_loss
    d, t = state.d, state.t
    n_iter = 0
    while n_iter < max_iter:
        n_iter += 1
        state.n_iter += 1
        if state.n_iter == 1:
            d = -g
            state.ys_hist, state.s_hist, state.rho = [], [], []
            state.h_diag = dt(1)
        else:
            y = g - state.prev_grad
            s = d * dt(t)
            ys = dt(y @ s)
            if ys > 1e-10:
                if len(state.ys_hist) == history_size:
                    state.ys_hist.pop(0)
                    state.s_hist.pop(0)
                    state.rho.pop(0)
                state.ys_hist.append(y)
                state.s_hist.append(s)
                state.rho.append(dt(1.0) / ys)
                state.h_diag = ys / dt(y @ y)
            m = len(state.ys_hist)
            al = [None] * m
            q = -g
            for i in range(m - 1, -1, -1):
                al[i] = dt(state.s_hist[i] @ q) * state.rho[i]
                q = q - al[i] * state.ys_hist[i]
            r = q * state.h_diag
            for i in range(m):
                be = dt(state.ys_hist[i] @ r) * state.rho[i]
                r = r + (al[i] - be) * state.s_hist[i]
            d = r.astype(x.dtype)
        state.prev_grad = g.copy()
        state.prev_loss = loss
        if state.n_iter == 1:
            t = min(1.0, dt(1.0) / np.abs(g).sum(dtype=x.dtype)) * lr
        else:
            t = lr
        gtd = dt(g @ d)
        if gtd > -tolerance_change:
            break
        ls_evals = 0
        if line_search_fn is not None:
            if line_search_fn != "strong_wolfe":
                raise RuntimeError("only 'strong_wolfe' is supported")
            x0 = x.copy()

            def phi(step):
                return fg(x0 + dt(step) * d)

            # (torch.optim.LBFGS.step does not forward tolerance_change: the search keeps its default 1e-9)
            loss, g, t, ls_evals = _strong_wolfe(
                phi, t, d, loss, g, gtd, max_ls=max_eval - evals)
            x += dt(t) * d
            opt = np.abs(g).max() <= tolerance_grad
        else:
            x += dt(t) * d
            opt = False
            if n_iter != max_iter:
                loss, g = fg(x)
                opt = np.abs(g).max() <= tolerance_grad
                ls_evals = 1
        evals += ls_evals
        state.func_evals += ls_evals
        if n_iter == max_iter:
            break
        if evals >= max_eval:
            break
        if opt:
            break
        if np.abs(d * dt(t)).max() <= tolerance_change:
            break
        if abs(loss - state.prev_loss) < tolerance_change:
            break
    state.d, state.t = d, t
    return first_loss


def l_bfgs(spec, transform, samples=None, init_x0=None, outer_max_iter=1000, tol=1e-6,
           eva_iter=10, metric="sc", trace=None, rng=None, **kwargs):
    """methods.py:509-569 with `transform` = MagStft / LogMelStft instance."""
    spec = np.asarray(spec)
    if init_x0 is None:
        rng = rng or np.random.default_rng(0)
        init_x0 = (rng.standard_normal(tuple(samples)) * 1e-6).astype(spec.dtype)  # :538
    x = np.array(init_x0, copy=True)
    shape = x.shape
    x2 = x.reshape(1, -1) if x.ndim == 1 else x
    flat = x2.reshape(-1)
    state = LbfgsState()

    def fg(v):
        loss, g = transform.loss_grad(v.reshape(x2.shape), _as_target(spec, x2))
        return loss, g.reshape(-1)

    def _as_target(s, xx):
        return s[None] if (s.ndim == 2 and xx.ndim == 2) else s

    def outer():
        lbfgs_minimize(fg, flat, state, **kwargs)                     # :553
        return transform.forward(flat.reshape(x2.shape))              # :554-556

    tgt = spec[None] if spec.ndim == 2 else spec
    training_loop(outer, tgt, outer_max_iter, tol, eva_iter, metric, trace)
    return flat.reshape(shape)
