"""Differentiable Griffin-Lim / ADMM / RTISI-LA: d(waveform)/d(spectrogram) like the reference's autograd path
(`spec.requires_grad=True`, test/test_griffin.py:54,65-66; README.md:8-9 advertises use inside training).

The forward pass records the per-iteration spectra and is assembled from libspecinv's building blocks
(STFT, element-wise update, ISTFT); the backward pass runs the hand-written adjoints of those blocks in
reverse (`specinv_istft_adjoint`, `specinv_gla_update_adjoint`, `specinv_stft_adjoint`,
`specinv_phase_init_adjoint`).  It is only taken when a gradient is actually requested - the inference path
stays on the fused kernel.
"""
from __future__ import annotations

import torch

from . import _lib
from .metrics import _from_sums


class _GriffinLimFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, spec3, plan, alpha, max_iter, tol, eva_iter, metric, on_eval):
        real_in = not spec3.is_complex()
        if real_in:
            mag = spec3.detach().contiguous()
            c0 = plan.phase_init(mag)                                  # methods.py:106
        else:
            c0 = spec3.detach().contiguous()
            mag = c0.abs().contiguous()                                # methods.py:110
        lr = alpha / (1.0 + alpha)                                     # :235
        p = c0
        x = plan.istft(c0)                                             # :233
        saved = []
        name = metric.upper()
        init_loss = previous = None
        for i in range(max_iter):                                      # _training_loop, :178
            r = plan.stft(x)                                           # closure, :241
            s_k, q = plan.gla_update(r, p, mag, lr)                    # :243-247
            saved.append(s_k)
            p = s_k
            x = plan.istft(q)                                          # :248
            if i % eva_iter == eva_iter - 1:                           # :180-190, on this call's |STFT|
                s = plan.metric_sums(r.abs(), mag)
                m_val, loss = _from_sums(name, s), s[0] / s[3]
                if on_eval is not None:
                    on_eval(i, m_val, loss)
                if not init_loss:
                    init_loss = loss
                elif (previous - loss) / init_loss < tol and previous > loss:
                    break
                previous = loss
        ctx.plan, ctx.lr, ctx.real_in = plan, lr, real_in
        # the recorded spectra are ordinary saved tensors: autograd's hooks and release rules apply to them, and
        # backward may run again under retain_graph=True like on the reference's op graph
        ctx.save_for_backward(mag, c0, *saved)
        return x

    @staticmethod
    def backward(ctx, g_y):
        plan, lr = ctx.plan, ctx.lr
        mag, c0, *spectra = ctx.saved_tensors
        gx = g_y.detach().to(plan.dtype).contiguous()
        gm = torch.zeros_like(mag)
        gp = None
        for s_k in reversed(spectra):
            gq = plan.istft_adjoint(gx)
            gr, gp = plan.gla_update_adjoint(gq, gp, s_k, mag, lr, gm)
            gx = plan.stft_adjoint(gr, plan.length)
        gc = plan.istft_adjoint(gx)                                    # x0 = ISTFT(C0)
        if gp is not None:
            gc = gc + gp                                               # pre_spec_0 = C0
        grad = _input_grad(ctx, plan, mag, c0, gc, gm)
        return grad, None, None, None, None, None, None, None


def _input_grad(ctx, plan, mag, c0, gc, gm):
    """Chain the cotangents of C0 (complex) and of the target magnitude back to the user's `spec`."""
    if ctx.real_in:
        plan.phase_init_adjoint(mag, gc.contiguous(), gm)              # C0 = phase_init(mag)
        return gm
    absc = c0.abs()
    unit = torch.where(absc > 0, c0 / absc, torch.zeros_like(c0))
    return gc + gm * unit                                              # target = |C0|


class _ADMMFn(torch.autograd.Function):
    """Same scheme for `ADMM` (methods.py:458-479): state (x, X, U); the pre-projection value V is what the
    projection's adjoint needs, so that is what gets recorded."""

    @staticmethod
    def forward(ctx, spec3, plan, rho, max_iter, tol, eva_iter, metric, on_eval):
        real_in = not spec3.is_complex()
        if real_in:
            mag = spec3.detach().contiguous()
            c0 = plan.phase_init(mag)
        else:
            c0 = spec3.detach().contiguous()
            mag = c0.abs().contiguous()
        X, U = c0, torch.zeros_like(c0)                                # :458-460
        x = plan.istft(c0)                                             # :461
        saved = []
        name = metric.upper()
        init_loss = previous = None
        for i in range(max_iter):
            r = plan.stft(x)                                           # :468
            X, U, v_k, y_n = plan.admm_update(r, X, U, mag, rho)       # :469-474
            saved.append(v_k)
            x = plan.istft(y_n)                                        # :475
            if i % eva_iter == eva_iter - 1:
                s = plan.metric_sums(r.abs(), mag)
                m_val, loss = _from_sums(name, s), s[0] / s[3]
                if on_eval is not None:
                    on_eval(i, m_val, loss)
                if not init_loss:
                    init_loss = loss
                elif (previous - loss) / init_loss < tol and previous > loss:
                    break
                previous = loss
        ctx.plan, ctx.rho, ctx.real_in = plan, rho, real_in
        ctx.save_for_backward(mag, c0, *saved)
        return x

    @staticmethod
    def backward(ctx, g_y):
        plan, rho = ctx.plan, ctx.rho
        mag, c0, *spectra = ctx.saved_tensors
        gx = g_y.detach().to(plan.dtype).contiguous()
        gm = torch.zeros_like(mag)
        gX = gU = None
        for v_k in reversed(spectra):
            gyn = plan.istft_adjoint(gx)
            gr, gX, gU = plan.admm_update_adjoint(gyn, gX, gU, v_k, mag, rho, gm)
            gx = plan.stft_adjoint(gr, plan.length)
        gc = plan.istft_adjoint(gx)                                    # x0 = ISTFT(C0)
        if gX is not None:
            gc = gc + gX                                               # X0 = C0 (U0 = 0 is a constant)
        return _input_grad(ctx, plan, mag, c0, gc, gm), None, None, None, None, None, None, None


class _RtisiFn(torch.autograd.Function):
    """`RTISI_LA` (methods.py:363-404): the recorded run and the backward sweep are one persistent kernel each."""

    @staticmethod
    def forward(ctx, spec3, plan, look_ahead, asym, max_iter, alpha):
        mag = spec3.detach().contiguous()
        x, rec = plan.rtisi_recorded(mag, look_ahead, asym, max_iter, alpha)
        ctx.plan, ctx.cfg = plan, (look_ahead, asym, max_iter, alpha)
        ctx.save_for_backward(mag, rec)
        return x

    @staticmethod
    def backward(ctx, g_y):
        mag, rec = ctx.saved_tensors
        gm = ctx.plan.rtisi_adjoint(mag, rec, g_y.detach().contiguous(), *ctx.cfg)
        return gm, None, None, None, None, None


def rtisi_differentiable(spec3, plan, look_ahead, asym, max_iter, alpha):
    return _RtisiFn.apply(spec3, plan, look_ahead, asym, max_iter, alpha)


def admm_differentiable(spec3, plan, rho, max_iter, tol, eva_iter, metric, on_eval=None):
    assert isinstance(metric, str) and metric.upper() in _lib.METRICS
    return _ADMMFn.apply(spec3, plan, rho, max_iter, tol, eva_iter, metric, on_eval)


def griffin_lim_differentiable(spec3, plan, alpha, max_iter, tol, eva_iter, metric, on_eval=None):
    assert isinstance(metric, str) and metric.upper() in _lib.METRICS
    return _GriffinLimFn.apply(spec3, plan, alpha, max_iter, tol, eva_iter, metric, on_eval)
