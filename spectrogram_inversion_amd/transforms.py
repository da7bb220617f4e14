"""Device transforms for `L_BFGS`: callables `x -> representation` whose forward pass and
analytic loss gradient run as fused HIP kernels (|STFT| and log1p(mel @ |STFT|)).

They stand in for the `transform_fn` closures of the reference's examples
(README.md:55-73, test/test_lbfgs.py:17-18, main.py:21-43) on the fast path; any other
differentiable callable is still accepted by `L_BFGS` and evaluated through autograd.
"""
from __future__ import annotations

import torch

from . import _lib
from .plan import Plan, StftArgs, require_gpu


class DeviceTransform:
    kind = None

    def __init__(self, n_fft, hop_length=None, win_length=None, window=None, center=True, pad_mode="reflect",
                 normalized=False, onesided=True):
        if not win_length:
            win_length = n_fft
        if not hop_length:
            hop_length = n_fft // 4
        self.n_fft, self.hop_length, self.win_length = int(n_fft), int(hop_length), int(win_length)
        self.window = window
        self.center, self.pad_mode, self.normalized, self.onesided = bool(center), pad_mode, bool(normalized), bool(onesided)
        self._plans = {}

    def _args(self, dtype) -> StftArgs:
        w = self.window
        if w is None:
            w = torch.ones(self.win_length, dtype=dtype)
        w = w.detach().to("cpu", dtype).reshape(-1)
        assert w.numel() == self.win_length and self.n_fft >= self.win_length
        if self.n_fft > self.win_length:
            left = (self.n_fft - self.win_length) // 2
            w = torch.nn.functional.pad(w, [left, self.n_fft - self.win_length - left])
        return StftArgs(self.n_fft, self.n_fft, self.hop_length, w.contiguous(), self.center, self.pad_mode,
                        self.normalized, self.onesided)

    def _mel(self):
        return None

    def _plan(self, x2):
        """The transform's own plan for this signal shape: a plan carries the transform kind and the (tiled) filterbank
        as state, so it is never taken from the cache the iterative methods share, and it is set up once."""
        device = require_gpu(x2.device)
        key = (tuple(x2.shape), x2.dtype, str(device))
        plan = self._plans.get(key)
        if plan is None:
            args = self._args(x2.dtype)
            plan = Plan(args, x2.shape[0], args.frame_count(x2.shape[1]), x2.dtype, device)
            plan.transform_setup(self.kind, self._mel())
            if len(self._plans) >= 2:                  # a transform is normally used at one shape
                self._plans.pop(next(iter(self._plans)))
            self._plans[key] = plan
        return plan

    def __call__(self, x):
        x2 = x.reshape(1, -1) if x.dim() == 1 else x
        v = self._plan(x2.to(require_gpu(x.device))).transform_forward(x2)
        return (v[0] if x.dim() == 1 else v).to(x.device)

    def bind(self, x, target):
        """Returns (forward(x), loss_and_grad(x)) closures for the optimiser."""
        x2 = x.reshape(1, -1) if x.dim() == 1 else x
        plan = self._plan(x2)
        tgt = target.reshape(plan.batch, plan.n_out, plan.n_frames)

        def fwd(v):
            return plan.transform_forward(v.reshape(x2.shape))

        def fg(v):
            loss, g = plan.transform_loss_grad(v.reshape(x2.shape), tgt)
            return loss, g.reshape(v.shape)

        def fg_dev(v, loss_ptr):
            """gradient now, loss left in device memory at `loss_ptr` (no host synchronisation)"""
            return plan.transform_loss_grad_dev(v.reshape(x2.shape), tgt, loss_ptr).reshape(v.shape)

        def fg_dev_stats(v, d, out_ptr):
            """gradient now; {loss, g.d, sum|g|, max|g|, max|d|} left in device memory at `out_ptr` (d None: d = g)"""
            return plan.transform_loss_grad_stats_dev(v.reshape(x2.shape), tgt, None if d is None else d.reshape(x2.shape),
                                                      out_ptr).reshape(v.shape)

        fg.dev = fg_dev
        fg.dev_stats = fg_dev_stats
        # what the device-resident optimiser needs (lbfgs.py:_step_device): the plan that owns the objective, and its target
        fg.device_objective = (plan, tgt, tuple(x2.shape))
        return fwd, fg


class MagSTFT(DeviceTransform):
    """x -> |torch.stft(x, n_fft, ...)|  (B, F, T)."""
    kind = _lib.TF_MAG


class LogMelSTFT(DeviceTransform):
    """x -> log1p(mel_fb @ |torch.stft(x, n_fft, ...)|)  (B, n_mels, T)."""
    kind = _lib.TF_LOGMEL

    def __init__(self, mel_fb, n_fft, **kw):
        super().__init__(n_fft, **kw)
        self.mel_fb = mel_fb

    def _mel(self):
        return self.mel_fb
