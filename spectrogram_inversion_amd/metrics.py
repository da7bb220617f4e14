"""Whole-tensor metrics of the reference (`torch_specinv/metrics.py`) on the HIP device.

`sc` (metrics.py:4-14, dB), `snr` (:17-29), `ser` (:32-43).  The three sums they need
(sum((a-b)^2), sum(a^2), sum(b^2)) come from one deterministic float64 reduction kernel.
"""
from __future__ import annotations

import math

import torch

from .plan import StftArgs, get_plan, require_gpu

__all__ = ["sc", "snr", "ser"]


def _util_plan(dtype, device):
    args = StftArgs(2, 2, 1, torch.ones(2, dtype=dtype), False, "reflect", False, True)
    return get_plan(args, 1, 1, dtype, device)


def _sums(a: torch.Tensor, b: torch.Tensor):
    assert a.shape == b.shape
    device = require_gpu(a.device)
    return _util_plan(a.dtype, device).metric_sums(a.reshape(-1), b.reshape(-1))


def _log10(v):
    return math.log10(v) if v > 0 else (-math.inf if v == 0 else math.nan)


def _from_sums(name: str, s) -> float:
    if name == "SC":                                           # metrics.py:14
        return 20.0 * (_log10(math.sqrt(s[0])) - _log10(math.sqrt(s[2])))
    if name == "SNR":                                          # metrics.py:28-29
        return -10.0 * _log10(s[0] / s[2]) if s[2] > 0 else math.nan
    if name == "SER":                                          # metrics.py:43
        return 10.0 * (_log10(s[1]) - _log10(s[0]))
    raise AssertionError(name)


def _metric(name, input, target):
    val = _from_sums(name, _sums(input, target))
    return torch.tensor(val, dtype=input.dtype, device=input.device)


def sc(input, target):
    """Spectral convergence in dB: 20*(log10||input-target|| - log10||target||)."""
    return _metric("SC", input, target)


def snr(input, target):
    """-10*log10(sum((input/||target|| - target/||target||)^2))."""
    return _metric("SNR", input, target)


def ser(input, target):
    """10*(log10(sum(input^2)) - log10(sum((input-target)^2)))."""
    return _metric("SER", input, target)
