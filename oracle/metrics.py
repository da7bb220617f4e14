"""Scalar metrics of the oracle (test infrastructure only).

Restates `torch_specinv/metrics.py`: `sc` (:4-14), `snr` (:17-29), `ser`
(:32-43), and the `F.mse_loss` used by `_training_loop` (methods.py:174,182).
All are whole-tensor (whole-batch) scalars.  Accumulation is done in float64,
which is at least as accurate as the reference's float32 pairwise sums.
"""
import numpy as np


def _sumsq(a) -> float:
    a = np.asarray(a, dtype=np.float64)
    return float(np.sum(a * a))


def sc(inp, target) -> float:
    """20*(log10||inp - target|| - log10||target||), metrics.py:14 (dB)."""
    d = np.asarray(inp, dtype=np.float64) - np.asarray(target, dtype=np.float64)
    with np.errstate(divide="ignore"):
        return float(20.0 * (np.log10(np.sqrt(_sumsq(d))) - np.log10(np.sqrt(_sumsq(target)))))


def snr(inp, target) -> float:
    """-10*log10(sum((inp/||t|| - t/||t||)^2)), metrics.py:28-29."""
    nrm = np.sqrt(_sumsq(target))
    d = np.asarray(inp, dtype=np.float64) / nrm - np.asarray(target, dtype=np.float64) / nrm
    with np.errstate(divide="ignore"):
        return float(-10.0 * np.log10(_sumsq(d)))


def ser(inp, target) -> float:
    """10*(log10 sum(inp^2) - log10 sum((inp - target)^2)), metrics.py:43."""
    d = np.asarray(inp, dtype=np.float64) - np.asarray(target, dtype=np.float64)
    with np.errstate(divide="ignore"):
        return float(10.0 * (np.log10(_sumsq(inp)) - np.log10(_sumsq(d))))


def mse(inp, target) -> float:
    """F.mse_loss(inp, target) (mean reduction), methods.py:174,182."""
    d = np.asarray(inp, dtype=np.float64) - np.asarray(target, dtype=np.float64)
    return _sumsq(d) / d.size


FUNCS = {"SC": sc, "SNR": snr, "SER": ser}
