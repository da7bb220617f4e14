"""Slaney-style mel filterbank (host-side utility for the L_BFGS / log-mel path).

The reference's README builds its mel example with `librosa.filters.mel`
(README.md:60-63), which is not installable here; this is an independent
implementation of the same published formula (Slaney's Auditory Toolbox mel
scale: linear below 1 kHz at 200/3 Hz per mel, logarithmic above with
log(6.4)/27 per mel; triangular filters normalised to unit area).
"""
from __future__ import annotations

import math

import numpy as np

_F_SP = 200.0 / 3.0
_MIN_LOG_HZ = 1000.0
_MIN_LOG_MEL = _MIN_LOG_HZ / _F_SP
_LOGSTEP = math.log(6.4) / 27.0


def hz_to_mel(f, htk: bool = False):
    f = np.asarray(f, dtype=np.float64)
    if htk:
        return 2595.0 * np.log10(1.0 + f / 700.0)
    lin = f / _F_SP
    with np.errstate(divide="ignore", invalid="ignore"):
        log = _MIN_LOG_MEL + np.log(np.maximum(f, 1e-300) / _MIN_LOG_HZ) / _LOGSTEP
    return np.where(f >= _MIN_LOG_HZ, log, lin)


def mel_to_hz(m, htk: bool = False):
    m = np.asarray(m, dtype=np.float64)
    if htk:
        return 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    lin = m * _F_SP
    log = _MIN_LOG_HZ * np.exp(_LOGSTEP * (m - _MIN_LOG_MEL))
    return np.where(m >= _MIN_LOG_MEL, log, lin)


def mel_filterbank(sr: int = 22050, n_fft: int = 2048, n_mels: int = 80,
                   fmin: float = 0.0, fmax: float | None = None, htk: bool = False,
                   norm: str | None = "slaney", dtype=np.float32) -> np.ndarray:
    """(n_mels, n_fft//2 + 1) triangular filterbank.  `htk`: the HTK mel scale instead of Slaney's; `norm="slaney"`
    (default) scales every filter to unit area, `None` leaves the peaks at 1 - the options of the function the
    reference's README uses (its default there is 128 bins; BASELINE's log-mel configuration has 80)."""
    if fmax is None:
        fmax = sr / 2.0
    assert norm in (None, "slaney")
    n_freq = n_fft // 2 + 1
    fft_f = np.linspace(0.0, sr / 2.0, n_freq)
    edges = mel_to_hz(np.linspace(hz_to_mel(fmin, htk), hz_to_mel(fmax, htk), n_mels + 2), htk)
    width = np.diff(edges)
    ramps = edges[:, None] - fft_f[None, :]
    fb = np.zeros((n_mels, n_freq), dtype=np.float64)
    for i in range(n_mels):
        rising = -ramps[i] / width[i]
        falling = ramps[i + 2] / width[i + 1]
        fb[i] = np.maximum(0.0, np.minimum(rising, falling))
    if norm == "slaney":
        fb *= (2.0 / (edges[2:n_mels + 2] - edges[:n_mels]))[:, None]
    return fb.astype(dtype)
