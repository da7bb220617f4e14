"""STFT / ISTFT building blocks of the oracle (test infrastructure only).

Restates, in NumPy, the helpers of the reference `torch_specinv/methods.py`:
`_args_helper` (:21-91), `_ola` (:114-132), `_istft` (:135-150) and the
`torch.stft` call sites (:241, :385, :464).  All arithmetic is carried out in
the real dtype of the input (float32 or float64), like the reference does.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import scipy.fft as sfft

_PAD_MODE = {"reflect": "reflect", "constant": "constant", "replicate": "edge",
             "circular": "wrap"}

# scipy.fft worker threads (set by bench.py's cpu_baseline leg; 1 elsewhere)
WORKERS = 1


@dataclass
class StftArgs:
    """Normalised STFT arguments (what `_args_helper` returns, methods.py:85-91)."""
    n_fft: int
    win_length: int
    hop_length: int
    window: np.ndarray            # real, length n_fft (centre-padded)
    center: bool = True
    pad_mode: str = "reflect"
    normalized: bool = False
    onesided: bool = True
    extra: dict = field(default_factory=dict)

    @property
    def n_freq(self) -> int:
        return self.n_fft // 2 + 1 if self.onesided else self.n_fft

    @property
    def padding(self) -> int:
        return self.n_fft // 2 if self.center else 0


def real_dtype(dtype) -> np.dtype:
    """methods.py:49-57 (complex -> matching real dtype)."""
    dtype = np.dtype(dtype)
    if dtype == np.complex64:
        return np.dtype(np.float32)
    if dtype == np.complex128:
        return np.dtype(np.float64)
    return dtype


def args_helper(n_freq: int, dtype, **stft_kwargs) -> StftArgs:
    """methods.py:21-91.  `n_freq` is `spec.shape[-2]`.

    Unknown kwargs are silently dropped (:42-46); `win_length`/`hop_length`
    that are falsy fall back to `n_fft` / `n_fft // 4` (:70-74); the window is
    ones(win_length) by default (:76-77) and centre-padded to n_fft with
    floor((n_fft-wl)/2) zeros on the left and ceil on the right (:80-83).
    """
    win_length = stft_kwargs.get("win_length", None)
    window = stft_kwargs.get("window", None)
    hop_length = stft_kwargs.get("hop_length", None)
    center = stft_kwargs.get("center", True)
    pad_mode = stft_kwargs.get("pad_mode", "reflect")
    normalized = stft_kwargs.get("normalized", False)
    onesided = stft_kwargs.get("onesided", None)

    rdt = real_dtype(dtype)
    if onesided is None:
        onesided = not (window is not None and np.iscomplexobj(window))
    n_fft = (n_freq - 1) * 2 if onesided else n_freq
    if not win_length:
        win_length = n_fft
    if not hop_length:
        hop_length = n_fft // 4
    if window is None:
        window = np.ones(win_length, dtype=rdt)
    window = np.asarray(window)
    assert n_fft >= win_length
    if n_fft > win_length:
        left = (n_fft - win_length) // 2
        right = (n_fft - win_length + 1) // 2
        window = np.pad(window, (left, right))
        win_length = n_fft
    return StftArgs(n_fft=int(n_fft), win_length=int(win_length),
                    hop_length=int(hop_length), window=window,
                    center=bool(center), pad_mode=str(pad_mode),
                    normalized=bool(normalized), onesided=bool(onesided))


def frame_count(length: int, a: StftArgs) -> int:
    """Number of frames torch.stft produces for a signal of `length` samples."""
    return 1 + (length + 2 * a.padding - a.n_fft) // a.hop_length


def signal_length(n_frames: int, a: StftArgs) -> int:
    """Length of the ISTFT output: conv_transpose1d size rule, methods.py:127."""
    return (n_frames - 1) * a.hop_length + a.n_fft - 2 * a.padding


def _frames(x: np.ndarray, a: StftArgs) -> np.ndarray:
    """(B, L) -> (B, T, N) strided view of the (padded) signal."""
    if a.center:
        p = a.n_fft // 2
        x = np.pad(x, ((0, 0), (p, p)), mode=_PAD_MODE[a.pad_mode])
    n_frames = 1 + (x.shape[-1] - a.n_fft) // a.hop_length
    sb, sl = x.strides
    return np.lib.stride_tricks.as_strided(
        x, shape=(x.shape[0], n_frames, a.n_fft),
        strides=(sb, sl * a.hop_length, sl), writeable=False)


def stft(x: np.ndarray, a: StftArgs, window: np.ndarray | None = None) -> np.ndarray:
    """`torch.stft(x, n_fft, **processed_args)` as called at methods.py:241.

    x: (B, L) real.  Returns (B, F, T) complex with F = n_freq.
    """
    w = a.window if window is None else window
    fr = _frames(x, a) * w.astype(x.dtype, copy=False)
    norm = "ortho" if a.normalized else "backward"
    if a.onesided:
        s = sfft.rfft(fr, n=a.n_fft, axis=-1, norm=norm, workers=WORKERS)
    else:
        s = sfft.fft(fr, n=a.n_fft, axis=-1, norm=norm, workers=WORKERS)
    return np.ascontiguousarray(np.swapaxes(s, 1, 2))


def inverse_frames(spec: np.ndarray, a: StftArgs) -> np.ndarray:
    """Per-frame inverse FFT, methods.py:141-146.  (B, F, T) -> (B, T, N) real."""
    norm = "ortho" if a.normalized else "backward"
    s = np.swapaxes(spec, 1, 2)
    if a.onesided:
        return sfft.irfft(s, n=a.n_fft, axis=-1, norm=norm, workers=WORKERS)
    return sfft.ifft(s, n=a.n_fft, axis=-1, norm=norm, workers=WORKERS).real


def overlap_add(frames: np.ndarray, hop: int, padding: int) -> np.ndarray:
    """`F.conv_transpose1d(x, diag(w), stride=hop, padding=padding)` without the
    window (methods.py:127): y[b, t*hop + k - padding] += frames[b, t, k].

    frames: (B, T, N).  Output (B, (T-1)*hop + N - 2*padding).
    """
    b, t, n = frames.shape
    full = np.zeros((b, (t - 1) * hop + n), dtype=frames.dtype)
    if n % hop == 0:
        # N/hop interleaved, non-overlapping sets of frames
        r = n // hop
        seg = frames.reshape(b, t, r, hop)
        for q in range(r):
            # segment q of frame t lands on hop-block t + q
            full.reshape(b, -1, hop)[:, q:q + t, :] += seg[:, :, q, :]
    else:
        for i in range(t):
            full[:, i * hop:i * hop + n] += frames[:, i, :]
    return full[:, padding:full.shape[1] - padding] if padding else full


def ola_envelope(n_frames: int, a: StftArgs, dtype=np.float32,
                 window: np.ndarray | None = None) -> np.ndarray:
    """Window-square envelope, methods.py:129-131: e[n] = sum_t w^2[n + p - t*hop]."""
    w = (a.window if window is None else window).astype(dtype, copy=False)
    fr = np.broadcast_to((w * w)[None, None, :], (1, n_frames, a.n_fft))
    return overlap_add(np.ascontiguousarray(fr), a.hop_length, a.padding)[0]


def istft(spec: np.ndarray, a: StftArgs, envelope: np.ndarray | None = None,
          window: np.ndarray | None = None):
    """`_istft` + `_ola`, methods.py:114-150.  Returns (x / e, e); no zero guard
    on the envelope, exactly like the reference (:132)."""
    fr = inverse_frames(spec, a)
    rdt = fr.dtype
    w = (a.window if window is None else window).astype(rdt, copy=False)
    y = overlap_add(fr * w, a.hop_length, a.padding)
    if envelope is None:
        envelope = ola_envelope(spec.shape[-1], a, dtype=rdt, window=window)
    with np.errstate(divide="ignore", invalid="ignore"):
        return y / envelope, envelope
