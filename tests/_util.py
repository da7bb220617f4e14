"""Shared helpers for the parity tests (fixture decoding, tolerances)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def hann(n, dtype=np.float32):
    return (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n) / n)).astype(dtype)


def rel_l2(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    a = a.astype(np.complex128 if np.iscomplexobj(a) else np.float64)
    b = b.astype(np.complex128 if np.iscomplexobj(b) else np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def segment_errors(y, ref, seg):
    """Per-segment error of a waveform, relative to the reference's RMS segment energy.  Griffin-Lim with
    inconsistent magnitudes is locally chaotic: when a bin of some frame passes close to zero the projection
    S*m/|S| is discontinuous there and two float32 implementations (or the reference in float32 and float64)
    part ways in that neighbourhood only.  A robust gate looks at the distribution over segments."""
    y, ref = np.asarray(y, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    y, ref = y.reshape(-1, y.shape[-1]), ref.reshape(-1, ref.shape[-1])
    n = (y.shape[1] // seg) * seg
    e = ((y - ref)[:, :n].reshape(y.shape[0], -1, seg) ** 2).sum(-1)
    r = (ref[:, :n].reshape(ref.shape[0], -1, seg) ** 2).sum(-1)
    return np.sqrt(e / r.mean()).reshape(-1)


def sc_linear(db):
    return 10.0 ** (np.asarray(db, dtype=np.float64) / 20.0)


def sweep_kwargs(meta_row, dtype=np.float32):
    """Decode one row of g3_sweep's `meta` into stft kwargs (see make_golden.py)."""
    wl, wname, hop, center, normalized, onesided, pad = str(meta_row).split("|")
    kw = dict(win_length=None if wl == "None" else int(wl),
              hop_length=None if hop == "None" else int(hop),
              center=bool(int(center)), normalized=bool(int(normalized)),
              onesided=bool(int(onesided)), pad_mode=pad)
    kw["window"] = hann(300, dtype) if wname == "hann300" else None
    return kw


G0_CASES = [
    dict(n_fft=256, hop_length=64, window="hann", center=True, pad_mode="reflect",
         normalized=False, onesided=True),
    dict(n_fft=256, hop_length=100, window="rect", center=True, pad_mode="constant",
         normalized=True, onesided=True),
    dict(n_fft=128, hop_length=32, window="hann", center=False, pad_mode="reflect",
         normalized=False, onesided=False),
    dict(n_fft=256, hop_length=64, window="hann200", center=True, pad_mode="replicate",
         normalized=False, onesided=True),
    dict(n_fft=256, hop_length=64, window="hann", center=True, pad_mode="circular",
         normalized=True, onesided=False),
]


def g0_kwargs(c):
    kw = dict(hop_length=c["hop_length"], center=c["center"], pad_mode=c["pad_mode"],
              normalized=c["normalized"], onesided=c["onesided"])
    if c["window"] == "hann":
        kw["window"] = hann(c["n_fft"])
    elif c["window"] == "hann200":
        kw["window"] = hann(200)
        kw["win_length"] = 200
    return kw


def finite_close(a, b, rtol, atol=0.0):
    """allclose on the finite entries + identical non-finite pattern (the reference divides
    by a zero envelope when center=False with a Hann window: methods.py:132)."""
    a = np.asarray(a)
    b = np.asarray(b)
    fa, fb = np.isfinite(a), np.isfinite(b)
    if not np.array_equal(fa, fb):
        return False
    if not fa.any():
        return True
    scale = np.abs(b[fb]).max()
    return bool(np.all(np.abs(a[fa] - b[fb]) <= atol + rtol * max(scale, 1e-30)))
