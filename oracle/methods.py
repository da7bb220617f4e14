"""Algorithm restatements of the oracle (test infrastructure only).

NumPy versions of `phase_init` (methods.py:572-615), `_training_loop`
(:153-190), `griffin_lim` (:193-270), `ADMM` (:415-506) and `RTISI_LA`
(:273-412) of the reference `torch_specinv/methods.py`.  Inputs/outputs are
NumPy arrays with the reference's shapes: spectrograms (F, T) or (B, F, T),
waveforms (L,) or (B, L).
"""
from __future__ import annotations

import math

import numpy as np

from . import metrics as _metrics
from . import stftlib as _stft
from .stftlib import StftArgs, args_helper, istft, stft


# --------------------------------------------------------------------------- #
# phase_init                                                                   #
# --------------------------------------------------------------------------- #
def phase_init(spec: np.ndarray, **stft_kwargs) -> np.ndarray:
    """methods.py:572-615.

    Rounding recipe (verified against the reference): the peak
    offset p and omega are evaluated in the input dtype in the written order
    with 2*pi first rounded to that dtype; the time cumsum accumulates in
    float64 and each partial sum is rounded back to the input dtype (ATen's CPU
    accumulate type); exp(1j*phi) is cos/sin of the rounded phi.
    """
    assert not np.iscomplexobj(spec)
    shape = spec.shape
    if spec.ndim == 2:
        spec = spec[None]
    assert spec.ndim == 3
    dt = spec.dtype.type
    a = args_helper(spec.shape[-2], spec.dtype, **stft_kwargs)
    n_fft, hop = a.n_fft, a.hop_length

    mid, up, dn = spec[:, 1:-1], spec[:, 2:], spec[:, :-2]
    peak = (mid > up) & (mid > dn)                       # :597  strict local max in freq
    k = np.arange(1, spec.shape[1] - 1, dtype=spec.dtype)[None, :, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        p = dt(0.5) * (dn - up) / (dn - dt(2) * mid + up)            # :604
        omega = dt(2 * math.pi) * (k + p) / dt(n_fft) * dt(hop)       # :605
    omega = np.where(peak, omega, dt(0)).astype(spec.dtype)

    # :607-609 scatter order: bin k, then k-1, then k+1 (last write wins);
    # peaks are never adjacent, so bin f takes omega[f] if f is a peak, else
    # omega[f-1] if f-1 is a peak (the k+1 write, executed last), else
    # omega[f+1] if f+1 is a peak.
    phase = np.zeros_like(spec)
    pk = np.zeros(spec.shape, dtype=bool)
    pk[:, 1:-1] = peak
    om = np.zeros_like(spec)
    om[:, 1:-1] = omega
    phase[:, :-1] = np.where(pk[:, 1:], om[:, 1:], phase[:, :-1])     # k-1 write
    phase[:, 1:] = np.where(pk[:, :-1], om[:, :-1], phase[:, 1:])     # k+1 write (later)
    phase = np.where(pk, om, phase)                                    # own bin

    phi = np.cumsum(phase.astype(np.float64), axis=2).astype(spec.dtype)   # :611
    ang = np.cos(phi.astype(np.float64)).astype(spec.dtype) + \
        1j * np.sin(phi.astype(np.float64)).astype(spec.dtype)             # :612
    out = (spec * ang).astype(np.result_type(spec.dtype, np.complex64))
    return out.reshape(shape)


# --------------------------------------------------------------------------- #
# shared driver                                                                #
# --------------------------------------------------------------------------- #
def _spec_formatter(spec: np.ndarray, **stft_kwargs):
    """methods.py:99-111."""
    assert 4 > spec.ndim > 1
    if spec.ndim == 2:
        spec = spec[None]
    if not np.iscomplexobj(spec):
        return phase_init(spec, **stft_kwargs), spec
    return spec, np.abs(spec)


def training_loop(closure, target, max_iter, tol, eva_iter, metric, trace=None):
    """methods.py:153-190.  Returns the number of closure calls made.

    `trace`, if given, receives (iteration_index, metric_value, mse) tuples at
    every evaluation.
    """
    assert eva_iter > 0
    assert max_iter > 0
    assert tol >= 0
    metric = metric.upper()
    assert metric in _metrics.FUNCS
    fn = _metrics.FUNCS[metric]
    init_loss = None
    previous_loss = None
    done = 0
    for i in range(max_iter):
        output = closure()
        done = i + 1
        if i % eva_iter == eva_iter - 1:
            m = fn(output, target)
            l2 = _metrics.mse(output, target)
            if trace is not None:
                trace.append((i, m, l2))
            if not init_loss:
                init_loss = l2
            elif (previous_loss - l2) / init_loss < tol and previous_loss > l2:
                break
            previous_loss = l2
    return done


def _finish(x, spec):
    """methods.py:267-270: squeeze unless the input was exactly (1, F, T)."""
    if not (spec.ndim == 3 and spec.shape[0] == 1):
        x = x.squeeze(0) if x.shape[0] == 1 else x
    return x


# --------------------------------------------------------------------------- #
# griffin_lim                                                                  #
# --------------------------------------------------------------------------- #
def griffin_lim(spec, max_iter=200, tol=1e-6, alpha=0.99, eva_iter=10, metric="sc",
                trace=None, return_state=False, **stft_kwargs):
    """methods.py:193-270 (Griffin-Lim / Fast Griffin-Lim)."""
    assert alpha >= 0
    spec = np.asarray(spec)
    cspec, target = _spec_formatter(spec, **stft_kwargs)
    a = args_helper(target.shape[-2], target.dtype, **stft_kwargs)
    rdt = target.dtype.type

    state = {"pre": cspec.copy()}
    state["x"], env = istft(cspec, a)                                  # :233
    state["x"] = state["x"].astype(target.dtype)
    lr = rdt(alpha / (1 + alpha))                                       # :235

    def closure():
        new = stft(state["x"], a)                                      # :241
        out = np.abs(new)                                              # :242
        new = new - state["pre"] * lr                                  # :243
        state["pre"] = new                                             # :244
        norm = np.abs(new) + rdt(1e-16)                                # :246
        new = new * target / norm                                      # :247
        x, _ = istft(new, a, envelope=env)                             # :248
        state["x"] = x.astype(target.dtype)
        return out

    done = training_loop(closure, target, max_iter, tol, eva_iter, metric, trace)
    x = _finish(state["x"], spec)
    if return_state:
        return x, {"iters": done, "pre_spec": state["pre"], "envelope": env}
    return x


# --------------------------------------------------------------------------- #
# ADMM                                                                         #
# --------------------------------------------------------------------------- #
def admm(spec, max_iter=1000, tol=1e-6, rho=0.1, eva_iter=10, metric="sc",
         trace=None, return_state=False, **stft_kwargs):
    """methods.py:415-506."""
    assert eva_iter > 0 and max_iter > 0 and tol >= 0
    assert metric.upper() in _metrics.FUNCS
    spec = np.asarray(spec)
    cspec, target = _spec_formatter(spec, **stft_kwargs)
    a = args_helper(target.shape[-2], target.dtype, **stft_kwargs)
    rdt = target.dtype.type
    rho = rdt(rho)

    st = {"X": cspec, "Y": cspec.copy(), "U": np.zeros_like(cspec)}
    st["x"], env = istft(cspec, a)                                     # :453
    st["x"] = st["x"].astype(target.dtype)

    def closure():
        rec = stft(st["x"], a)                                         # :464
        out = np.abs(rec)
        z = (rho * st["Y"] + rec) / (rdt(1) + rho)                     # :467
        u = st["U"] + st["X"] - z                                      # :468
        x_ = z - u                                                     # :471
        norm = np.abs(x_) + rdt(1e-16)                                 # :472
        x_ = x_ * target / norm                                        # :473
        y = x_ + u                                                     # :475
        sig, _ = istft(y, a, envelope=env)                             # :477
        st.update(X=x_, Y=y, U=u, x=sig.astype(target.dtype))
        return out

    done = training_loop(closure, target, max_iter, tol, eva_iter, metric, trace)
    x = _finish(st["x"], spec)
    if return_state:
        return x, {"iters": done, "X": st["X"], "U": st["U"], "envelope": env}
    return x


# --------------------------------------------------------------------------- #
# RTISI-LA                                                                     #
# --------------------------------------------------------------------------- #
def rtisi_asym_windows(a: StftArgs, dtype):
    """Asymmetric analysis windows, methods.py:318-336.  Returns (c, a1, a2)."""
    w = a.window.astype(dtype, copy=False)
    n, hop = a.win_length, a.hop_length
    coeff = dtype.type(hop) / (w @ w)                                   # :318
    keep = (n - 1) // hop                                               # :322
    wf = w[::-1]
    a1 = np.zeros(n, dtype=dtype)
    for i in range(keep):                                               # :326-329
        s = (i + 1) * hop
        a1[s:] += wf[:n - s]
    a1 *= coeff
    a2 = np.zeros(n, dtype=dtype)
    for i in range(keep + 1):                                           # :332-335
        s = i * hop
        a2[s:] += wf[:n - s]
    a2 *= coeff
    return coeff, a1, a2


def rtisi_la(spec, look_ahead=-1, asymmetric_window=False, max_iter=25, alpha=0.99,
             step_dump=None, **stft_kwargs):
    """methods.py:273-412 (RTISI with look-ahead).

    `step_dump`, if a list, receives for the very first inner step a dict of
    the state before/after (used for single-step parity: SURVEY 8c / G5).
    """
    assert max_iter > 0
    assert alpha >= 0
    spec = np.asarray(spec)
    assert not np.iscomplexobj(spec)
    assert 4 > spec.ndim > 1
    target = spec[None] if spec.ndim == 2 else spec
    dtype = target.dtype
    rdt = dtype.type
    a = args_helper(target.shape[-2], dtype, **stft_kwargs)
    n, hop = a.n_fft, a.hop_length
    w = a.window.astype(dtype, copy=False)
    coeff, a1, a2 = rtisi_asym_windows(a, dtype)
    keep = (a.win_length - 1) // hop
    if look_ahead < 0:
        look_ahead = keep
    la = look_ahead
    steps = target.shape[2]
    b = target.shape[0]
    mpad = np.pad(target, ((0, 0), (0, 0), (la, la)))                   # :339

    import scipy.fft as sfft
    norm = "ortho" if a.normalized else "backward"
    if a.onesided:
        def ifr(s):      # (B, F, K) -> (B, K, N) real frames
            return sfft.irfft(np.swapaxes(s, 1, 2), n=n, axis=-1, norm=norm)

        def ffr(fr):     # (B, K, N) -> (B, F, K)
            return np.swapaxes(sfft.rfft(fr, n=n, axis=-1, norm=norm), 1, 2)
    else:
        def ifr(s):
            return sfft.ifft(np.swapaxes(s, 1, 2), n=n, axis=-1, norm=norm).real

        def ffr(fr):
            return np.swapaxes(sfft.fft(fr, n=n, axis=-1, norm=norm), 1, 2)

    cdt = np.result_type(dtype, np.complex64)
    # frames are kept as (B, K, N) here (the reference keeps (B, N, K))
    kept = np.zeros((b, keep, n), dtype=dtype)                          # :354
    first = ifr((mpad[:, :, la:la + 1]).astype(cdt)).astype(dtype)      # :353-358
    update = np.concatenate([np.zeros((b, la, n), dtype=dtype), first], axis=1)

    lr = rdt(alpha / (1 + alpha))
    wsyn = w * coeff
    pre = None
    outs = []
    for i in range(steps + la):                                         # :363
        for j in range(max_iter):                                       # :364
            allf = np.concatenate([kept, update], axis=1)               # (B, K+LA+1, N)
            x = _stft.overlap_add(allf * wsyn, hop, 0)                  # :365-368
            x = x[:, keep * hop:]                                       # :370
            sb, sl = x.strides
            fr = np.lib.stride_tricks.as_strided(
                x, shape=(b, la + 1, n), strides=(sb, sl * hop, sl), writeable=False)
            if asymmetric_window:                                       # :371-383
                head = fr[:, :-1] * w
                tail = fr[:, -1:] * (a2 if j else a1)
                new = ffr(np.concatenate([head, tail], axis=1))
            else:
                new = ffr(fr * w)                                       # :385
            new = new.astype(cdt)
            raw = new
            if j:                                                       # :387-392
                new = new - lr * pre
            elif i:
                new = np.concatenate(
                    [new[:, :, :-1] - lr * pre[:, :, 1:], new[:, :, -1:]], axis=2)
            pre = new
            nrm = np.abs(new) + rdt(1e-16)                              # :394
            new = new * mpad[:, :, i:i + la + 1] / nrm                  # :395-396
            new_update = ifr(new).astype(dtype)                         # :398
            if step_dump is not None and i == 0 and j == 0:
                step_dump.append({"x": x.copy(), "stft": raw.copy(),
                                  "projected": new.copy(), "update": new_update.copy()})
            update = new_update
        outs.append(update[:, 0])                                       # :401
        kept = np.concatenate([kept[:, 1:], update[:, :1]], axis=1)     # :402-403
        update = np.concatenate([update[:, 1:], np.zeros((b, 1, n), dtype=dtype)], axis=1)

    allx = np.stack(outs[la:], axis=1)                                  # :406  (B, T, N)
    y = _stft.overlap_add(allx * w, hop, a.win_length // 2 if a.center else 0)
    env = _stft.ola_envelope(steps, a, dtype=dtype)
    with np.errstate(divide="ignore", invalid="ignore"):
        x = (y / env).astype(dtype)                                     # :407-408
    return _finish(x, spec)
