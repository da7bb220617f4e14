"""Seeded random sweep over (n_fft, hop, frames, batch, window, centre, pad mode, normalized, sidedness, dtype,
method): the device result after a few iterations against the oracle.  A safety net over the many kernel variants
(fused with hop = n_fft/2, /4, /8 at n_fft 512 ... 4096, frame kernel, generic radix schedules); every case prints
the path it took.  Needs an MI355X: `-m gpu`."""
import os

import numpy as np
import pytest
import torch

import oracle
from _util import finite_close, hann

pytestmark = pytest.mark.gpu

import spectrogram_inversion_amd as si                          # noqa: E402
from spectrogram_inversion_amd.plan import args_helper, get_plan  # noqa: E402

DEV = torch.device("cuda", 0)
N_FFTS = [64, 96, 128, 250, 256, 400, 512, 512, 1000, 1024, 1024, 2048, 2048, 4096]


def draw(seed, wave_level=False):
    rng = np.random.default_rng(1000 + seed)
    n_fft = int(rng.choice([512, 1024, 2048, 4096] if wave_level else N_FFTS))
    kind = rng.integers(0, 5)
    hop = [n_fft // 2, n_fft // 4, n_fft // 8, n_fft // 3, int(rng.integers(1, n_fft))][kind]
    hop = max(1, hop)
    onesided = bool(rng.random() < 0.8) or n_fft % 2 == 1
    if n_fft % 2:
        onesided = False
    dtype = np.float64 if rng.random() < 0.25 else np.float32
    if wave_level:
        onesided, dtype = True, np.float32
    center = bool(rng.random() < 0.8)
    pad_mode = str(rng.choice(["reflect", "reflect", "constant", "replicate", "circular"]))
    normalized = bool(rng.random() < 0.3)
    wkind = rng.integers(0, 3)
    if wkind == 0:
        win_length, window = n_fft, hann(n_fft, dtype)
    elif wkind == 1:
        win_length, window = n_fft, None                           # rectangular (methods.py:76-77)
    else:
        win_length = int(rng.integers(max(2, n_fft // 2), n_fft + 1))
        window = hann(win_length, dtype) + dtype(0.05)            # shorter than n_fft: centre-padded (:80-83)
    frames = int(rng.integers(2, 48))
    batch = int(rng.integers(1, 4))
    method = "gla" if rng.random() < 0.6 else "admm"
    coef = float(rng.choice([0.0, 0.3, 0.99])) if method == "gla" else float(rng.choice([0.2, 1.0]))
    kw = dict(win_length=win_length, window=window, hop_length=hop, center=center, pad_mode=pad_mode,
              normalized=normalized, onesided=onesided)
    F = n_fft // 2 + 1 if onesided else n_fft
    mag = (rng.random((batch, F, frames)) + 0.02).astype(dtype)
    return n_fft, kw, mag, method, coef


# SPECINV_EXTRA_SEEDS="a:b" adds seeds a..b-1 (even: the full draw, odd: wave-level shapes) for an occasional wider sweep
# (1000:1800 -> 905 pass; 1291, a rectangular win_length < n_fft window with hop = n_fft/2, is ill-conditioned at the
# signal's end: every kernel path AND the float32 oracle leave the float64 oracle by 2e-2 there after 3 iterations,
# tools/dbg_seed.py 1291; round 6, 1000:1800 -> 1732 pass: 1291 again and three RTISI draws with hop > n_fft / 2 - 1012, 1141, 1168 -
# where the float32 oracle itself is 2e-3 from the float64 one and the kernel 2 - 2.5 x that, over the 2 x bar)
_extra = os.environ.get("SPECINV_EXTRA_SEEDS", "")
EXTRA = list(range(*map(int, _extra.split(":")))) if _extra else []


@pytest.mark.parametrize("seed", list(range(60)) + list(range(100, 160)) + EXTRA)
def test_random_configuration(seed, request):
    # seeds 100..159 (and the odd extra ones): float32 one-sided pow-2 sizes only
    n_fft, kw, mag, method, coef = draw(seed, wave_level=(100 <= seed < 160) or (seed >= 1000 and seed % 2 == 1))
    if seed % 3:
        request.getfixturevalue("chunked_kernel")    # these shapes are all small: keep the fused kernel in the sweep
    pad = n_fft // 2 if kw["center"] else 0
    length = (mag.shape[2] - 1) * kw["hop_length"] + n_fft - 2 * pad
    if length < 1 or (kw["center"] and kw["pad_mode"] in ("reflect", "circular") and pad >= length):
        pytest.skip("torch.stft itself refuses this padding")
    okw = dict(kw)
    tkw = dict(kw)
    if kw["window"] is not None:
        tkw["window"] = torch.from_numpy(kw["window"])
    iters = 3
    init = oracle.phase_init(mag, **okw)
    if method == "gla":
        ref = oracle.griffin_lim(init, max_iter=iters, alpha=coef, tol=0, **okw)
        y = si.griffin_lim(torch.from_numpy(init).to(DEV), max_iter=iters, alpha=coef, tol=0, verbose=False, **tkw)
    else:
        ref = oracle.admm(init, max_iter=iters, rho=coef, tol=0, **okw)
        y = si.ADMM(torch.from_numpy(init).to(DEV), max_iter=iters, rho=coef, tol=0, verbose=False, **tkw)
    y = y.cpu().numpy()
    a = args_helper(torch.from_numpy(mag), **tkw)
    path = get_plan(a, mag.shape[0], mag.shape[2], torch.float32 if mag.dtype == np.float32 else torch.float64, DEV).path
    print(f"seed {seed}: n_fft {n_fft} hop {kw['hop_length']} frames {mag.shape[2]} batch {mag.shape[0]} {mag.dtype.name} "
          f"{method} center={kw['center']} {kw['pad_mode']} norm={kw['normalized']} onesided={kw['onesided']} -> {path}")
    assert y.shape == np.asarray(ref).reshape(y.shape).shape
    ref = np.asarray(ref).reshape(y.shape)
    tol = 2e-4 if mag.dtype == np.float32 else 1e-9
    # compare on the samples where the envelope is not tiny (rectangular / short windows without centring divide
    # by almost nothing at the edges and amplify rounding there): weight by the envelope like the reference's own
    # tests effectively do
    env = oracle.stftlib.ola_envelope(mag.shape[2], oracle.args_helper(mag.shape[1], mag.dtype, **okw), mag.dtype)
    good = env > 1e-3 * env.max()
    assert np.array_equal(np.isfinite(y), np.isfinite(ref))
    assert finite_close(y[..., good], ref[..., good], tol), (seed, path)


# (extra seeds 1000:1400 -> 396 of 400 inside the bound, the rest within 2.5x of it: the recursion amplifies rounding noise)
@pytest.mark.parametrize("seed", list(range(30)) + EXTRA)
def test_random_rtisi_configuration(seed):
    """RTISI_LA (asymmetric window: the numerically stable variant) against the oracle, and the streaming form against
    the whole-signal form, on random shapes."""
    rng = np.random.default_rng(5000 + seed)
    n_fft = int(rng.choice([64, 128, 200, 256, 512, 512, 1024, 1024, 2048]))
    hop = int(rng.choice([n_fft // 4, n_fft // 4, n_fft // 2, n_fft // 8, n_fft // 3, int(rng.integers(n_fft // 8, n_fft))]))
    dtype = np.float64 if rng.random() < 0.3 else np.float32
    frames = int(rng.integers(3, 14))
    batch = int(rng.integers(1, 3))
    keep = (n_fft - 1) // hop
    la = int(rng.choice([-1, 0, 1, 2, 3, min(keep + 1, 7)]))
    iters = int(rng.integers(1, 4))
    alpha = float(rng.choice([0.0, 0.5, 0.99]))
    mag = (rng.random((batch, n_fft // 2 + 1, frames)) + 0.02).astype(dtype)
    w = hann(n_fft, dtype)
    ref = oracle.rtisi_la(mag, look_ahead=la, asymmetric_window=True, max_iter=iters, alpha=alpha, hop_length=hop, window=w)
    tw = torch.from_numpy(w)
    y = si.RTISI_LA(torch.from_numpy(mag).to(DEV), look_ahead=la, asymmetric_window=True, max_iter=iters, alpha=alpha,
                    verbose=False, hop_length=hop, window=tw)
    ref = np.asarray(ref).reshape(tuple(y.shape))
    tol = 5e-4 if dtype == np.float32 else 1e-9
    yn = y.cpu().numpy()
    assert np.array_equal(np.isfinite(yn), np.isfinite(ref))
    fin = np.isfinite(ref)
    if dtype == np.float32:
        # the recursion amplifies rounding differently from shape to shape (little overlap, many iterations): the
        # yardstick is the oracle's own float32-vs-float64 distance on the same problem
        w64 = hann(n_fft, np.float64)
        ref64 = np.asarray(oracle.rtisi_la(mag.astype(np.float64), look_ahead=la, asymmetric_window=True, max_iter=iters,
                                           alpha=alpha, hop_length=hop, window=w64)).reshape(yn.shape)
        scale = np.abs(ref64[fin]).max()
        noise = np.abs(ref[fin] - ref64[fin]).max() / scale
        err = np.abs(yn[fin] - ref64[fin]).max() / scale
        assert err <= max(tol, 2 * noise), (seed, n_fft, hop, la, iters, alpha, err, noise)
    else:
        assert finite_close(yn, ref, tol), (seed, n_fft, hop, la, iters, alpha, dtype)
    good = fin.all(axis=0) if fin.ndim == 2 else fin
    # streaming: same samples (bit-identical when both run the generic kernel, to rounding when the whole-signal
    # call took the wave-level kernel)
    s = si.RTISIStream(n_fft // 2 + 1, batch=batch, look_ahead=la, asymmetric_window=True, max_iter=iters, alpha=alpha,
                       max_push=4, dtype=torch.from_numpy(mag).dtype, device=DEV, hop_length=hop, window=tw)
    pieces, t = [], 0
    while t < frames:
        k = int(rng.integers(1, 5))
        pieces.append(s.push(torch.from_numpy(mag[:, :, t:t + k]).to(DEV)))
        t += k
    pieces.append(s.flush())
    z = torch.cat(pieces, 1)
    assert z.shape == y.shape
    zn = z.cpu().numpy()
    assert np.array_equal(np.isfinite(zn), np.isfinite(yn))
    assert finite_close(zn[..., good], yn[..., good], max(tol, 2 * noise) if dtype == np.float32 else tol)
