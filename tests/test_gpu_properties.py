"""Size-independent properties at BASELINE.json's full frame sizes, and degenerate shapes.  Needs an MI355X."""
import numpy as np
import pytest
import torch

import oracle
from _util import hann, rel_l2, sc_linear

pytestmark = pytest.mark.gpu

import spectrogram_inversion_amd as si                               # noqa: E402
from spectrogram_inversion_amd.plan import Plan, args_helper          # noqa: E402

DEV = torch.device("cuda", 0)


def N(t):
    return t.detach().cpu().numpy()


def plan(n_fft, hop, frames, batch, dtype=torch.float32, generic=False):
    np_dt = np.float32 if dtype == torch.float32 else np.float64
    a = args_helper(torch.empty(1, n_fft // 2 + 1, 1, dtype=dtype), hop_length=hop, window=torch.from_numpy(hann(n_fft, np_dt)))
    p = Plan(a, batch, frames, dtype, DEV)
    if generic:
        p.force_generic(True)
    return p


@pytest.mark.parametrize("n_fft,hop,frames,batch", [(2048, 512, 1024, 16), (1024, 256, 2048, 8)])
def test_stft_istft_round_trip_and_linearity_full_size(n_fft, hop, frames, batch):
    """ISTFT(STFT(x)) = x (Hann, 75 % overlap: the envelope division makes the pair exact) and STFT is linear -
    at the frame counts of BASELINE configs 2 and 4."""
    p = plan(n_fft, hop, frames, batch)
    torch.manual_seed(0)
    x = torch.randn(batch, p.length, device=DEV)
    y = torch.randn(batch, p.length, device=DEV)
    sx, sy = p.stft(x), p.stft(y)
    assert rel_l2(N(p.istft(sx)), N(x)) < 2e-6
    lin = p.stft(2.5 * x - 0.75 * y)
    assert rel_l2(N(lin), N(2.5 * sx - 0.75 * sy)) < 2e-6
    # Parseval-type checksum: sum |S|^2 over the two-sided spectrum equals n_fft * sum over frames of |w x|^2
    g = plan(n_fft, hop, frames, batch, generic=True)
    assert rel_l2(N(g.stft(x)), N(sx)) < 2e-6


def test_consistent_spectrogram_is_a_fixed_point_full_size():
    """A magnitude that IS the STFT of a signal, started from its true phase, must stay put: one Griffin-Lim
    iteration (alpha = 0) returns the same waveform, and the spectral convergence is ~float32 epsilon."""
    n_fft, hop, frames, batch = 2048, 512, 1024, 8
    p = plan(n_fft, hop, frames, batch)
    assert p.fast_path
    torch.manual_seed(1)
    x = torch.randn(batch, p.length, device=DEV)
    spec = p.stft(x)
    p.gla_init(spec, None, 0.0)
    s = p.iterate(3, eval_last=True)
    assert rel_l2(N(p.wave()), N(x)) < 5e-6
    assert np.sqrt(s[0] / s[2]) < 1e-5                           # linear SC


def test_admm_full_frame_size_fast_vs_float64():
    """BASELINE config 4 frame size (n_fft 1024, hop 256, 2048 frames): fused float32 ADMM against the generic
    kernels in float64 after 3 iterations (rho = 1: no rounding amplification)."""
    n_fft, hop, frames, batch = 1024, 256, 2048, 4
    rng = np.random.default_rng(4)
    mag = torch.from_numpy(rng.random((batch, n_fft // 2 + 1, frames), dtype=np.float32)).to(DEV)
    f32, f64 = plan(n_fft, hop, frames, batch), plan(n_fft, hop, frames, batch, torch.float64)
    assert f32.fast_path and not f64.fast_path
    init = f32.phase_init(mag)
    f32.keep_state()
    f32.admm_init(init, None, 1.0)
    f64.admm_init(init.to(torch.complex128), None, 1.0)
    f32.iterate(3)
    f64.iterate(3)
    assert rel_l2(N(f32.wave()), N(f64.wave())) < 5e-6
    assert rel_l2(N(f32.state_spec(1)), N(f64.state_spec(1))) < 5e-5


def test_rtisi_fast_equals_generic_at_config3_frame_size():
    rng = np.random.default_rng(3)
    mag = torch.from_numpy(rng.random((2, 1025, 40), dtype=np.float32)).to(DEV)
    w = torch.from_numpy(hann(2048))
    out = []
    for generic in (False, True):
        a = args_helper(mag, hop_length=512, window=w)
        p = Plan(a, 2, 40, torch.float32, DEV)
        if generic:
            p.force_generic(True)
        out.append(N(p.rtisi(mag, 3, True, 25, 0.99)))
    assert rel_l2(out[0], out[1]) < 2e-3          # 25 its x 43 frames of float32 noise, asymmetric window (stable)


@pytest.mark.parametrize("shape,n_fft,hop", [((1, 3, 4), 4, 1), ((1, 65, 4), 128, 32), ((3, 9, 2), 16, 16), ((1, 257, 4), 512, 128)])
def test_degenerate_shapes(shape, n_fft, hop):
    """The shortest signals torch.stft's reflect padding admits (length just above n_fft/2), hop = n_fft (no
    overlap), n_fft = 4: the generic kernels against the oracle."""
    rng = np.random.default_rng(shape[1])
    mag = rng.random(shape, dtype=np.float32) + 0.1
    w = np.ones(n_fft, dtype=np.float32)
    for fn, ofn, kw in ((si.griffin_lim, oracle.griffin_lim, dict(alpha=0.5)), (si.ADMM, oracle.admm, dict(rho=0.5))):
        ref = ofn(mag, max_iter=3, tol=0, hop_length=hop, window=w, **kw)
        y = N(fn(torch.from_numpy(mag).to(DEV), max_iter=3, tol=0, verbose=False, hop_length=hop,
                 window=torch.from_numpy(w), **kw))
        assert y.shape == ref.shape
        fin = np.isfinite(ref)
        assert np.array_equal(np.isfinite(y), fin)
        if fin.any():
            assert np.abs(y[fin] - ref[fin]).max() <= 1e-4 * max(1.0, np.abs(ref[fin]).max())


def test_large_batch_of_short_items(chunked_kernel):
    """Many independent items (batch 512): sharding unit of the multi-GPU path; every item equals its solo run."""
    rng = np.random.default_rng(9)
    mag = rng.random((512, 513, 8), dtype=np.float32)
    w = torch.from_numpy(hann(1024))
    y = N(si.griffin_lim(torch.from_numpy(mag).to(DEV), max_iter=4, alpha=0.3, tol=0, verbose=False, hop_length=256, window=w))
    for i in (0, 255, 511):
        yi = N(si.griffin_lim(torch.from_numpy(mag[i]).to(DEV), max_iter=4, alpha=0.3, tol=0, verbose=False,
                              hop_length=256, window=w))
        np.testing.assert_array_equal(y[i], yi)          # bitwise: items do not interact


@pytest.mark.parametrize("n_fft,hop,frames", [(1024, 256, 120_000), (512, 128, 200_000), (2048, 256, 40_000),
                                             (1024, 300, 50_000), (2048, 333, 30_000)])   # last two: chunked frame kernel
def test_very_long_signal(n_fft, hop, frames):
    """One item of up to 200 000 frames (30 M samples; 32-bit offsets inside a row, 64-bit across) against the float64
    kernels: the beginning, the middle and the end of the waveform, the whole waveform and the spectral convergence."""
    g = torch.Generator().manual_seed(frames)
    mag = torch.rand((1, n_fft // 2 + 1, frames), generator=g) + 0.01
    w = torch.from_numpy(hann(n_fft))
    p32 = Plan(args_helper(mag, hop_length=hop, window=w), 1, frames, torch.float32, DEV)
    p64 = Plan(args_helper(mag.double(), hop_length=hop, window=w.double()), 1, frames, torch.float64, DEV)
    assert p32.path in ("fused", "frame") and p64.path == "generic"
    assert p32.path_code == (1 if n_fft % hop == 0 else 3)
    c0 = p64.phase_init(mag.double().to(DEV))
    p32.gla_init(c0.to(torch.complex64), None, 0.3)
    p64.gla_init(c0, None, 0.3)
    p32.iterate(4)
    p64.iterate(4)
    s32, s64 = p32.iterate(1, eval_last=True), p64.iterate(1, eval_last=True)
    y32, y64 = p32.wave(), p64.wave()
    assert y32.shape == (1, (frames - 1) * hop)
    n = 50 * hop
    for sl in (slice(0, n), slice(y64.shape[1] // 2, y64.shape[1] // 2 + n), slice(-n, None)):
        assert rel_l2(N(y32[:, sl]), N(y64[:, sl])) < 2e-5
    assert rel_l2(N(y32), N(y64)) < 1e-4                     # the north-star bar; (typically 2e-5)
    assert abs(np.sqrt(s32[0] / s32[2]) - np.sqrt(s64[0] / s64[2])) < 1e-5


def test_wide_batch(chunked_kernel):
    """8 192 short items on the fused kernel (grid and batch-index arithmetic), spot-checked against solo runs."""
    rng = np.random.default_rng(10)
    mag = rng.random((8192, 257, 12), dtype=np.float32)
    w = torch.from_numpy(hann(512))
    y = N(si.griffin_lim(torch.from_numpy(mag).to(DEV), max_iter=3, alpha=0.3, tol=0, verbose=False, hop_length=128, window=w))
    assert np.isfinite(y).all()
    for i in (0, 4095, 8191):
        yi = N(si.griffin_lim(torch.from_numpy(mag[i]).to(DEV), max_iter=3, alpha=0.3, tol=0, verbose=False,
                              hop_length=128, window=w))
        np.testing.assert_array_equal(y[i], yi)
