"""The wave-level coverage kernel `k_wave_iter` (csrc/kernels_wave.h, round 6): float64 at power-of-two n_fft 128 ... 2048 and float32
at 128 / 256 - both dtypes and two of the three frame sizes of the reference's own sweep (test/test_griffin.py:9-32,
test/consts.py:1-3) - through the C ABI against the oracle, and against the workgroup-level kernels it replaced on the same
buffers.  Needs an MI355X: `-m gpu`."""
import numpy as np
import pytest
import torch

import oracle
from _util import hann, rel_l2

pytestmark = pytest.mark.gpu

import spectrogram_inversion_amd as si                                    # noqa: E402
from spectrogram_inversion_amd.plan import Plan, args_helper               # noqa: E402

DEV = torch.device("cuda", 0)


def N(t):
    return t.detach().cpu().numpy()


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _plan(init, frames, dtype, wave, monkeypatch, **kw):
    monkeypatch.setenv("SPECINV_GENERIC_WAVE", "1" if wave else "0")
    tkw = dict(kw)
    tkw["window"] = torch.from_numpy(kw["window"])
    p = Plan(args_helper(T(init), **tkw), init.shape[0], frames, torch.float32 if dtype == np.float32 else torch.float64, DEV)
    p.force_generic(True)
    return p


SWEEP = [  # n_fft, hop, frames, batch, extra stft kwargs
    (128, 32, 23, 5, {}),                                           # five items of 23 frames: groups of eight frames straddle items
    (128, 100, 9, 3, dict(win_length=100)),                         # the reference's win_length < n_fft, a hop that divides nothing
    (256, 64, 19, 3, dict(pad_mode="constant")),
    (256, 64, 12, 2, dict(center=False)),                           # the envelope vanishes at the edges: the reference's NaN pattern
    (256, 77, 11, 2, dict(pad_mode="circular", normalized=True)),
    (512, 128, 10, 3, dict(pad_mode="replicate")),
    (512, 300, 7, 2, dict(win_length=300, normalized=True)),
    (1024, 256, 9, 2, {}),
    (2048, 512, 7, 2, dict(pad_mode="reflect")),
    (2048, 333, 6, 1, dict(center=False)),
    (128, 32, 17, 3, dict(onesided=False)),
    (512, 100, 11, 2, dict(onesided=False, win_length=300)),        # the reference's own two-sided parametrisation
    (1024, 256, 9, 2, dict(onesided=False, normalized=True)),
    (2048, 512, 6, 2, dict(onesided=False, pad_mode="constant")),
    (400, 160, 13, 3, {}),                                          # the sizes that are not powers of two: radix 10 / 5 / 2 passes
    (400, 100, 9, 2, dict(onesided=False, pad_mode="replicate")),
    (800, 200, 8, 2, dict(win_length=600, normalized=True)),
    (800, 300, 7, 2, dict(center=False)),
    (1000, 250, 7, 2, dict(pad_mode="circular")),
    (1000, 333, 6, 1, dict(onesided=False, win_length=800)),
    (4096, 1024, 9, 2, {}),                                         # a frame on a team of two (float32) / four (float64) waves
    (4096, 1000, 8, 1, dict(win_length=3000, onesided=False)),
    (8192, 2048, 9, 1, dict(pad_mode="constant")),                  # ... of four / eight waves
    (8192, 3000, 7, 1, dict(onesided=False, normalized=True)),
    (16384, 4096, 9, 1, {}),                                        # float32 only (see the test): eight waves
]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n_fft,hop,frames,batch,extra", SWEEP)
def test_wave_kernel_kwarg_sweep_vs_oracle(monkeypatch, dtype, n_fft, hop, frames, batch, extra):
    """Griffin-Lim (4 iterations, alpha 0.5) and ADMM (3 iterations, rho 1.0) from a complex start on `k_wave_iter`, every stft
    kwarg of the reference's sweep: waveform against the oracle in the plan's dtype and in float64, the evaluation's sums and the
    stored state against the workgroup-level kernel (`k_iter_pair`: the same update_core per bin on another transform)."""
    if n_fft == 16384 and dtype == np.float64:
        pytest.skip("float64 frames of 16384 samples are kernels_big.h's (rows through device memory)")
    rng = np.random.default_rng(n_fft + hop + frames)
    wl = extra.get("win_length", n_fft)
    w = hann(wl, dtype)
    kw = dict(hop_length=hop, **extra)
    mag = (rng.random((batch, n_fft // 2 + 1 if extra.get("onesided", True) else n_fft, frames)) + 0.05).astype(dtype)
    cd = np.complex64 if dtype == np.float32 else np.complex128
    init = (mag * np.exp(1j * rng.uniform(-np.pi, np.pi, mag.shape))).astype(cd)
    tol = 2e-5 if dtype == np.float32 else 1e-10
    for method, arg, iters in (("griffin_lim", 0.5, 4), ("admm", 1.0, 3)):
        okw = dict(max_iter=iters, alpha=arg) if method == "griffin_lim" else dict(max_iter=iters, rho=arg)
        with np.errstate(all="ignore"):
            ref = getattr(oracle, method)(init, tol=0, window=w, **okw, **kw)
            ref64 = getattr(oracle, method)(init.astype(np.complex128), tol=0, window=hann(wl, np.float64), **okw, **kw)
        out = {}
        for arm in ("wave", "workgroup"):
            p = _plan(init, frames, dtype, arm == "wave", monkeypatch, window=w, **kw)
            (p.gla_init if method == "griffin_lim" else p.admm_init)(T(init), None, arg)
            assert p.launch_geometry["kernel"] == ("k_wave_iter" if arm == "wave" else "k_iter_pair"), p.launch_geometry
            p.iterate(iters - 1)
            sums = p.iterate(1, eval_last=True)
            out[arm] = (N(p.wave()), N(p.state_spec(0)), np.array(sums[:2]),
                        N(p.state_spec(1)) if method == "admm" else None)
        y, fin = out["wave"][0], np.isfinite(ref64)
        assert np.array_equal(np.isfinite(y), fin)                              # the reference's 0 / 0 where the envelope vanishes
        e, e0 = rel_l2(y[fin], ref64[fin]), rel_l2(ref[fin], ref64[fin])
        assert e < max(3 * e0, 5 * tol), (method, e, e0)
        yw = out["workgroup"][0]
        assert np.array_equal(np.isfinite(yw), fin) and rel_l2(y[fin], yw[fin]) < max(3 * e0, 5 * tol)
        for a, b in ((out["wave"][1], out["workgroup"][1]), (out["wave"][3], out["workgroup"][3])):
            if a is None:
                continue
            f2 = np.isfinite(b)
            assert np.array_equal(np.isfinite(a), f2) and rel_l2(a[f2], b[f2]) < max(10 * e0, 5 * tol)
        if np.isfinite(out["workgroup"][2]).all():
            np.testing.assert_allclose(out["wave"][2], out["workgroup"][2], rtol=1e-5 if dtype == np.float32 else 1e-12)


@pytest.mark.parametrize("dtype,n_fft,hop,frames,batch", [(np.float32, 128, 32, 600, 70), (np.float32, 256, 64, 300, 160),
                                                          (np.float64, 1024, 256, 200, 24), (np.float64, 2048, 512, 150, 20)])
def test_wave_kernel_walks_more_frames_than_the_chip_holds(monkeypatch, dtype, n_fft, hop, frames, batch):
    """More groups of frames than resident waves: every wave of the launch walks several (the grid is sized to the chip, not to the
    problem).  Ten iterations from the magnitudes (phase_init), the whole batch against the workgroup-level kernel - waveforms and
    the evaluation every fifth iteration - and three items against the oracle."""
    rng = np.random.default_rng(n_fft)
    mag = (rng.random((batch, n_fft // 2 + 1, frames)) + 0.05).astype(dtype)
    w = hann(n_fft, dtype)
    out = {}
    for arm in ("wave", "workgroup"):
        p = _plan(mag, frames, dtype, arm == "wave", monkeypatch, window=w, hop_length=hop)
        if arm == "wave":
            geo = p.launch_geometry
            assert geo["kernel"] == "k_wave_iter" and geo["waves"] * (1 if n_fft >= 1024 else (4 if n_fft == 512 else 8)) < batch * frames, geo
        c0 = p.phase_init(T(mag))
        p.gla_init(c0, None, 0.3)
        done, evals = p.run(10, 5, 0.0, "sc")
        out[arm] = (N(p.wave()), np.array([[m, l] for _, m, l in evals]), N(c0))
    tol = 5e-5 if dtype == np.float32 else 1e-10
    assert rel_l2(out["wave"][0], out["workgroup"][0]) < tol
    np.testing.assert_allclose(out["wave"][1], out["workgroup"][1], rtol=1e-4 if dtype == np.float32 else 1e-10)
    items = [0, batch // 2, batch - 1]
    ref = oracle.griffin_lim(out["wave"][2][items], max_iter=10, alpha=0.3, tol=0, hop_length=hop, window=w)
    for k, b in enumerate(items):
        assert rel_l2(out["wave"][0][b], ref[k]) < (1e-4 if dtype == np.float32 else 1e-9), (b, rel_l2(out["wave"][0][b], ref[k]))


def test_wave_kernel_is_what_float64_and_small_frames_run(monkeypatch):
    """The drop-in functions on the reference's own test sizes (test/consts.py:1-3: n_fft 128 / 256 / 512, both dtypes) land on the
    wave-level coverage kernel where the plan says it wins - float64 at 128 ... 2048, float32 at 128 / 256, one- and two-sided - and
    on the kernels that served them before everywhere else (float32 512: the packed frame kernels; other sizes)."""
    monkeypatch.delenv("SPECINV_GENERIC_WAVE", raising=False)
    for dtype, n_fft, onesided, want in ((torch.float64, 128, True, "k_wave_iter"), (torch.float64, 512, True, "k_wave_iter"),
                                         (torch.float64, 2048, True, "k_wave_iter"), (torch.float32, 128, True, "k_wave_iter"),
                                         (torch.float32, 256, True, "k_wave_iter"), (torch.float32, 512, True, "k_semi"),
                                         (torch.float64, 512, False, "k_wave_iter"), (torch.float64, 1024, False, "k_wave_iter"),
                                         (torch.float32, 256, False, "k_wave_iter"), (torch.float64, 4096, True, "k_wave_iter"), (torch.float64, 8192, True, "k_wave_iter"), (torch.float32, 8192, True, "k_wave_iter"),
                                         (torch.float32, 16384, True, "k_wave_iter"), (torch.float64, 16384, True, "k_iter_pair"),
                                         (torch.float64, 1000, True, "k_wave_iter"), (torch.float32, 400, True, "k_wave_iter"),
                                         (torch.float32, 800, False, "k_wave_iter"), (torch.float64, 1200, True, "k_iter_pair")):
        F = n_fft // 2 + 1 if onesided else n_fft
        mag = torch.rand((2, F, 12), dtype=dtype, device=DEV) + 0.05
        p = Plan(args_helper(mag, hop_length=n_fft // 4, onesided=onesided, window=torch.hann_window(n_fft, dtype=dtype)), 2, 12, dtype, DEV)
        p.gla_init(None, mag, 0.5)
        assert p.launch_geometry["kernel"] == want, (dtype, n_fft, onesided, p.launch_geometry)
    # ... and end to end through the drop-in function, float64 at the reference's largest test size
    rng = np.random.default_rng(1)
    mag = (rng.random((2, 257, 20)) + 0.05)
    y = N(si.griffin_lim(T(mag), max_iter=8, alpha=0.5, tol=0, verbose=False, hop_length=128, window=torch.from_numpy(hann(512, np.float64))))
    ref = oracle.griffin_lim(oracle.phase_init(mag, hop_length=128, window=hann(512, np.float64)), max_iter=8, alpha=0.5, tol=0,
                             hop_length=128, window=hann(512, np.float64))
    assert rel_l2(y, ref) < 1e-9


@pytest.mark.parametrize("dtype,n_fft,ov,frames,batch,extra", [
    (np.float32, 128, 4, 301, 7, {}), (np.float32, 256, 4, 150, 5, dict(pad_mode="constant")), (np.float32, 256, 2, 77, 3, {}),
    (np.float32, 128, 2, 200, 3, dict(center=False)), (np.float64, 256, 4, 90, 4, {}), (np.float64, 512, 4, 64, 3, dict(normalized=True)),
    (np.float64, 1024, 4, 70, 2, {}), (np.float64, 1024, 8, 100, 2, dict(pad_mode="circular")), (np.float64, 512, 2, 40, 2, dict(center=False)),
    (np.float64, 4096, 4, 40, 2, {}), (np.float32, 4096, 8, 60, 1, dict(pad_mode="replicate")),
    (np.float64, 8192, 4, 30, 1, {}), (np.float32, 8192, 2, 29, 2, dict(center=False)), (np.float32, 16384, 4, 28, 1, {}),
])
def test_register_overlap_add_equals_frames_plus_ola(monkeypatch, dtype, n_fft, ov, frames, batch, extra):
    """hop = n_fft / 2, / 4, / 8: `k_wave_iter` walks chunks of consecutive frames with the overlap-add in registers (partial sums
    carried from frame to frame, a finished hop-block divided by the envelope and stored per frame; methods.py:127-132) instead of
    writing frames for `k_ola`.  The sums are taken in ascending frame order like k_ola's, on rounded windowed samples like the
    frames buffer's: after one iteration the samples outside the chunk boundaries are bit-identical to the frames + k_ola form but
    for the few where the two instantiations' transforms contract a multiply-add differently (3 % measured, one ulp), and the
    boundary blocks (k_wave_seams: left partial + right partial, another association) agree to rounding; after five iterations
    of Griffin-Lim and ADMM both agree with the oracle.
    SPECINV_WAVE_CHUNK pins short chunks so that every shape has several boundaries per item."""
    hop = n_fft // ov
    rng = np.random.default_rng(n_fft + ov)
    w = hann(n_fft, dtype)
    mag = (rng.random((batch, n_fft // 2 + 1, frames)) + 0.05).astype(dtype)
    cd = np.complex64 if dtype == np.float32 else np.complex128
    init = (mag * np.exp(1j * rng.uniform(-np.pi, np.pi, mag.shape))).astype(cd)
    kw = dict(hop_length=hop, **extra)
    monkeypatch.setenv("SPECINV_WAVE_CHUNK", str(max(2 * ov, 9)))
    for method, arg in (("griffin_lim", 0.5), ("admm", 1.0)):
        res = {}
        for arm in ("registers", "frames"):
            monkeypatch.setenv("SPECINV_WAVE_OLA", "1" if arm == "registers" else "0")
            p = _plan(init, frames, dtype, True, monkeypatch, window=w, **kw)
            (p.gla_init if method == "griffin_lim" else p.admm_init)(T(init), None, arg)
            geo = p.launch_geometry
            assert geo["kernel"] == "k_wave_iter" and geo["overlap_add"] == arm and (geo["chunks"] < frames) == (arm == "registers"), (arm, geo)
            p.iterate(1)
            y1 = N(p.wave())
            p.iterate(3)
            sums = p.iterate(1, eval_last=True)
            res[arm] = (y1, N(p.wave()), np.array(sums[:2]), geo["chunks"])
        a, b = res["registers"], res["frames"]
        fin = np.isfinite(b[0])
        assert np.array_equal(np.isfinite(a[0]), fin)
        same = (a[0] == b[0]) | ~fin
        nch = a[3]
        assert nch >= 3
        seam_share = (nch + 1) * (ov - 1) * hop / a[0].shape[1]
        assert 1.0 - same.mean() <= 0.5 * seam_share + 0.1, (1.0 - same.mean(), seam_share)   # mostly at the chunk boundaries
        eps = np.finfo(dtype).eps
        assert np.abs(a[0][fin] - b[0][fin]).max() <= 8 * eps * np.abs(b[0][fin]).max()
        with np.errstate(all="ignore"):
            ref64 = getattr(oracle, method)(init.astype(np.complex128), tol=0, window=hann(n_fft, np.float64), max_iter=5,
                                            **({"alpha": arg} if method == "griffin_lim" else {"rho": arg}), **kw)
            ref = getattr(oracle, method)(init, tol=0, window=w, max_iter=5, **({"alpha": arg} if method == "griffin_lim" else {"rho": arg}), **kw)
        f5 = np.isfinite(ref64)
        e0 = rel_l2(ref[f5], ref64[f5])
        tol = 2e-5 if dtype == np.float32 else 1e-10
        assert np.array_equal(np.isfinite(a[1]), f5) and rel_l2(a[1][f5], ref64[f5]) < max(3 * e0, 5 * tol)
        if np.isfinite(b[2]).all():
            np.testing.assert_allclose(a[2], b[2], rtol=1e-5 if dtype == np.float32 else 1e-12)


@pytest.mark.parametrize("dtype,n_fft,hop,frames,batch,extra", [
    (np.float32, 128, 50, 90, 5, dict(win_length=100)), (np.float32, 256, 77, 120, 3, dict(pad_mode="circular", normalized=True)),
    (np.float64, 512, 100, 130, 3, dict(onesided=False, win_length=300)), (np.float32, 512, 300, 60, 4, dict(win_length=400)),
    (np.float64, 2048, 333, 70, 2, dict(center=False)), (np.float32, 1024, 256, 80, 3, dict(onesided=False)),
    (np.float64, 256, 255, 50, 2, {}), (np.float32, 128, 5, 700, 2, dict(pad_mode="constant")),
    (np.float32, 2048, 512, 48, 3, dict(onesided=False, pad_mode="replicate")), (np.float64, 1024, 200, 75, 2, dict(win_length=800)),
    (np.float64, 128, 32, 100, 3, dict(onesided=False)),
    (np.float32, 400, 160, 100, 5, {}), (np.float64, 400, 100, 70, 3, dict(pad_mode="constant")), (np.float32, 800, 200, 60, 3, dict(onesided=False)),
    (np.float64, 800, 333, 50, 2, dict(win_length=700)), (np.float32, 1000, 250, 64, 3, dict(normalized=True)),
    (np.float64, 1000, 999, 30, 2, dict(onesided=False)), (np.float32, 1000, 77, 140, 2, dict(center=False)),
    (np.float32, 4096, 1000, 40, 2, dict(win_length=3000)),
])
def test_ring_overlap_add_equals_frames_plus_ola(monkeypatch, dtype, n_fft, hop, frames, batch, extra):
    """Every other hop below n_fft (odd ones, hops above n_fft / 2, one sample short of n_fft) and two-sided spectrograms: the same
    chunk walk with the overlap-add in an LDS ring of n_fft samples per lane group (kernels_wave.h, OV == 1) - a frame's samples
    added at (t hop + s) mod n_fft, its first hop samples complete, divided by the envelope and stored (methods.py:127-132); the
    n_fft - hop samples either side of a chunk boundary go through k_wave_seams.  Against the frames + k_ola form of the same
    kernel after one iteration (same order of summation: equal but at the boundaries and where a multiply-add contracts
    differently), and against the oracle after five iterations of Griffin-Lim and ADMM.
    SPECINV_WAVE_CHUNK pins short chunks so that every shape has several boundaries per item."""
    onesided = extra.get("onesided", True)
    rng = np.random.default_rng(n_fft + hop)
    wl = extra.get("win_length", n_fft)
    w = hann(wl, dtype)
    F = n_fft // 2 + 1 if onesided else n_fft
    mag = (rng.random((batch, F, frames)) + 0.05).astype(dtype)
    cd = np.complex64 if dtype == np.float32 else np.complex128
    init = (mag * np.exp(1j * rng.uniform(-np.pi, np.pi, mag.shape))).astype(cd)
    kw = dict(hop_length=hop, **extra)
    over = -(-n_fft // hop)
    monkeypatch.setenv("SPECINV_WAVE_CHUNK", str(max(2 * over, 9)))
    for method, arg in (("griffin_lim", 0.5), ("admm", 1.0)):
        res = {}
        for arm in ("ring", "frames"):
            monkeypatch.setenv("SPECINV_WAVE_OLA", "1" if arm == "ring" else "0")
            p = _plan(init, frames, dtype, True, monkeypatch, window=w, **kw)
            (p.gla_init if method == "griffin_lim" else p.admm_init)(T(init), None, arg)
            geo = p.launch_geometry
            assert geo["kernel"] == "k_wave_iter" and geo["overlap_add"] == arm and (geo["chunks"] < frames) == (arm == "ring"), (arm, geo)
            p.iterate(1)
            y1 = N(p.wave())
            p.iterate(3)
            sums = p.iterate(1, eval_last=True)
            res[arm] = (y1, N(p.wave()), np.array(sums[:2]), geo["chunks"])
        a, b = res["ring"], res["frames"]
        fin = np.isfinite(b[0])
        assert np.array_equal(np.isfinite(a[0]), fin)
        same = (a[0] == b[0]) | ~fin
        nch = a[3]
        assert nch >= 3
        seam_share = (nch + 1) * (n_fft - hop) / a[0].shape[1]
        assert 1.0 - same.mean() <= 0.5 * seam_share + 0.1, (1.0 - same.mean(), seam_share)   # mostly at the chunk boundaries
        eps = np.finfo(dtype).eps
        assert np.abs(a[0][fin] - b[0][fin]).max() <= 8 * eps * np.abs(b[0][fin]).max()
        with np.errstate(all="ignore"):
            ref64 = getattr(oracle, method)(init.astype(np.complex128), tol=0, window=hann(wl, np.float64), max_iter=5,
                                            **({"alpha": arg} if method == "griffin_lim" else {"rho": arg}), **kw)
            ref = getattr(oracle, method)(init, tol=0, window=w, max_iter=5, **({"alpha": arg} if method == "griffin_lim" else {"rho": arg}), **kw)
        f5 = np.isfinite(ref64)
        e0 = rel_l2(ref[f5], ref64[f5])
        tol = 2e-5 if dtype == np.float32 else 1e-10
        assert np.array_equal(np.isfinite(a[1]), f5) and rel_l2(a[1][f5], ref64[f5]) < max(3 * e0, 5 * tol)
        if np.isfinite(b[2]).all():
            np.testing.assert_allclose(a[2], b[2], rtol=1e-5 if dtype == np.float32 else 1e-12)


@pytest.mark.parametrize("batch,frames", [(24, 400), (64, 1024)])
def test_float64_2048_on_a_two_wave_team_at_full_occupancy(monkeypatch, batch, frames):
    """float64 at n_fft 2048 runs a frame on the 128 lanes of a two-wave workgroup (eight points per lane: room for the register
    overlap-add's partial sums).  What is a wave-private exchange elsewhere is a workgroup barrier there - and a missing one (the
    edge frames' parked samples overwritten by the other wave's first-pass outputs) only showed with every workgroup slot of the
    chip taken: BASELINE C2's shape in float64 (B 64, T 1024) and a ragged one, five iterations, both forms of the overlap-add
    against the workgroup-level kernel - 1e-12 on every item, item edges included - and the evaluation's sums."""
    rng = np.random.default_rng(batch)
    mag = (rng.random((batch, 1025, frames)) + 0.05)
    init = mag * np.exp(1j * rng.uniform(-np.pi, np.pi, mag.shape))
    w = hann(2048, np.float64)
    res = {}
    for arm in ("workgroup", "frames", "registers"):
        monkeypatch.setenv("SPECINV_WAVE_OLA", "0" if arm == "frames" else "1")
        p = _plan(init, frames, np.float64, arm != "workgroup", monkeypatch, window=w, hop_length=512)
        p.gla_init(T(init), None, 0.3)
        geo = p.launch_geometry
        if arm != "workgroup":
            assert geo["kernel"] == "k_wave_iter" and geo["waves_per_workgroup"] == 2 and (geo["chunks"] < frames) == (arm == "registers"), geo
        p.iterate(4)
        sums = p.iterate(1, eval_last=True)
        res[arm] = (N(p.wave()), np.array(sums[:2]))
    ref = res["workgroup"]
    for arm in ("frames", "registers"):
        y = res[arm][0]
        per_item = np.linalg.norm(y - ref[0], axis=1) / np.linalg.norm(ref[0], axis=1)
        assert per_item.max() < 1e-12, (arm, int(per_item.argmax()), per_item.max())
        edge = 16 * 512
        assert np.abs(y[:, :edge] - ref[0][:, :edge]).max() < 1e-12 * np.abs(ref[0]).max()
        assert np.abs(y[:, -edge:] - ref[0][:, -edge:]).max() < 1e-12 * np.abs(ref[0]).max()
        np.testing.assert_allclose(res[arm][1], ref[1], rtol=1e-12)


@pytest.mark.parametrize("dtype,n_fft,frames,batch,waves_per_wg", [
    (np.float64, 4096, 512, 16, 4), (np.float32, 4096, 512, 32, 2), (np.float64, 8192, 128, 16, 8), (np.float32, 8192, 256, 16, 4),
    (np.float32, 16384, 128, 16, 8),
])
def test_large_frames_on_teams_at_full_occupancy(monkeypatch, dtype, n_fft, frames, batch, waves_per_wg):
    """n_fft 4096 / 8192 / 16384: a frame on the lanes of a two- to eight-wave workgroup (profiles/r06_generic.txt's shapes: every
    team slot of the chip taken), hop = n_fft / 4, five iterations with the overlap-add in registers and on frames + k_ola against
    the workgroup-level kernels."""
    rng = np.random.default_rng(n_fft)
    mag = (rng.random((batch, n_fft // 2 + 1, frames)) + 0.05).astype(dtype)
    cd = np.complex64 if dtype == np.float32 else np.complex128
    init = (mag * np.exp(1j * rng.uniform(-np.pi, np.pi, mag.shape))).astype(cd)
    w = hann(n_fft, dtype)
    res = {}
    for arm in ("workgroup", "frames", "registers"):
        monkeypatch.setenv("SPECINV_WAVE_OLA", "0" if arm == "frames" else "1")
        p = _plan(init, frames, dtype, arm != "workgroup", monkeypatch, window=w, hop_length=n_fft // 4)
        p.gla_init(T(init), None, 0.3)
        geo = p.launch_geometry
        if arm != "workgroup":
            assert geo["kernel"] == "k_wave_iter" and geo["waves_per_workgroup"] == waves_per_wg and geo["overlap_add"] == arm, geo
        p.iterate(4)
        sums = p.iterate(1, eval_last=True)
        res[arm] = (N(p.wave()), np.array(sums[:2]))
        del p
    ref = res["workgroup"]
    tol = 2e-5 if dtype == np.float32 else 1e-12
    for arm in ("frames", "registers"):
        y = res[arm][0]
        per_item = np.linalg.norm(y - ref[0], axis=1) / np.linalg.norm(ref[0], axis=1)
        # (float32 on random magnitudes: a bin that passes close to zero amplifies rounding in its item - 2e-5 on one of 32 items of the
        # 4096 case, 1 - 3e-6 on the others; the float64 cases of the same code hold 1e-12 everywhere)
        assert np.median(per_item) < tol / 2 and per_item.max() < (5 * tol if dtype == np.float32 else tol), (arm, int(per_item.argmax()), per_item.max())
        edge = 4 * n_fft
        assert np.abs(y[:, :edge] - ref[0][:, :edge]).max() < 10 * tol * np.abs(ref[0]).max()
        assert np.abs(y[:, -edge:] - ref[0][:, -edge:]).max() < 10 * tol * np.abs(ref[0]).max()
        np.testing.assert_allclose(res[arm][1], ref[1], rtol=1e-5 if dtype == np.float32 else 1e-12)


@pytest.mark.parametrize("dtype,n_fft,hop,frames,batch,extra", [
    (np.float64, 2048, 500, 700, 20, {}),                            # the ring on a two-wave team
    (np.float32, 4096, 1000, 400, 24, dict(onesided=False)),         # ... on a two-wave float32 team, two-sided
    (np.float32, 400, 160, 2048, 64, {}),                            # bench.py's W400 leg: three frames per wave
    (np.float64, 512, 100, 1024, 24, dict(onesided=False, win_length=300)),
    (np.float32, 1000, 250, 1024, 24, {}),
])
def test_ring_at_full_occupancy(monkeypatch, dtype, n_fft, hop, frames, batch, extra):
    """The ring overlap-add with every wave slot of the chip taken (the team's missing barrier of this round only showed there):
    five iterations against the workgroup-level kernels on the same input, item by item and at the item edges."""
    rng = np.random.default_rng(n_fft + hop)
    onesided = extra.get("onesided", True)
    wl = extra.get("win_length", n_fft)
    w = hann(wl, dtype)
    F = n_fft // 2 + 1 if onesided else n_fft
    mag = (rng.random((batch, F, frames)) + 0.05).astype(dtype)
    cd = np.complex64 if dtype == np.float32 else np.complex128
    init = (mag * np.exp(1j * rng.uniform(-np.pi, np.pi, mag.shape))).astype(cd)
    res = {}
    for arm in ("workgroup", "ring"):
        p = _plan(init, frames, dtype, arm == "ring", monkeypatch, window=w, hop_length=hop, **extra)
        p.gla_init(T(init), None, 0.3)
        geo = p.launch_geometry
        if arm == "ring":
            assert geo["kernel"] == "k_wave_iter" and geo["overlap_add"] == "ring" and geo["waves"] >= 1000, geo   # every slot its kernel has
        p.iterate(4)
        sums = p.iterate(1, eval_last=True)
        res[arm] = (N(p.wave()), np.array(sums[:2]))
        del p
    y, ref = res["ring"][0], res["workgroup"][0]
    tol = 2e-5 if dtype == np.float32 else 1e-12
    per_item = np.linalg.norm(y - ref, axis=1) / np.linalg.norm(ref, axis=1)
    assert per_item.max() < tol, (int(per_item.argmax()), per_item.max())
    edge = 4 * n_fft
    assert np.abs(y[:, :edge] - ref[:, :edge]).max() < 10 * tol * np.abs(ref).max()
    assert np.abs(y[:, -edge:] - ref[:, -edge:]).max() < 10 * tol * np.abs(ref).max()
    np.testing.assert_allclose(res["ring"][1], res["workgroup"][1], rtol=1e-5 if dtype == np.float32 else 1e-12)


def test_long_signals_keep_the_frames_form(monkeypatch):
    """`k_wave_iter` indexes frames with 32 bits and addresses a wave's lane groups relative to the first; with the register
    overlap-add the groups walk chunks up to a whole item apart, so a signal of 8 n_frames (n_fft + 2) >= 2^31 elements keeps the
    frames + k_ola form (float64 n_fft 2048, 140 000 frames) - against the workgroup-level kernels on the same input."""
    rng = np.random.default_rng(8)
    frames, n_fft, hop = 140_000, 2048, 512
    mag = (rng.random((1, n_fft // 2 + 1, frames)) + 0.05)
    w = hann(n_fft, np.float64)
    res = {}
    for arm in ("wave", "workgroup"):
        p = _plan(mag, frames, np.float64, arm == "wave", monkeypatch, window=w, hop_length=hop)
        c0 = p.phase_init(T(mag))
        p.gla_init(c0, None, 0.3)
        if arm == "wave":
            geo = p.launch_geometry
            assert geo["kernel"] == "k_wave_iter" and geo["chunks"] == frames, geo       # no chunks: the frames form
        p.iterate(3)
        res[arm] = N(p.wave())
    assert rel_l2(res["wave"], res["workgroup"]) < 1e-12
