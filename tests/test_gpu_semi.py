"""The any-hop wave-level iteration (k_semi + k_ola: n_fft 1024 / 2048 with hop != n_fft/4 or centre = False)
against the oracle and the generic kernels, through the C ABI.  Needs an MI355X: `-m gpu`."""
import os

import numpy as np
import pytest
import torch

import oracle
from _util import finite_close, hann, rel_l2, sc_linear

pytestmark = pytest.mark.gpu

import spectrogram_inversion_amd as si                          # noqa: E402
from spectrogram_inversion_amd.plan import Plan, args_helper    # noqa: E402

DEV = torch.device("cuda", 0)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def N(t):
    return t.detach().cpu().numpy()


def make_plan(n_fft, hop, frames, batch, dtype=torch.float32, **kw):
    w = kw.pop("window", None)
    w = torch.from_numpy(hann(n_fft)) if w is None else torch.from_numpy(w)
    probe = torch.empty((1, n_fft // 2 + 1, 1))
    return Plan(args_helper(probe, hop_length=hop, window=w.to(dtype), **kw), batch, frames, dtype, DEV)


# n_fft, hop, frames, batch, extra: hops that do not divide n_fft, odd hops (unaligned frame starts), hop = n_fft/16
# and 3 n_fft/8, too few frames for the fused kernel, no centring
SHAPES = [(1024, 64, 40, 2, {}), (2048, 768, 21, 2, {}), (2048, 333, 17, 1, {}), (1024, 100, 33, 3, {}),
          (2048, 1024, 3, 2, {}), (1024, 256, 5, 2, {}), (2048, 512, 2, 1, dict(pad_mode="constant")),
          (1024, 256, 12, 2, dict(center=False, window=np.ones(1024, dtype=np.float32))),
          (2048, 300, 10, 1, dict(pad_mode="circular")), (512, 64, 50, 2, {}), (512, 100, 31, 2, {}),
          (4096, 1000, 12, 1, {}), (4096, 1024, 5, 2, {}), (512, 128, 20, 2, dict(center=False, window=np.ones(512, dtype=np.float32))), (1024, 192, 16, 2, dict(normalized=True, pad_mode="replicate"))]


@pytest.mark.parametrize("n_fft,hop,frames,batch,extra", SHAPES)
def test_semi_gla_matches_oracle(n_fft, hop, frames, batch, extra):
    rng = np.random.default_rng(n_fft + hop + frames)
    mag = rng.random((batch, n_fft // 2 + 1, frames), dtype=np.float32) + 0.01
    okw = dict(extra)
    w = okw.pop("window", hann(n_fft))
    init = oracle.phase_init(mag, hop_length=hop, window=w, **okw)
    trace = []
    ref, st = oracle.griffin_lim(init, max_iter=6, alpha=0.3, tol=0, eva_iter=3, hop_length=hop, window=w, trace=trace,
                                 return_state=True, **okw)
    plan = make_plan(n_fft, hop, frames, batch, **dict(extra))
    assert plan.path == "frame"               # wave-level FFT frame kernel, not the LDS Stockham kernels
    plan.gla_init(T(init), None, 0.3)
    done, evals = plan.run(6, 3, 0.0, "sc")
    assert done == 6 and len(evals) == 2
    y = N(plan.wave())
    assert rel_l2(y, ref.reshape(y.shape)) < 1e-4, rel_l2(y, ref.reshape(y.shape))
    got = sc_linear(np.array([m for _, m, _ in evals]))
    want = sc_linear(np.array([m for _, m, _ in trace]))
    assert np.abs(got - want).max() < 1e-5
    assert rel_l2(N(plan.state_spec(0)), st["pre_spec"]) < 3e-4


@pytest.mark.parametrize("n_fft,hop,frames,batch,extra", SHAPES[:4])
@pytest.mark.parametrize("rho", [0.1, 1.0])
def test_semi_admm_matches_oracle(n_fft, hop, frames, batch, extra, rho):
    rng = np.random.default_rng(11)
    mag = rng.random((batch, n_fft // 2 + 1, frames), dtype=np.float32) + 0.01
    w = hann(n_fft)
    init = oracle.phase_init(mag, hop_length=hop, window=w)
    ref, st = oracle.admm(init, max_iter=3, rho=rho, tol=0, hop_length=hop, window=w, return_state=True)
    plan = make_plan(n_fft, hop, frames, batch)
    assert plan.path == "frame"
    plan.keep_state()
    plan.admm_init(T(init), None, rho)
    plan.iterate(3)
    tol = 3e-4 if rho == 0.1 else 5e-5
    assert rel_l2(N(plan.wave()), ref.reshape(batch, -1)) < tol
    assert rel_l2(N(plan.state_spec(0)), st["X"]) < tol
    assert rel_l2(N(plan.state_spec(1)), st["U"]) < 20 * tol


@pytest.mark.parametrize("n_fft,hop,frames,batch,extra", SHAPES)
def test_semi_equals_generic(n_fft, hop, frames, batch, extra):
    """Same plan forced onto the generic kernels: waveforms, evaluation sums and state agree to rounding."""
    rng = np.random.default_rng(5)
    mag = T(rng.random((batch, n_fft // 2 + 1, frames), dtype=np.float32) + 0.01)
    fast, gen = make_plan(n_fft, hop, frames, batch, **dict(extra)), make_plan(n_fft, hop, frames, batch, **dict(extra))
    gen.force_generic(True)
    assert fast.path == "frame" and gen.path == "generic"
    out = []
    for p in (fast, gen):
        p.gla_init(None, mag, 0.99)
        p.iterate(4)
        s = p.iterate(1, eval_last=True)
        out.append((N(p.wave()), s, N(p.state_spec(0))))
    # (a handful of frames leave the edges with a tiny envelope: float32 rounding noise is amplified
    # ~10x per iteration by any two implementations, tools/acc_small.py)
    tol = 5e-4 if frames <= 5 else 5e-5
    assert rel_l2(out[0][0], out[1][0]) < tol
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=2e-5)
    assert rel_l2(out[0][2], out[1][2]) < tol


def test_semi_vs_float64_at_speech_size():
    """A 25 ms / 10 ms style analysis (hop does not divide n_fft) at a realistic size against the float64 kernels."""
    n_fft, hop, frames, batch = 1024, 160, 600, 4
    mag = torch.rand((batch, n_fft // 2 + 1, frames), generator=torch.Generator().manual_seed(2)) + 0.01
    p32 = make_plan(n_fft, hop, frames, batch)
    p64 = make_plan(n_fft, hop, frames, batch, dtype=torch.float64)
    assert p32.path == "frame" and p64.path == "generic"
    c0 = p64.phase_init(mag.double().to(DEV))
    p32.gla_init(c0.to(torch.complex64), None, 0.3)
    p64.gla_init(c0, None, 0.3)
    p32.iterate(9)
    p64.iterate(9)
    s32, s64 = p32.iterate(1, eval_last=True), p64.iterate(1, eval_last=True)
    assert rel_l2(N(p32.wave()), N(p64.wave())) < 1e-4
    sc32, sc64 = np.sqrt(s32[0] / s32[2]), np.sqrt(s64[0] / s64[2])
    assert abs(sc32 - sc64) < 1e-5


def test_semi_nan_stays_in_its_frame():
    """centre = False with a Hann window: the envelope is 0 at the first sample, the reference divides 0 by 0 there
    (methods.py:132) and the NaN spreads over the frames that cover it in the next iteration - exactly those."""
    n_fft, hop, frames = 1024, 256, 12
    rng = np.random.default_rng(3)
    mag = rng.random((1, n_fft // 2 + 1, frames), dtype=np.float32) + 0.01
    ref = oracle.griffin_lim(mag, max_iter=2, alpha=0.5, tol=0, hop_length=hop, window=hann(n_fft), center=False)
    y = N(si.griffin_lim(T(mag), max_iter=2, alpha=0.5, tol=0, verbose=False, hop_length=hop,
                         window=torch.from_numpy(hann(n_fft)), center=False))
    assert np.isnan(ref).any() and finite_close(y, ref.reshape(y.shape), 1e-4)


def test_semi_public_api_demo_call():
    """main.py:13-45: griffin_lim(spec, max_iter, alpha=0.3, window, win_length, hop_length=128) on a complex STFT."""
    x = torch.randn(2, 16000, generator=torch.Generator().manual_seed(1)).to(DEV)
    w = torch.hann_window(1024, device=DEV)
    spec = torch.stft(x, 1024, hop_length=128, win_length=1024, window=w, return_complex=True).abs()
    y = si.griffin_lim(spec, max_iter=30, alpha=0.3, tol=0, verbose=False, win_length=1024, window=w, hop_length=128)
    assert y.shape == (2, 16000)
    again = torch.stft(y, 1024, hop_length=128, win_length=1024, window=w, return_complex=True).abs()
    assert float(si.sc(again, spec)) < -12.0         # spectral convergence in dB after 30 iterations


# ---- the frame kernel over chunks of frames with the overlap-add in LDS (k_hop; path code 3) --------------------------
# n_fft, hop, frames, batch, extra: odd hop (frames start at odd samples), hop = n_fft / 16 (long chunk floor), hop above
# n_fft / 2, hop == n_fft (nothing shared between frames), no centring, every pad mode, a normalized transform
HOP_SHAPES = [(1024, 160, 64, 3, {}), (2048, 333, 40, 2, {}), (512, 100, 97, 2, {}), (1024, 64, 70, 2, {}),
              (2048, 768, 33, 2, {}), (512, 512, 24, 2, dict(window=np.ones(512, dtype=np.float32))),
              (1024, 250, 41, 2, dict(center=False, window=np.ones(1024, dtype=np.float32))),
              (2048, 300, 30, 1, dict(pad_mode="circular")), (1024, 192, 48, 2, dict(normalized=True, pad_mode="replicate")),
              (512, 96, 64, 2, dict(pad_mode="constant")), (1024, 256, 40, 2, dict(center=False, window=np.ones(1024, dtype=np.float32))),
              (1024, 200, 36, 2, dict(center=False))]     # Hann without centring: the reference's 0/0 at both ends


@pytest.mark.parametrize("method", ["gla", "admm"])
@pytest.mark.parametrize("n_fft,hop,frames,batch,extra", HOP_SHAPES)
def test_chunked_frame_kernel(n_fft, hop, frames, batch, extra, method, chunked_kernel, monkeypatch):
    """k_hop against the oracle (waveform, spectral state, evaluation sums) and against k_semi + k_ola on the same
    input: the ring adds the frames in the gather's order, so apart from the chunk seams the two agree to rounding."""
    rng = np.random.default_rng(n_fft + hop + frames)
    mag = rng.random((batch, n_fft // 2 + 1, frames), dtype=np.float32) + 0.01
    okw = dict(extra)
    w = okw.pop("window", hann(n_fft))
    init = oracle.phase_init(mag, hop_length=hop, window=w, **okw)
    trace = []
    gla = method == "gla"
    n_it = 6 if gla else 3                      # (ADMM amplifies rounding noise quickly: SURVEY 8c)
    if gla:
        ref, st = oracle.griffin_lim(init, max_iter=n_it, alpha=0.3, tol=0, eva_iter=3, hop_length=hop, window=w,
                                     trace=trace, return_state=True, **okw)
    else:
        ref, st = oracle.admm(init, max_iter=n_it, rho=0.2, tol=0, eva_iter=3, hop_length=hop, window=w, trace=trace,
                              return_state=True, **okw)

    def run(keep=True):
        plan = make_plan(n_fft, hop, frames, batch, **dict(extra))
        plan.keep_state(keep)
        (plan.gla_init if gla else plan.admm_init)(T(init), None, 0.3 if gla else 0.2)
        done, evals = plan.run(n_it, 3, 0.0, "sc")
        return plan, N(plan.wave()), evals

    if gla:
        # the default for Griffin-Lim: the same chunk walk with the momentum carried as a signal (k_hop_td) - waveform and
        # evaluated metric against the oracle; pre_spec is only formed by the kernel below
        plan_td, y_td, evals_td = run(keep=False)
        # (large hops stay on k_hop: the signal form's emission loop costs more there than the state traffic it saves)
        pad_, len_ = (n_fft // 2 if okw.get("center", True) else 0), plan_td.length
        pairs = hop % 2 == 0 and pad_ % 2 == 0 and len_ % 2 == 0
        td_expected = hop <= ({512: 320, 1024: 1024, 2048: 800} if pairs else {512: 128, 1024: 448, 2048: 416})[n_fft]
        assert plan_td.launch_geometry["kernel"] == ("k_hop_td" if td_expected else "k_hop"), plan_td.launch_geometry
        ref_td = ref.reshape(y_td.shape)
        assert np.array_equal(np.isfinite(y_td), np.isfinite(ref_td))
        ok_td = np.isfinite(ref_td)
        assert rel_l2(y_td[ok_td], ref_td[ok_td]) < 1e-4, rel_l2(y_td[ok_td], ref_td[ok_td])
        got_td = sc_linear(np.array([m for _, m, _ in evals_td]))
        want_td = sc_linear(np.array([m for _, m, _ in trace]))
        assert np.array_equal(np.isnan(got_td), np.isnan(want_td))
        assert np.nan_to_num(np.abs(got_td - want_td)).max() < 1e-5
    plan, y, evals = run()
    assert plan.path == "frame" and plan.path_code == 3
    assert plan.launch_geometry["kernel"] == "k_hop"
    ref = ref.reshape(y.shape)
    assert np.array_equal(np.isfinite(y), np.isfinite(ref))
    ok = np.isfinite(ref)
    assert rel_l2(y[ok], ref[ok]) < (1e-4 if gla else 3e-4), rel_l2(y[ok], ref[ok])
    got = sc_linear(np.array([m for _, m, _ in evals]))
    want = sc_linear(np.array([m for _, m, _ in trace]))
    assert np.array_equal(np.isnan(got), np.isnan(want))       # (NaN samples make the whole-batch metric NaN, as in the reference)
    assert np.nan_to_num(np.abs(got - want)).max() < (1e-5 if gla else 1e-4)
    state = N(plan.state_spec(0))
    want_state = st["pre_spec" if gla else "X"]
    okf = np.isfinite(want_state)
    assert rel_l2(state[okf], want_state[okf]) < (2e-5 if gla else 3e-4)
    monkeypatch.setenv("SPECINV_DISABLE_HOP", "1")
    plan2, y2, _ = run()
    assert plan2.path_code == 2
    assert np.array_equal(np.isfinite(y), np.isfinite(y2))
    assert rel_l2(y[ok], y2[ok]) < (2e-6 if gla else 1e-4), rel_l2(y[ok], y2[ok])


def test_chunked_frame_kernel_many_chunks_long_signal(chunked_kernel):
    """A long single item: many chunks, a seam every few frames.  Against the float64 oracle from the same start
    spectrum the error stays at float32 level everywhere (no seam artefacts); iterating in two calls continues from
    the ping-pong buffers."""
    n_fft, hop, frames = 1024, 200, 700
    rng = np.random.default_rng(8)
    mag = rng.random((1, n_fft // 2 + 1, frames), dtype=np.float32) + 0.01
    w = hann(n_fft)
    init = oracle.phase_init(mag, hop_length=hop, window=w)
    ref = oracle.griffin_lim(init.astype(np.complex128), max_iter=5, alpha=0.3, tol=0, hop_length=hop,
                             window=w.astype(np.float64))
    plan = make_plan(n_fft, hop, frames, 1)
    plan.gla_init(T(init), None, 0.3)
    plan.iterate(2)
    plan.iterate(3)
    assert plan.path_code == 3
    y = N(plan.wave())
    assert rel_l2(y, ref.reshape(y.shape)) < 1e-4
    assert np.abs(y - ref.reshape(y.shape)).max() < 1e-3 * np.abs(ref).max()


# SPECINV_EXTRA_SEEDS="a:b": more seeds for an occasional wider sweep (1000:1600 -> 595 of 600 within the tolerance; the
# others hold one of Griffin-Lim's local chaotic events - a bin passing close to zero - where k_hop, k_semi and the
# float32 oracle all leave the float64 oracle by 1e-4 ... 1e-3 in the same three hop-blocks, tools/dbg_hop_seed.py)
_extra = os.environ.get("SPECINV_EXTRA_SEEDS", "")
EXTRA = list(range(*map(int, _extra.split(":")))) if _extra else []


@pytest.mark.parametrize("seed", list(range(40)) + EXTRA)
def test_chunked_frame_kernel_random_shapes(seed, chunked_kernel, monkeypatch):
    """Random (n_fft, hop, frames, batch, centring, pad mode, method): the chunked kernel with the overlap-add in LDS
    and the frame-at-a-time kernel + gather run the same per-frame arithmetic, so after 3 iterations from the same
    start they agree to rounding (the seams and the reciprocal envelope are the only differences); evaluation sums
    too."""
    rng = np.random.default_rng(4000 + seed)
    n_fft = int(rng.choice([512, 1024, 2048]))
    hop = int(rng.choice([int(rng.integers(n_fft // 16, n_fft + 1)), n_fft // 3, n_fft // 5, n_fft // 2 + 1]))
    frames = int(rng.integers(20, 260))
    batch = int(rng.integers(1, 4))
    center = bool(rng.random() < 0.75)
    pad_mode = str(rng.choice(["reflect", "constant", "replicate", "circular"]))
    method = "gla" if rng.random() < 0.6 else "admm"
    w = (hann(n_fft) + np.float32(0.05)) if rng.random() < 0.7 else np.ones(n_fft, dtype=np.float32)
    pad = n_fft // 2 if center else 0
    length = (frames - 1) * hop + n_fft - 2 * pad
    if center and pad_mode in ("reflect", "circular") and pad >= length:
        pytest.skip("torch.stft itself refuses this padding")
    mag = T(rng.random((batch, n_fft // 2 + 1, frames), dtype=np.float32) + 0.02)

    def run(keep=True):
        plan = make_plan(n_fft, hop, frames, batch, window=w, center=center, pad_mode=pad_mode)
        plan.keep_state(keep)
        (plan.gla_init if method == "gla" else plan.admm_init)(None, mag, 0.3 if method == "gla" else 0.5)
        done, evals = plan.run(3, 3, 0.0, "sc")
        return plan.path_code, N(plan.wave()), (N(plan.state_spec(0)) if keep else None), evals[0][1]

    code, y, st, m = run()
    assert code == 3
    if method == "gla":
        # the default kernel (momentum carried as a signal where the hop is small enough): same iterates up to where the linear
        # combination is rounded
        code_td, y_td, _, m_td = run(keep=False)
        assert code_td == 3 and np.array_equal(np.isfinite(y), np.isfinite(y_td))
        okt = np.isfinite(y)
        assert rel_l2(y_td[okt], y[okt]) < 2e-5, (rel_l2(y_td[okt], y[okt]), n_fft, hop, frames, batch, center, pad_mode)
        assert abs(m_td - m) < 1e-3 * max(1.0, abs(m))
    monkeypatch.setenv("SPECINV_DISABLE_HOP", "1")
    code2, y2, st2, m2 = run()
    assert code2 == 2
    tol = 5e-6 if method == "gla" else 2e-4
    assert np.array_equal(np.isfinite(y), np.isfinite(y2))
    ok = np.isfinite(y2)
    assert rel_l2(y[ok], y2[ok]) < tol, (rel_l2(y[ok], y2[ok]), n_fft, hop, frames, batch, center, pad_mode, method)
    assert rel_l2(st, st2) < tol
    assert abs(m - m2) < 1e-3 * max(1.0, abs(m2))
