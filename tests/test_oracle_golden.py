"""Pins the CPU oracle (oracle/) against the golden fixtures captured from the
unmodified reference (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

import oracle
from oracle.lbfgs import LogMelStft, MagStft
from _util import (G0_CASES, finite_close, g0_kwargs, hann, load_golden, rel_l2, sc_linear,
                   sweep_kwargs)
from spectrogram_inversion_amd.mel import mel_filterbank


# ---- G0: stft / istft / envelope ------------------------------------------- #
@pytest.mark.parametrize("i", range(len(G0_CASES)))
def test_stft_istft(i):
    g = load_golden("g0_stft")
    c = G0_CASES[i]
    kw = g0_kwargs(c)
    x = g["x"]
    spec_ref = g[f"spec{i}"]
    a = oracle.args_helper(spec_ref.shape[-2], np.float32, **kw)
    assert a.n_fft == c["n_fft"]
    np.testing.assert_array_equal(a.window, g[f"win{i}"])
    s = oracle.stft(x, a)
    assert s.shape == spec_ref.shape
    assert rel_l2(s, spec_ref) < 2e-6
    y, env = oracle.istft(spec_ref, a)
    assert y.shape == g[f"istft{i}"].shape
    assert finite_close(env, g[f"env{i}"], 1e-6)
    # compare before the envelope division: with center=False the Hann envelope is ~0 at the
    # edges and the division amplifies float32 rounding without bound (methods.py:132)
    e = g[f"env{i}"]
    assert finite_close(y * e, g[f"istft{i}"] * e, 2e-5)


# ---- G1: phase_init --------------------------------------------------------- #
@pytest.mark.parametrize("i", range(4))
def test_phase_init(i):
    g = load_golden("g1_phase_init")
    mag = g[f"mag{i}"]
    hop = int(g[f"hop{i}"])
    kw = {"hop_length": hop} if hop else {}
    out = oracle.phase_init(mag, **kw)
    ref = g[f"out{i}"]
    assert out.shape == ref.shape and out.dtype == ref.dtype
    # <= 2 ulp of the magnitude on each component; in practice ~1 ulp
    err = np.abs(out - ref).max()
    assert err <= 4 * np.finfo(np.float32).eps * np.abs(ref).max(), err
    out64 = oracle.phase_init(mag.astype(np.float64), **kw)
    assert np.abs(out64 - g[f"out64_{i}"]).max() < 1e-9


# ---- G2: griffin_lim -------------------------------------------------------- #
@pytest.mark.parametrize("alpha", [0.0, 0.3, 0.99])
@pytest.mark.parametrize("it", [1, 10, 100])
def test_gla_waveforms(alpha, it):
    g = load_golden("g2_gla")
    kw = dict(hop_length=int(g["hop"]), window=g["window"])
    key = f"a{alpha}_it{it}"
    trace = []
    y = oracle.griffin_lim(g["init"], max_iter=it, alpha=alpha, tol=0, eva_iter=10, trace=trace, **kw)
    ref, ref64 = g["wave_" + key], g["wave64_" + key]
    noise = rel_l2(ref, ref64)                        # the reference's own fp32 noise floor
    assert y.shape == ref.shape
    assert rel_l2(y, ref) < max(4 * noise, 2e-6), (rel_l2(y, ref), noise)
    if it == 100:
        tr = g["trace_" + key]
        got = np.array([[m, l] for _, m, l in trace])
        assert got.shape == tr.shape
        assert np.abs(sc_linear(got[:, 0]) - sc_linear(tr[:, 0])).max() < 1e-5
        np.testing.assert_allclose(got[:, 1], tr[:, 1], rtol=2e-4)


@pytest.mark.parametrize("fixture", ["g14_wellcond", "g15_wellcond_1024", "g16a_wellcond_2048"])
@pytest.mark.parametrize("alpha", [0.0, 0.3, 0.99])
def test_gla_wellconditioned_100_iterations(alpha, fixture):
    """g14 / g15 / g16a (n_fft 512 / 1024 / 2048): consistent magnitudes, perturbed true phase - 100 iterations stay well-conditioned (the reference's own
    float32-vs-float64 distance is 1e-6), so the waveform gate is the strict one."""
    g = load_golden(fixture)
    y = oracle.griffin_lim(g["init"], max_iter=100, alpha=alpha, tol=0, eva_iter=10, hop_length=int(g["hop"]), window=g["window"])
    ref, ref64 = g[f"wave_a{alpha}"], g[f"wave64_a{alpha}"]
    noise = rel_l2(ref, ref64)
    assert noise < 3e-6, noise
    assert rel_l2(y, ref) < min(1e-4, max(6 * noise, 3e-6)), (rel_l2(y, ref), noise)


def test_c2_headline_items_against_the_reference_run():
    """g16b: BASELINE configs[1] through the unmodified reference (B = 64, n_fft 2048, hop 512, 1024 frames, 100 iterations,
    alpha 0.3, magnitudes default_rng(1234) like bench.py's rank 0).  The oracle inverts items 0 and 63 (batch items are
    independent: the fixture records that the reference's B = 2 run of them equals their rows of the B = 64 run bit for bit):
    the per-pair metric trace within 1e-5 (linear SC) of the reference's float32 trace of the same pair, and the waveforms as close
    to the reference's float32 result as that is to its own float64 run, hop segment by hop segment."""
    g = load_golden("g16b_c2_headline")
    assert bool(g["rows_equal_pair_run"].all())
    items = [int(i) for i in g["items"]]
    mag = np.random.default_rng(int(g["seed"])).random((64, 1025, 1024), dtype=np.float32)
    chk = g["mag_checksum"]
    assert float(mag.astype(np.float64).sum()) == chk[0] and float(mag[0, 5, 7]) == chk[1] and float(mag[63, 1024, 1023]) == chk[2]
    hop, win = int(g["hop"]), g["window"]
    trace = []
    y = oracle.griffin_lim(np.ascontiguousarray(mag[items]), max_iter=100, alpha=0.3, tol=0, eva_iter=10, hop_length=hop, window=win,
                           trace=trace)
    got = np.array([[m, l] for _, m, l in trace])
    want = g["trace_pair"]
    assert np.abs(sc_linear(got[:, 0]) - sc_linear(want[:, 0])).max() < 1e-5, np.abs(sc_linear(got[:, 0]) - sc_linear(want[:, 0])).max()
    np.testing.assert_allclose(got[:, 1], want[:, 1], rtol=2e-4)
    for k, it in enumerate(items):
        ref = g[f"wave_{it}"].astype(np.float64)
        seg = np.linalg.norm((y[k] - ref).reshape(-1, hop), axis=1) / np.maximum(g[f"segnorm64_{it}"], 1e-30)
        own = g[f"segerr_{it}"]                         # the reference's float32 vs its float64, same segments
        assert np.median(seg) < 4 * np.median(own) + 1e-6, (it, np.median(seg), np.median(own))
        assert np.quantile(seg, 0.9) < 6 * np.quantile(own, 0.9) + 1e-5, (it, np.quantile(seg, 0.9), np.quantile(own, 0.9))
        assert rel_l2(y[k], ref) < max(10 * float(g[f"noise_{it}"]), 1e-4), (it, rel_l2(y[k], ref), float(g[f"noise_{it}"]))


@pytest.mark.parametrize("metric", ["snr", "ser"])
def test_gla_from_magnitude(metric):
    g = load_golden("g2_gla")
    kw = dict(hop_length=int(g["hop"]), window=g["window"])
    trace = []
    y = oracle.griffin_lim(g["mag"], max_iter=20, alpha=0.3, tol=0, eva_iter=5, metric=metric,
                           trace=trace, **kw)
    assert rel_l2(y, g["wave_mag_" + metric]) < 1e-4
    got = np.array([m for _, m, _ in trace])
    np.testing.assert_allclose(got, g["trace_mag_" + metric][:, 0], atol=2e-3)


def test_gla_early_stop_and_shapes():
    g = load_golden("g2_gla")
    kw = dict(hop_length=int(g["hop"]), window=g["window"])
    y, st = oracle.griffin_lim(g["init"], max_iter=2000, eva_iter=10, return_state=True, **kw)
    # the stop test compares float32-noise-level loss differences, so allow one evaluation of slack
    assert abs(st["iters"] - int(g["iters_tol"])) <= 10, (st["iters"], int(g["iters_tol"]))
    y2 = oracle.griffin_lim(g["mag"][0], max_iter=3, alpha=0.3, tol=0, **kw)
    assert y2.shape == g["wave_2d"].shape and rel_l2(y2, g["wave_2d"]) < 1e-5
    y3 = oracle.griffin_lim(g["mag"][:1], max_iter=3, alpha=0.3, tol=0, **kw)
    assert y3.shape == g["wave_1ft"].shape and rel_l2(y3, g["wave_1ft"]) < 1e-5


# ---- G3: stft-kwarg sweep --------------------------------------------------- #
def _sweep_ids():
    return list(range(len(load_golden("g3_sweep")["meta"])))


@pytest.mark.parametrize("i", _sweep_ids())
def test_kwarg_sweep(i):
    g = load_golden("g3_sweep")
    kw = sweep_kwargs(g["meta"][i])
    spec = g[f"spec{i}"]
    y = oracle.griffin_lim(spec, max_iter=2, alpha=0.5, **kw)
    assert y.shape == g[f"gla{i}"].shape
    assert finite_close(y, g[f"gla{i}"], 5e-5), g["meta"][i]
    z = oracle.admm(spec, max_iter=2, rho=0.5, **kw)
    assert finite_close(z, g[f"admm{i}"], 5e-5), g["meta"][i]


# ---- G4: ADMM --------------------------------------------------------------- #
@pytest.mark.parametrize("rho", [0.1, 1.0])
@pytest.mark.parametrize("it", [1, 2, 5])
def test_admm_waveforms(rho, it):
    g = load_golden("g4_admm")
    kw = dict(hop_length=int(g["hop"]), window=g["window"])
    key = f"r{rho}_it{it}"
    y = oracle.admm(g["init"], max_iter=it, rho=rho, tol=0, **kw)
    ref, ref64 = g["wave_" + key], g["wave64_" + key]
    # rho=0.1 amplifies rounding differences ~10x per iteration (5.8e-7 at 1 it,
    # 1e-4 at 5 it in the reference's own fp32-vs-fp64 comparison)
    tol = {1: 5e-6, 2: 5e-5, 5: 5e-4}[it]
    assert rel_l2(y, ref) < tol, (rel_l2(y, ref), rel_l2(ref, ref64))


@pytest.mark.parametrize("rho", [0.1, 1.0])
def test_admm_200_sc(rho):
    g = load_golden("g4_admm")
    kw = dict(hop_length=int(g["hop"]), window=g["window"])
    trace = []
    oracle.admm(g["init"], max_iter=200, rho=rho, tol=0, eva_iter=10, trace=trace, **kw)
    tr, tr64 = g[f"trace_r{rho}_it200"], g[f"trace64_r{rho}_it200"]
    got = sc_linear(np.array([m for _, m, _ in trace]))
    # rho=0.1 is chaotic w.r.t. rounding: gate at the reference's own fp32/fp64 spread
    spread = np.abs(sc_linear(tr[:, 0]) - sc_linear(tr64[:, 0])).max()
    assert np.abs(got - sc_linear(tr[:, 0])).max() < max(3 * spread, 1e-5)


# ---- G5: RTISI-LA ----------------------------------------------------------- #
def _rtisi_ids():
    return list(range(len(load_golden("g5_rtisi")["meta"])))


@pytest.mark.parametrize("i", _rtisi_ids())
def test_rtisi(i):
    g = load_golden("g5_rtisi")
    hop, la, asym, alpha = str(g["meta"][i]).split("|")
    mag = g[f"mag_h{hop}"]
    y = oracle.rtisi_la(mag, look_ahead=int(la), asymmetric_window=bool(int(asym)), max_iter=3,
                        alpha=float(alpha), hop_length=int(hop), window=g["window"])
    ref, ref64 = g[f"wave{i}"], g[f"wave64_{i}"]
    noise = rel_l2(ref, ref64)
    assert y.shape == ref.shape
    # asymmetric_window=False amplifies rounding noise (zero-phase first frame)
    tol = max(5 * noise, 1e-5) if int(asym) else max(30 * noise, 1e-4)
    assert rel_l2(y, ref) < tol, (g["meta"][i], rel_l2(y, ref), noise)


@pytest.mark.parametrize("asym", [True, False])
def test_rtisi_single_steps(asym):
    g = load_golden("g5_rtisi")
    y = oracle.rtisi_la(g["mag_single"], look_ahead=1, asymmetric_window=asym, max_iter=1, alpha=0.99,
                        hop_length=64, window=hann(256))
    assert rel_l2(y, g[f"wave_single_asym{int(asym)}"]) < (1e-5 if asym else 1e-4)


@pytest.mark.parametrize("j", range(3))
@pytest.mark.parametrize("asym", [True, False])
def test_rtisi_stft_options(j, asym):
    g = load_golden("g5_rtisi")
    opts = [dict(win_length=300, window=None, hop_length=None, center=True, normalized=True, onesided=True),
            dict(win_length=300, window=hann(300), hop_length=128, center=False, normalized=False,
                 onesided=False),
            dict(win_length=None, window=None, hop_length=128, center=True, normalized=False, onesided=False)]
    y = oracle.rtisi_la(g[f"opt{j}_spec"], look_ahead=2, asymmetric_window=asym, max_iter=2, **opts[j])
    ref = g[f"opt{j}_asym{int(asym)}"]
    assert y.shape == ref.shape
    assert finite_close(y, ref, 2e-3), (j, asym)


# ---- G6: L-BFGS ------------------------------------------------------------- #
def test_lbfgs_mag_gradient():
    g = load_golden("g6_lbfgs")
    a = oracle.args_helper(129, np.float32)          # torch.stft(x, 256): rectangular window, hop 64
    tr = MagStft(a)
    loss, grad = tr.loss_grad(g["mag_x0"], g["mag_spec"])
    assert abs(loss - float(g["mag_loss0"])) < 1e-5 * float(g["mag_loss0"])
    assert rel_l2(grad, g["mag_grad0"]) < 1e-5


@pytest.mark.parametrize("tag,kw", [("plain", dict(max_iter=10)),
                                    ("wolfe", dict(max_iter=10, line_search_fn="strong_wolfe")),
                                    ("hist3", dict(max_iter=12, history_size=3))])
def test_lbfgs_mag_trajectory(tag, kw):
    g = load_golden("g6_lbfgs")
    a = oracle.args_helper(129, np.float32)
    trace = []
    x = oracle.l_bfgs(g["mag_spec"], MagStft(a), init_x0=g["mag_x0"], outer_max_iter=2, tol=0,
                      eva_iter=1, trace=trace, **kw)
    ref_tr = g[f"mag_trace_{tag}"]
    got = np.array([[m, l] for _, m, l in trace])
    if tag == "wolfe":
        # the cubic interpolation's discriminant flips on 1-ulp float32 loss differences in this
        # problem (checked against the real optimiser), so the trajectory itself cannot be pinned;
        # the search logic is pinned in float64 by test_lbfgs_rosenbrock instead
        assert abs(got[0, 1] - ref_tr[0, 1]) < 0.1 * ref_tr[0, 1], (got, ref_tr)
        return
    # first outer step is tight; later ones inherit float32 dot-product noise
    assert abs(got[0, 1] - ref_tr[0, 1]) < 2e-3 * ref_tr[0, 1], (got, ref_tr)
    assert rel_l2(x, g[f"mag_x_{tag}"]) < 5e-2


@pytest.mark.parametrize("tag,kw", [("wolfe", dict(max_iter=40, history_size=5, line_search_fn="strong_wolfe")),
                                    ("wolfe_h100", dict(max_iter=25, line_search_fn="strong_wolfe")),
                                    ("fixed", dict(max_iter=30, lr=1e-3, history_size=4))])
def test_lbfgs_rosenbrock(tag, kw):
    """float64 analytic problem: the restated optimiser must retrace torch.optim.LBFGS exactly."""
    from oracle.lbfgs import LbfgsState, lbfgs_minimize
    g = load_golden("g9_lbfgs_rosen")
    losses = []

    def fg(x):
        a, b = x[1:] - x[:-1] ** 2, 1.0 - x[:-1]
        f = float((100.0 * a * a + b * b).sum())
        gr = np.zeros_like(x)
        gr[1:] += 200.0 * a
        gr[:-1] += -400.0 * a * x[:-1] - 2.0 * b
        losses.append(f)
        return f, gr

    x = g["x0"].copy()
    st = LbfgsState()
    for _ in range(2):
        lbfgs_minimize(fg, x, st, **kw)
    ref = g[f"losses_{tag}"]
    assert len(losses) == len(ref)
    np.testing.assert_allclose(losses, ref, rtol=5e-4, atol=1e-7)
    np.testing.assert_allclose(x, g[f"x_{tag}"], rtol=1e-4, atol=1e-6)


def test_lbfgs_mag_three_inner():
    g = load_golden("g6_lbfgs")
    a = oracle.args_helper(129, np.float32)
    x = oracle.l_bfgs(g["mag_spec"], MagStft(a), init_x0=g["mag_x0"], outer_max_iter=1, tol=0,
                      eva_iter=1, max_iter=3)
    assert rel_l2(x, g["mag_x_3inner"]) < 1e-4


def test_logmel_forward_gradient():
    g = load_golden("g6_lbfgs")
    fb = mel_filterbank(22050, 2048, 80)
    np.testing.assert_array_equal(fb[:, ::64], g["mel_fb_check"])
    a = oracle.args_helper(1025, np.float32, hop_length=512, window=hann(2048))
    tr = LogMelStft(a, fb)
    v = tr.forward(g["mel_x"])
    assert rel_l2(v, g["mel_fwd"]) < 1e-5
    loss, grad = tr.loss_grad(g["mel_x"], g["mel_target"])
    assert abs(loss - float(g["mel_loss"])) < 1e-5 * float(g["mel_loss"])
    assert rel_l2(grad, g["mel_grad"]) < 1e-5


@pytest.mark.parametrize("tag,kw", [("plain", {}), ("wolfe", dict(line_search_fn="strong_wolfe"))])
def test_logmel_first_outer_step(tag, kw):
    g = load_golden("g6_lbfgs")
    fb = mel_filterbank(22050, 2048, 80)
    a = oracle.args_helper(1025, np.float32, hop_length=512, window=hann(2048))
    trace = []
    x = oracle.l_bfgs(g["mel_target"], LogMelStft(a, fb), init_x0=g["mel_x"], outer_max_iter=1, tol=0,
                      eva_iter=1, trace=trace, **kw)
    ref_tr = g[f"mel_trace_{tag}"]
    assert abs(trace[0][2] - ref_tr[0, 1]) < 5e-3 * ref_tr[0, 1], (trace, ref_tr)
    assert rel_l2(x, g[f"mel_x_{tag}"]) < 2e-2


# ---- G7 / G8 ---------------------------------------------------------------- #
def test_metrics():
    g = load_golden("g7_metrics")
    a, b = g["a"], g["b"]
    got = np.array([oracle.sc(a, b), oracle.snr(a, b), oracle.ser(a, b), oracle.mse(a, b)])
    np.testing.assert_allclose(got, g["vals64"], rtol=1e-6)
    np.testing.assert_allclose(got, g["vals"], rtol=1e-4, atol=1e-4)


def test_float64():
    g = load_golden("g8_f64")
    kw = dict(hop_length=64, window=g["window"])
    init = g["init"]
    assert np.abs(oracle.phase_init(g["mag"], **kw) - init).max() < 1e-9
    assert rel_l2(oracle.griffin_lim(init, max_iter=1, alpha=0.3, tol=0, **kw), g["gla1"]) < 1e-12
    assert rel_l2(oracle.griffin_lim(init, max_iter=5, alpha=0.3, tol=0, **kw), g["gla5"]) < 1e-11
    assert rel_l2(oracle.admm(init, max_iter=1, rho=0.1, tol=0, **kw), g["admm1"]) < 1e-12
    assert rel_l2(oracle.admm(init, max_iter=5, rho=0.1, tol=0, **kw), g["admm5"]) < 1e-10
    y = oracle.rtisi_la(g["mag"], look_ahead=2, asymmetric_window=True, max_iter=2, **kw)
    assert y.dtype == np.float64 and rel_l2(y, g["rtisi"]) < 1e-9


# ---- G13: the shapes the wave-level kernels specialise in ----------------------------------- #
@pytest.mark.parametrize("tag", [str(m) for m in load_golden("g13_wave_level_shapes")["meta"]])
def test_wave_level_shapes(tag):
    g = load_golden("g13_wave_level_shapes")
    n_fft, hop = (int(v) for v in tag.split("_"))
    w = hann(n_fft)
    init = g[f"init_{tag}"]
    assert rel_l2(oracle.phase_init(g[f"mag_{tag}"], hop_length=hop, window=w), init) < 1e-6
    y = oracle.griffin_lim(init, max_iter=5, alpha=0.3, tol=0, hop_length=hop, window=w)
    assert rel_l2(np.asarray(y).reshape(g[f"gla_{tag}"].shape), g[f"gla_{tag}"]) < 1e-4
    z = oracle.admm(init, max_iter=3, rho=1.0, tol=0, hop_length=hop, window=w)
    assert rel_l2(np.asarray(z).reshape(g[f"admm_{tag}"].shape), g[f"admm_{tag}"]) < 1e-4
    if f"rtisi_{tag}" in g.files:
        la = -1 if hop * 8 > n_fft else 3
        r = oracle.rtisi_la(g[f"mag_{tag}"][:, :, :12].astype(np.float64), look_ahead=la, asymmetric_window=True, max_iter=2,
                            alpha=0.99, hop_length=hop, window=hann(n_fft, np.float64))
        assert rel_l2(np.asarray(r).reshape(g[f"rtisi64_{tag}"].shape), g[f"rtisi64_{tag}"]) < 1e-9
