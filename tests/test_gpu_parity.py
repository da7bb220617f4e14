"""Parity of the HIP path (through the C ABI) against the golden fixtures captured from the
reference and against the CPU oracle on seeded inputs.  Needs an MI355X: `-m gpu`."""
import numpy as np
import pytest
import torch

import oracle
from _util import (G0_CASES, segment_errors, finite_close, g0_kwargs, hann, load_golden, rel_l2, sc_linear,
                   sweep_kwargs)

pytestmark = pytest.mark.gpu

import spectrogram_inversion_amd as si                      # noqa: E402
from spectrogram_inversion_amd.plan import Plan, args_helper, get_plan   # noqa: E402


def dev():
    return torch.device("cuda", 0)


def T(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(dev()) if dtype is None else t.to(dev(), dtype)


def tkw(kw):
    """numpy stft kwargs -> torch kwargs (window as tensor)."""
    out = dict(kw)
    if out.get("window") is not None:
        out["window"] = torch.from_numpy(np.asarray(out["window"]))
    return out


def N(t):
    return t.detach().cpu().numpy()


def plan_for(spec_shape, dtype, **kw):
    b, f, t = spec_shape
    probe = torch.empty((1, f, 1), dtype=dtype)
    a = args_helper(probe, **tkw(kw))
    return get_plan(a, b, t, dtype, dev())


# ---- library really is the HIP one ----------------------------------------------------------
def test_native_library_loaded():
    from spectrogram_inversion_amd import _lib
    lib = _lib.load()
    assert lib.specinv_abi_version() == 1
    with open("/proc/self/maps") as fh:
        assert "libspecinv.so" in fh.read()


# ---- G0: stft / istft / envelope --------------------------------------------------------------
@pytest.mark.parametrize("i", range(len(G0_CASES)))
def test_stft_istft(i):
    g = load_golden("g0_stft")
    kw = g0_kwargs(G0_CASES[i])
    spec_ref = g[f"spec{i}"]
    plan = plan_for(spec_ref.shape, torch.float32, **kw)
    s = N(plan.stft(T(g["x"])))
    assert s.shape == spec_ref.shape
    assert rel_l2(s, spec_ref) < 3e-6
    env = N(plan.envelope())
    assert finite_close(env, g[f"env{i}"], 1e-6)
    y = N(plan.istft(T(spec_ref)))
    e = g[f"env{i}"]
    assert y.shape == g[f"istft{i}"].shape
    assert finite_close(y * e, g[f"istft{i}"] * e, 2e-5)


# ---- G1: phase_init -------------------------------------------------------------------------------
@pytest.mark.parametrize("i", range(4))
def test_phase_init(i):
    g = load_golden("g1_phase_init")
    mag = g[f"mag{i}"]
    hop = int(g[f"hop{i}"])
    kw = {"hop_length": hop} if hop else {}
    out = N(si.phase_init(T(mag), **kw))
    ref = g[f"out{i}"]
    assert out.shape == ref.shape and out.dtype == ref.dtype
    # bit-exact up to the last-place rounding of cos/sin (<= 2 ulp of the magnitude)
    err = np.abs(out - ref).max()
    assert err <= 4 * np.finfo(np.float32).eps * np.abs(ref).max(), err
    out64 = N(si.phase_init(T(mag.astype(np.float64)), **kw))
    assert np.abs(out64 - g[f"out64_{i}"]).max() < 1e-9


# ---- G2: griffin_lim ---------------------------------------------------------------------------------
# g2's magnitudes are random (inconsistent): at 100 iterations of alpha = 0.3 ONE bin parts float32 runs.  Diagnosis
# (tools/near_zero_event.py, profiles/r04_near_zero_event.txt): at iteration 95 the DC bin of frame 8 of item 0 - a REAL number -
# passes zero: S = -8.5e-7 in float64 (3.5e-6 of its target magnitude, where the frame's typical ratio is 0.8), +7.6e-8 on the
# float32 frame kernel.  The projection S m / |S| turns that into -m or +m: a sign flip of one bin, 1.3e-2 of that hop segment at
# once, 2.6e-3 five iterations later, 4.9e-4 of the whole waveform at iteration 100 (the reference's own float32 and float64 runs
# happen to fall on the same side).  Which side a kernel lands on is decided by 1e-7 of rounding in its FFT, not by the
# projection's arithmetic: rounds 2 / 3 / 4 measured the event on the fused kernels with the approximate projection, on the frame
# kernel with IEEE divisions and with one-rounding reciprocals, on the pre_spec kernel with two-rounding reciprocals
# (tools/refchain_study.py, profiles/r04_refchain.txt); every other of the 45 strict-gate cases passes on every arithmetic.  The
# cases listed here keep the segment-distribution form of the gate.
STRICT_XFAIL = {(0.3, 100, "frame")}


def _strict_or_segments(y, ref, gate, hop, key, err):
    if key in STRICT_XFAIL:
        seg = segment_errors(y, ref, hop)
        assert np.quantile(seg, 0.75) < gate and seg.max() < 3e-2, (key, err, np.quantile(seg, 0.75), seg.max())
    else:
        assert err < gate, (key, err)


@pytest.mark.parametrize("alpha", [0.0, 0.3, 0.99])
@pytest.mark.parametrize("it", [1, 10, 100])
def test_gla_waveforms(alpha, it):
    """The drop-in `griffin_lim` on its default arithmetic (round 4: the reference's operation order in the projection, a true
    envelope division) against the reference's waveforms: the STRICT gate - rel-L2 <= 1e-4 (north-star bar) and within 6 x the
    reference's own float32-vs-float64 noise - at 1, 10 and 100 iterations, and the spectral convergence to 1e-5."""
    g = load_golden("g2_gla")
    kw = dict(hop_length=int(g["hop"]), window=torch.from_numpy(g["window"]))
    key = f"a{alpha}_it{it}"
    y = N(si.griffin_lim(T(g["init"]), max_iter=it, alpha=alpha, tol=0, verbose=False, eva_iter=10, **kw))
    ref, ref64 = g["wave_" + key], g["wave64_" + key]
    noise = rel_l2(ref, ref64)
    assert y.shape == ref.shape
    gate = min(1e-4, max(6 * noise, 3e-6))
    # (a problem this small runs on the frame kernel; its one near-zero event keeps the segment form of the gate)
    _strict_or_segments(y, ref, gate, int(g["hop"]), (alpha, it, "frame"), rel_l2(y, ref))
    if it == 100:
        w = g["window"]
        a = oracle.args_helper(g["init"].shape[1], np.float32, hop_length=int(g["hop"]), window=w)
        target = np.abs(g["init"])
        sc_y = np.linalg.norm(np.abs(oracle.stft(y, a)) - target) / np.linalg.norm(target)
        sc_ref = np.linalg.norm(np.abs(oracle.stft(ref, a)) - target) / np.linalg.norm(target)
        assert abs(sc_y - sc_ref) < 1e-5, (sc_y, sc_ref)


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("path", ["frame", "fused", "fused_prespec"])
@pytest.mark.parametrize("alpha", [0.0, 0.3, 0.99])
@pytest.mark.parametrize("it", [10, 100])
def test_gla_waveforms_every_kernel_and_arithmetic(alpha, it, path, exact, monkeypatch):
    """The g2 waveforms after 10 and 100 iterations on each float32 wave-level kernel - the frame kernel (default for a problem this
    small), the signal-form fused kernel, the fused kernel on pre_spec - with the default arithmetic (`set_exact(True)`: the
    reference's operation order, (S m) r with r the correctly rounded 1 / |S|, torch_specinv/methods.py:246-247; a true division by
    the envelope, :132) and with the approximate copies (`set_exact(False)`).  Exact arithmetic: the STRICT gate
    min(1e-4, 6 x the reference's float32-vs-float64 noise), the one near-zero event excepted (STRICT_XFAIL).  Approximate
    arithmetic: the strict gate at 10 iterations, the segment-distribution form at 100."""
    from spectrogram_inversion_amd.plan import Plan
    if not exact and not si.has_approx():
        pytest.skip("the approximate-projection kernels are not in this build (SPECINV_BUILD_APPROX=1)")
    g = load_golden("g2_gla")
    hop, w = int(g["hop"]), torch.from_numpy(g["window"])
    key = f"a{alpha}_it{it}"
    ref, ref64 = g["wave_" + key], g["wave64_" + key]
    noise = rel_l2(ref, ref64)
    gate = min(1e-4, max(6 * noise, 3e-6))
    init = T(g["init"])
    if path != "frame":
        monkeypatch.setenv("SPECINV_SMALL_FRAMES", "0")
    p = Plan(args_helper(init, hop_length=hop, window=w), init.shape[0], init.shape[2], torch.float32, dev())
    p.set_exact(exact)
    p.keep_state(path == "fused_prespec")
    p.gla_init(init, None, alpha)
    want = {"frame": "k_semi", "fused": "k_fused_td", "fused_prespec": "k_fused"}[path]
    assert p.launch_geometry["kernel"] == want, p.launch_geometry
    p.run(it, 10, 0.0, "sc")
    y = N(p.wave())
    err = rel_l2(y, ref)
    if exact or it < 100:
        _strict_or_segments(y, ref, gate, hop, (alpha, it, path), err)
    else:
        seg = segment_errors(y, ref, hop)
        assert np.quantile(seg, 0.75) < gate and seg.max() < 3e-2, (alpha, path, err, np.quantile(seg, 0.75), seg.max())


def test_exact_projection_switch_of_the_drop_in_functions(monkeypatch):
    """The module-level switch (the drop-in signatures are the reference's): the default is the reference's operation order;
    `set_exact_projection(False)` / SPECINV_EXACT=0 make `griffin_lim` / `ADMM` take the approximate kernels - the result equals a
    plan run with `set_exact(False)` and differs from the default arithmetic's in the last bits only."""
    from spectrogram_inversion_amd.plan import Plan, clear_plan_cache
    g = load_golden("g2_gla")
    kw = dict(hop_length=int(g["hop"]), window=torch.from_numpy(g["window"]))
    init = T(g["init"])
    monkeypatch.delenv("SPECINV_EXACT", raising=False)
    if not si.has_approx():
        # a default build (round 6) does not carry the approximate copies: the switch is accepted and changes nothing
        want = N(si.griffin_lim(init, max_iter=10, alpha=0.3, tol=0, verbose=False, **kw))
        try:
            si.set_exact_projection(False)
            assert np.array_equal(N(si.griffin_lim(init, max_iter=10, alpha=0.3, tol=0, verbose=False, **kw)), want)
        finally:
            si.set_exact_projection(None)
        return
    exact = N(si.griffin_lim(init, max_iter=10, alpha=0.3, tol=0, verbose=False, **kw))
    exact_admm = N(si.ADMM(init, max_iter=3, rho=1.0, tol=0, verbose=False, **kw))
    try:
        si.set_exact_projection(False)
        approx = N(si.griffin_lim(init, max_iter=10, alpha=0.3, tol=0, verbose=False, **kw))
        approx_admm = N(si.ADMM(init, max_iter=3, rho=1.0, tol=0, verbose=False, **kw))
    finally:
        si.set_exact_projection(None)
    monkeypatch.setenv("SPECINV_EXACT", "0")
    assert np.array_equal(N(si.griffin_lim(init, max_iter=10, alpha=0.3, tol=0, verbose=False, **kw)), approx)
    monkeypatch.delenv("SPECINV_EXACT")
    assert np.array_equal(N(si.griffin_lim(init, max_iter=10, alpha=0.3, tol=0, verbose=False, **kw)), exact)
    for on, want in ((True, exact), (False, approx)):
        p = Plan(args_helper(init, **kw), init.shape[0], init.shape[2], torch.float32, dev())
        p.set_exact(on)
        p.gla_init(init, None, 0.3)
        p.iterate(10)
        assert np.array_equal(N(p.wave()), want)
    assert not np.array_equal(exact, approx) and rel_l2(exact, approx) < 1e-5
    ref = oracle.admm(g["init"], max_iter=3, rho=1.0, tol=0, hop_length=int(g["hop"]), window=g["window"])
    assert rel_l2(exact_admm, ref) < 1e-5 and rel_l2(approx_admm, ref) < 1e-5


@pytest.mark.parametrize("path", ["default", "fused", "fused_prespec", "frame_lds", "frame_lds_prespec", "generic", "generic_workgroup", "float64",
                                  "float64_workgroup"])
@pytest.mark.parametrize("alpha", [0.0, 0.3, 0.99])
def test_gla_wellconditioned_100_iterations(alpha, path, monkeypatch):
    """g14 (consistent magnitudes of a real signal, true phase perturbed by 0.5 rad): 100 iterations held to the STRICT
    gate min(1e-4, 6 x the reference's float32-vs-float64 noise) - no segment statistics - on every kernel path: the
    frame kernel (default for a problem this small), the fused chunk-walking kernel with the momentum carried as a signal
    (`k_fused_td<4, 4>`) and on pre_spec itself (`k_fused<4, 4>`), the chunked frame kernel with the overlap-add in LDS, the
    generic kernels (the wave-level coverage kernel `k_wave_iter` with the overlap-add in its registers since round 6, and the
    workgroup-level kernel it replaced), and the same two in float64.  (g2's random magnitudes are inconsistent: there a near-zero
    bin can decorrelate a neighbourhood between ANY two float32 runs, see test_gla_waveforms.)"""
    from spectrogram_inversion_amd.plan import Plan, clear_plan_cache
    g = load_golden("g14_wellcond")
    ref, ref64 = g[f"wave_a{alpha}"], g[f"wave64_a{alpha}"]
    noise = rel_l2(ref, ref64)
    gate = min(1e-4, max(6 * noise, 3e-6))
    init = T(g["init"])
    hop, w = int(g["hop"]), torch.from_numpy(g["window"])
    if path in ("fused", "fused_prespec", "frame_lds", "frame_lds_prespec"):
        monkeypatch.setenv("SPECINV_SMALL_FRAMES", "0")
    if path in ("frame_lds", "frame_lds_prespec"):
        monkeypatch.setenv("SPECINV_DISABLE_FUSED", "1")
    if path.startswith("float64"):
        init, w = init.to(torch.complex128), w.double()
    if path.endswith("_workgroup"):
        monkeypatch.setenv("SPECINV_GENERIC_WAVE", "0")
    p = Plan(args_helper(init, hop_length=hop, window=w), init.shape[0], init.shape[2], w.dtype, dev())
    if path.startswith("generic"):
        p.force_generic(True)
    want = {"default": "k_semi", "fused": "k_fused_td", "fused_prespec": "k_fused", "frame_lds": "k_hop_td", "frame_lds_prespec": "k_hop",
            "generic": "k_wave_iter", "generic_workgroup": "k_iter_pair", "float64": "k_wave_iter", "float64_workgroup": "k_iter_pair"}[path]
    p.keep_state(path.endswith("_prespec"))
    p.gla_init(init, None, alpha)
    assert p.launch_geometry["kernel"] == want, p.launch_geometry
    done, _ = p.run(100, 10, 0.0, "sc")
    y = N(p.wave())
    if path.startswith("float64"):
        assert rel_l2(y, ref64) < 1e-9, rel_l2(y, ref64)
    else:
        assert rel_l2(y, ref) < gate, (path, rel_l2(y, ref), rel_l2(y, ref64), noise)


@pytest.mark.parametrize("kernel", ["k_fused4_td", "k_fused4", "k_fused", "k_hop_td", "k_hop", "k_iter_pair", "k_wave_iter"])
@pytest.mark.parametrize("alpha", [0.0, 0.3, 0.99])
def test_gla_wellconditioned_hop_quarter_100_iterations(alpha, kernel, monkeypatch):
    """g15 (g14's construction at n_fft 1024 / hop 256, the shape class of the headline): 100 iterations of the reference in
    float32 against every kernel that serves the shape, STRICT gate min(1e-4, 6 x the reference's float32-vs-float64 noise).
    `k_fused4_td` (the default) carries the momentum as the signal z_t = x_t - lr z_{t-1} instead of pre_spec: with alpha 0.3 it
    adds the starting spectrum's (-lr)^t share for 15 iterations, with 0.99 for 30, with 0 never; `k_fused4` iterates on
    pre_spec itself (keep_state)."""
    from spectrogram_inversion_amd.plan import Plan
    g = load_golden("g15_wellcond_1024")
    ref, ref64 = g[f"wave_a{alpha}"], g[f"wave64_a{alpha}"]
    noise = rel_l2(ref, ref64)
    gate = min(1e-4, max(6 * noise, 3e-6))
    init = T(g["init"])
    hop, w = int(g["hop"]), torch.from_numpy(g["window"])
    monkeypatch.setenv("SPECINV_SMALL_FRAMES", "0")
    if kernel == "k_fused":
        monkeypatch.setenv("SPECINV_FUSED_TEMPLATE", "1")
    if kernel in ("k_hop", "k_hop_td"):
        monkeypatch.setenv("SPECINV_DISABLE_FUSED", "1")
    if kernel == "k_iter_pair":
        monkeypatch.setenv("SPECINV_GENERIC_WAVE", "0")      # (the wave-level kernel serves forced-generic plans since round 6)
    p = Plan(args_helper(init, hop_length=hop, window=w), init.shape[0], init.shape[2], w.dtype, dev())
    if kernel in ("k_iter_pair", "k_wave_iter"):
        p.force_generic(True)
    p.keep_state(kernel not in ("k_fused4_td", "k_hop_td"))
    p.gla_init(init, None, alpha)
    assert p.launch_geometry["kernel"] == kernel, p.launch_geometry
    done, evals = p.run(100, 10, 0.0, "sc")
    y = N(p.wave())
    assert rel_l2(y, ref) < gate, (kernel, rel_l2(y, ref), rel_l2(y, ref64), noise)
    # the evaluated metric (|STFT(x_t)| against the target every 10 iterations) against the float64 oracle's trace
    trace = []
    oracle.griffin_lim(g["init"].astype(np.complex128), max_iter=100, alpha=alpha, tol=0, eva_iter=10, hop_length=hop,
                       window=g["window"].astype(np.float64), trace=trace)
    got = sc_linear(np.array([m for _, m, _ in evals]))
    want = sc_linear(np.array([m for _, m, _ in trace]))
    assert np.abs(got - want).max() < 1e-5, np.abs(got - want).max()


@pytest.mark.parametrize("kernel", ["k_fused4_td", "k_fused4", "k_hop_td", "k_hop"])
@pytest.mark.parametrize("alpha", [0.0, 0.3, 0.99])
def test_gla_wellconditioned_2048_100_iterations(alpha, kernel, monkeypatch):
    """g16a: g14's construction at n_fft 2048 / hop 512 - the HEADLINE instantiation (`k_fused4_td<16>`, what BASELINE C2 runs) and
    the other kernels that serve the shape, against 100 iterations of the unmodified reference (torch_specinv/methods.py:237-250) in
    float32, STRICT gate min(1e-4, 6 x the reference's own float32-vs-float64 noise).  run(100, 10) takes the signal-form kernel
    through all four variants: with alpha 0.3 the c0 term is added for 15 iterations (early; early + evaluating at iteration 10),
    with 0.99 for 30, with 0 never; from then on the late launch - 77 of C2's 100 - and late + evaluating."""
    from spectrogram_inversion_amd.plan import Plan
    g = load_golden("g16a_wellcond_2048")
    ref, ref64 = g[f"wave_a{alpha}"], g[f"wave64_a{alpha}"]
    noise = rel_l2(ref, ref64)
    gate = min(1e-4, max(6 * noise, 3e-6))
    init = T(g["init"])
    hop, w = int(g["hop"]), torch.from_numpy(g["window"])
    assert (init.shape[1], hop) == (1025, 512)
    monkeypatch.setenv("SPECINV_SMALL_FRAMES", "0")
    if kernel in ("k_hop", "k_hop_td"):
        monkeypatch.setenv("SPECINV_DISABLE_FUSED", "1")
    p = Plan(args_helper(init, hop_length=hop, window=w), init.shape[0], init.shape[2], w.dtype, dev())
    p.keep_state(kernel not in ("k_fused4_td", "k_hop_td"))
    p.gla_init(init, None, alpha)
    assert p.launch_geometry["kernel"] == kernel, p.launch_geometry
    done, evals = p.run(100, 10, 0.0, "sc")
    assert done == 100 and len(evals) == 10
    y = N(p.wave())
    assert rel_l2(y, ref) < gate, (kernel, rel_l2(y, ref), rel_l2(y, ref64), noise)
    trace = []
    oracle.griffin_lim(g["init"].astype(np.complex128), max_iter=100, alpha=alpha, tol=0, eva_iter=10, hop_length=hop,
                       window=g["window"].astype(np.float64), trace=trace)
    got = sc_linear(np.array([m for _, m, _ in evals]))
    want = sc_linear(np.array([m for _, m, _ in trace]))
    assert np.abs(got - want).max() < 1e-5, np.abs(got - want).max()


def test_gla_trace_and_spectral_convergence():
    g = load_golden("g2_gla")
    kw = dict(hop_length=int(g["hop"]), window=torch.from_numpy(g["window"]))
    for alpha in (0.0, 0.3, 0.99):
        spec = T(g["init"])
        a = args_helper(spec, **kw)
        plan = get_plan(a, spec.shape[0], spec.shape[2], torch.float32, dev())
        plan.gla_init(spec, None, alpha)
        done, evals = plan.run(100, 10, 0.0, "sc")
        assert done == 100 and len(evals) == 10
        tr = g[f"trace_a{alpha}_it100"]
        got = np.array([[m, l] for _, m, l in evals])
        assert np.abs(sc_linear(got[:, 0]) - sc_linear(tr[:, 0])).max() < 1e-5      # |dSC_lin| <= 1e-5
        np.testing.assert_allclose(got[:, 1], tr[:, 1], rtol=2e-4)


@pytest.mark.parametrize("metric", ["snr", "ser"])
def test_gla_from_magnitude(metric):
    g = load_golden("g2_gla")
    kw = dict(hop_length=int(g["hop"]), window=torch.from_numpy(g["window"]))
    y = N(si.griffin_lim(T(g["mag"]), max_iter=20, alpha=0.3, tol=0, eva_iter=5, metric=metric, verbose=False, **kw))
    assert rel_l2(y, g["wave_mag_" + metric]) < 1e-4


def test_gla_early_stop_and_shapes():
    g = load_golden("g2_gla")
    kw = dict(hop_length=int(g["hop"]), window=torch.from_numpy(g["window"]))
    spec = T(g["init"])
    a = args_helper(spec, **kw)
    plan = get_plan(a, spec.shape[0], spec.shape[2], torch.float32, dev())
    plan.gla_init(spec, None, 0.99)
    done, _ = plan.run(2000, 10, 1e-6, "sc")
    assert abs(done - int(g["iters_tol"])) <= 10, (done, int(g["iters_tol"]))
    y2 = N(si.griffin_lim(T(g["mag"][0]), max_iter=3, alpha=0.3, tol=0, verbose=False, **kw))
    assert y2.shape == g["wave_2d"].shape and rel_l2(y2, g["wave_2d"]) < 1e-5
    y3 = N(si.griffin_lim(T(g["mag"][:1]), max_iter=3, alpha=0.3, tol=0, verbose=False, **kw))
    assert y3.shape == g["wave_1ft"].shape and rel_l2(y3, g["wave_1ft"]) < 1e-5


def test_cpu_tensor_roundtrip_device():
    g = load_golden("g2_gla")
    kw = dict(hop_length=int(g["hop"]), window=torch.from_numpy(g["window"]))
    y = si.griffin_lim(torch.from_numpy(g["mag"]), max_iter=2, alpha=0.3, tol=0, verbose=False, **kw)
    assert y.device.type == "cpu" and y.shape == (2, 4992)


# ---- G3: stft-kwarg sweep --------------------------------------------------------------------------------
def _sweep_ids():
    return list(range(len(load_golden("g3_sweep")["meta"])))


@pytest.mark.parametrize("i", _sweep_ids())
def test_kwarg_sweep(i):
    g = load_golden("g3_sweep")
    kw = tkw(sweep_kwargs(g["meta"][i]))
    spec = T(g[f"spec{i}"])
    y = N(si.griffin_lim(spec, max_iter=2, alpha=0.5, verbose=False, **kw))
    assert y.shape == g[f"gla{i}"].shape
    assert finite_close(y, g[f"gla{i}"], 5e-5), g["meta"][i]
    z = N(si.ADMM(spec, max_iter=2, rho=0.5, verbose=False, **kw))
    assert finite_close(z, g[f"admm{i}"], 5e-5), g["meta"][i]


# ---- G4: ADMM ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rho", [0.1, 1.0])
@pytest.mark.parametrize("it", [1, 2, 5])
def test_admm_waveforms(rho, it):
    g = load_golden("g4_admm")
    kw = dict(hop_length=int(g["hop"]), window=torch.from_numpy(g["window"]))
    key = f"r{rho}_it{it}"
    y = N(si.ADMM(T(g["init"]), max_iter=it, rho=rho, tol=0, verbose=False, **kw))
    tol = {1: 5e-6, 2: 5e-5, 5: 5e-4}[it]      # rho=0.1 amplifies rounding ~10x per iteration
    assert rel_l2(y, g["wave_" + key]) < tol


@pytest.mark.parametrize("rho", [0.1, 1.0])
def test_admm_200_sc(rho):
    g = load_golden("g4_admm")
    kw = dict(hop_length=int(g["hop"]), window=torch.from_numpy(g["window"]))
    spec = T(g["init"])
    a = args_helper(spec, **kw)
    plan = get_plan(a, spec.shape[0], spec.shape[2], torch.float32, dev())
    plan.admm_init(spec, None, rho)
    _, evals = plan.run(200, 10, 0.0, "sc")
    tr, tr64 = g[f"trace_r{rho}_it200"], g[f"trace64_r{rho}_it200"]
    got = sc_linear(np.array([m for _, m, _ in evals]))
    spread = np.abs(sc_linear(tr[:, 0]) - sc_linear(tr64[:, 0])).max()
    assert np.abs(got - sc_linear(tr[:, 0])).max() < max(3 * spread, 3e-3 if rho == 0.1 else 1e-5)


# ---- G7 / G8 -----------------------------------------------------------------------------------------------------
def test_metrics():
    g = load_golden("g7_metrics")
    a, b = T(g["a"]), T(g["b"])
    got = np.array([si.sc(a, b).item(), si.snr(a, b).item(), si.ser(a, b).item()])
    np.testing.assert_allclose(got, g["vals64"][:3], rtol=2e-6)
    np.testing.assert_allclose(got, g["vals"][:3], rtol=1e-4, atol=1e-4)


def test_float64():
    g = load_golden("g8_f64")
    kw = dict(hop_length=64, window=torch.from_numpy(g["window"]))
    init = T(g["init"])
    assert np.abs(N(si.phase_init(T(g["mag"]), **kw)) - g["init"]).max() < 1e-9
    f = dict(tol=0, verbose=False)
    assert rel_l2(N(si.griffin_lim(init, max_iter=1, alpha=0.3, **f, **kw)), g["gla1"]) < 1e-11
    assert rel_l2(N(si.griffin_lim(init, max_iter=5, alpha=0.3, **f, **kw)), g["gla5"]) < 1e-10
    assert rel_l2(N(si.ADMM(init, max_iter=1, rho=0.1, **f, **kw)), g["admm1"]) < 1e-11
    assert rel_l2(N(si.ADMM(init, max_iter=5, rho=0.1, **f, **kw)), g["admm5"]) < 1e-9


# ---- seeded oracle comparisons at sizes the oracle handles in seconds ---------------------------------------------------
@pytest.mark.parametrize("n_fft,hop,frames,batch", [(1024, 256, 64, 2), (2048, 512, 48, 3), (512, 128, 33, 1),
                                                   (300, 75, 20, 2)])
def test_gla_vs_oracle(n_fft, hop, frames, batch):
    rng = np.random.default_rng(7 + n_fft)
    mag = rng.random((batch, n_fft // 2 + 1, frames), dtype=np.float32)
    w = hann(n_fft)
    ref = oracle.griffin_lim(mag, max_iter=12, alpha=0.3, tol=0, hop_length=hop, window=w)
    y = N(si.griffin_lim(T(mag), max_iter=12, alpha=0.3, tol=0, verbose=False, hop_length=hop,
                         window=torch.from_numpy(w)))
    assert rel_l2(y, ref) < 1e-4, rel_l2(y, ref)


@pytest.mark.parametrize("n_fft,hop,frames,onesided,dtype", [
    (176, 44, 40, True, np.float32),      # 2^4 * 11: a prime factor above the unrolled butterflies (direct DFT stage)
    (254, 64, 30, True, np.float32),      # 2 * 127
    (1018, 255, 12, True, np.float32),    # 2 * 509, hop not dividing n_fft
    (169, 43, 24, False, np.float32),     # 13^2, odd length: two-sided only
    (97, 25, 30, False, np.float64),      # a prime length: one direct DFT
    (286, 71, 20, True, np.float64),      # 2 * 11 * 13
    (448, 112, 16, True, np.float32),     # 2^6 * 7: the radix-7 butterfly
    (8192, 2048, 6, True, np.float32),    # 128 KiB of LDS per frame pair (above the 48 KiB default limit)
    (4096, 1024, 6, True, np.float64),    # the same in float64
])
def test_large_prime_factors_vs_oracle(n_fft, hop, frames, onesided, dtype):
    """n_fft with prime factors the kernels have no unrolled butterfly for (>= 11): every such stage is a direct DFT with
    one output per thread (`fft_stage_any`); Griffin-Lim, ADMM and the transforms against the oracle."""
    rng = np.random.default_rng(n_fft)
    F = n_fft // 2 + 1 if onesided else n_fft
    mag = rng.random((2, F, frames)).astype(dtype)
    w = hann(n_fft).astype(dtype)
    kw = dict(hop_length=hop, onesided=onesided)
    tol = 2e-4 if dtype == np.float32 else 1e-9
    ref = oracle.griffin_lim(mag, max_iter=6, alpha=0.3, tol=0, window=w, **kw)
    y = N(si.griffin_lim(T(mag), max_iter=6, alpha=0.3, tol=0, verbose=False, window=torch.from_numpy(w), **kw))
    assert y.dtype == dtype and rel_l2(y, ref) < tol, rel_l2(y, ref)
    ref = oracle.admm(mag, max_iter=3, rho=0.5, tol=0, window=w, **kw)
    y = N(si.ADMM(T(mag), max_iter=3, rho=0.5, tol=0, verbose=False, window=torch.from_numpy(w), **kw))
    assert rel_l2(y, ref) < tol, rel_l2(y, ref)
    # RTISI-LA amplifies rounding noise along the signal: the yardstick is what the oracle itself loses
    # between float32 and float64 arithmetic on the same input
    r = dict(look_ahead=1, asymmetric_window=True, max_iter=3, alpha=0.5)
    if 2 * n_fft * (16 if dtype == np.float64 else 8) > 120 * 1024:
        # every look-ahead frame has its own pair of LDS buffers in the persistent RTISI kernel: refused, not rerouted
        with pytest.raises(NotImplementedError, match="LDS"):
            si.RTISI_LA(T(mag), verbose=False, window=torch.from_numpy(w), **r, **kw)
        return
    ref64 = oracle.rtisi_la(mag.astype(np.float64), window=w.astype(np.float64), **r, **kw)
    ref = oracle.rtisi_la(mag, window=w, **r, **kw)
    y = N(si.RTISI_LA(T(mag), verbose=False, window=torch.from_numpy(w), **r, **kw))
    e, e0 = segment_errors(y, ref64, 4 * hop), segment_errors(ref, ref64, 4 * hop)
    assert np.median(e) < max(10 * tol, 5 * np.median(e0)), (np.median(e), np.median(e0))      # typical stretch
    assert rel_l2(y, ref64) < max(10 * tol, 20 * rel_l2(ref, ref64)), (rel_l2(y, ref64), rel_l2(ref, ref64))   # no blow-up


@pytest.mark.parametrize("n_fft,dtype,onesided", [
    (2048, np.float64, True), (2048, np.float64, False), (1024, np.float64, True), (2048, np.float32, False),   # k_iter_pair_dr
    (1024, np.float32, True), (1000, np.float64, True),                                                          # Stockham + LDS twiddles
    (512, np.float64, False), (4096, np.float32, True),                                                          # Stockham as before
    (512, np.float64, True), (256, np.float64, True), (256, np.float32, True), (128, np.float32, True),          # k_wave_iter (round 6)
])
def test_coverage_iteration_kernels_agree_and_match_the_oracle(monkeypatch, n_fft, dtype, onesided):
    """Round 5's forms of the coverage path's iteration kernel - the digit-reversed in-place transform with an octant twiddle table
    in LDS (float64 1024 / 2048, float32 2048), the Stockham kernels with the twiddle table copied to LDS (four-wave workgroups) -
    against the plain Stockham kernel (SPECINV_GENERIC_DR=0, SPECINV_GENERIC_TWLDS=0: same butterflies, another order / another
    table: equal to rounding) and against the oracle, Griffin-Lim and ADMM, odd frame counts included (a last pair of one frame)."""
    rng = np.random.default_rng(n_fft + int(onesided))
    hop, frames = n_fft // 4, 7
    F = n_fft // 2 + 1 if onesided else n_fft
    mag = (rng.random((2, F, frames)) + 0.05).astype(dtype)
    w = hann(n_fft, dtype)
    kw = dict(hop_length=hop, onesided=onesided)
    init = oracle.phase_init(mag, window=w, **kw)
    tol = 2e-5 if dtype == np.float32 else 1e-10
    for method, run, okw in (("griffin_lim", si.griffin_lim, dict(max_iter=4, alpha=0.5)), ("admm", si.ADMM, dict(max_iter=3, rho=1.0))):
        ref = getattr(oracle, method)(init, tol=0, window=w, **okw, **kw)
        ref64 = getattr(oracle, method)(init.astype(np.complex128), tol=0, window=hann(n_fft, np.float64), **okw, **kw)
        out = {}
        # (round 6: one-sided float64 at n_fft 128 ... 2048 and float32 at 128 / 256 run the wave-level kernel by default - arm "new";
        # "workgroup" is what ran there before)
        for arm, env in (("new", {}), ("workgroup", {"SPECINV_GENERIC_WAVE": "0"}),
                         ("plain", {"SPECINV_GENERIC_DR": "0", "SPECINV_GENERIC_TWLDS": "0", "SPECINV_GENERIC_WAVE": "0"})):
            for k_ in ("SPECINV_GENERIC_DR", "SPECINV_GENERIC_TWLDS", "SPECINV_GENERIC_WAVE"):
                monkeypatch.delenv(k_, raising=False)
            for k_, v_ in env.items():
                monkeypatch.setenv(k_, v_)
            from spectrogram_inversion_amd.plan import clear_plan_cache
            clear_plan_cache()                                   # (the plan picks its kernels when it is created)
            a = args_helper(T(init), window=torch.from_numpy(w), **kw)
            plan = get_plan(a, 2, frames, torch.from_numpy(w).dtype, dev())
            plan.force_generic(True)
            (plan.gla_init if method == "griffin_lim" else plan.admm_init)(T(init), None, okw.get("alpha", okw.get("rho")))
            plan.iterate(okw["max_iter"])
            out[arm] = N(plan.wave())
        clear_plan_cache()
        e, e0 = rel_l2(out["new"], ref64), rel_l2(ref, ref64)
        assert e < max(3 * e0, 5 * tol), (method, e, e0)
        assert rel_l2(out["new"], out["plain"]) < max(3 * e0, 5 * tol), (method, rel_l2(out["new"], out["plain"]))
        assert rel_l2(out["workgroup"], out["plain"]) < max(3 * e0, 5 * tol), (method, rel_l2(out["workgroup"], out["plain"]))


@pytest.mark.parametrize("n_fft,hop,frames,kw", [
    (512, 128, 9, dict()),
    (512, 100, 40, dict(win_length=300)),                          # the reference's own two-sided shape (test/test_griffin.py:24-32)
    (1024, 256, 7, dict(pad_mode="constant")),
    (1024, 300, 6, dict(pad_mode="replicate")),
    (2048, 512, 6, dict(normalized=True)),
    (4096, 1024, 5, dict()),
])
def test_two_sided_float32_on_the_frame_kernel(monkeypatch, n_fft, hop, frames, kw):
    """onesided=False in float32 runs on the wave-level frame kernel since round 5 (`k_semi2`: the per-bin update a second time on
    the conjugate spectrum with the mirror bins' own target and state, the two results averaged to the Hermitian part that
    ifft(.).real sees - methods.py:142-146, :243-247, :467-475): Griffin-Lim and ADMM from a complex start whose two halves are
    NOT mirror images (random phases per bin, as phase_init gives), a target whose halves differ - against the coverage kernels
    (SPECINV_DISABLE_TWOSIDED=1: the same arithmetic per bin, another transform) and the float64 oracle; the evaluation's sums over
    all N bins and the state read back in the user layout as well."""
    from spectrogram_inversion_amd.plan import clear_plan_cache
    rng = np.random.default_rng(n_fft + hop)
    mag = (rng.random((2, n_fft, frames)) + 0.05).astype(np.float32)
    wl = kw.get("win_length", n_fft)
    w = hann(wl, np.float32)
    okw = dict(hop_length=hop, onesided=False, **kw)
    init = (mag * np.exp(1j * rng.uniform(-np.pi, np.pi, mag.shape))).astype(np.complex64)
    for method, arg, iters in (("griffin_lim", 0.5, 5), ("admm", 1.0, 4)):
        mk = dict(max_iter=iters, alpha=arg) if method == "griffin_lim" else dict(max_iter=iters, rho=arg)
        ref64 = getattr(oracle, method)(init.astype(np.complex128), tol=0, window=hann(wl, np.float64), **mk, **okw)
        ref32 = getattr(oracle, method)(init, tol=0, window=w, **mk, **okw)
        out, sums, state = {}, {}, {}
        for arm in ("frame", "chunked", "coverage"):               # k_semi2 + gather overlap-add / k_hop2 (overlap-add in LDS) / generic
            monkeypatch.setenv("SPECINV_DISABLE_TWOSIDED", "1" if arm == "coverage" else "0")
            monkeypatch.delenv("SPECINV_SMALL_FRAMES", raising=False)
            if arm == "chunked":
                if n_fft > 2048:
                    continue                                       # (k_hop2 stops at n_fft 2048: ring + scratch in LDS)
                monkeypatch.setenv("SPECINV_SMALL_FRAMES", "0")    # (pins the chunked kernel whatever the number of frames)
            clear_plan_cache()
            a = args_helper(T(init), window=torch.from_numpy(w), **okw)
            plan = get_plan(a, 2, frames, torch.float32, dev())
            assert plan.path_code == {"frame": 2, "chunked": 3, "coverage": 0}[arm], (arm, plan.path_code)
            (plan.gla_init if method == "griffin_lim" else plan.admm_init)(T(init), T(mag), arg)
            sums[arm] = plan.iterate(iters, eval_last=True)
            out[arm] = N(plan.wave())
            state[arm] = N(plan.state_spec(0 if method == "griffin_lim" else 2))
        clear_plan_cache()
        e, e0 = rel_l2(out["frame"], ref64), rel_l2(ref32, ref64)
        assert e < max(3 * e0, 1e-4), (method, e, e0)
        assert rel_l2(out["frame"], out["coverage"]) < max(3 * e0, 1e-4), (method, rel_l2(out["frame"], out["coverage"]))
        np.testing.assert_allclose(sums["frame"], sums["coverage"], rtol=1e-4)
        assert state["frame"].shape == (2, n_fft, frames)
        assert rel_l2(state["frame"], state["coverage"]) < max(10 * e0, 1e-4), rel_l2(state["frame"], state["coverage"])
        if "chunked" in out:
            # (the same frame body; the overlap-add divides by the envelope's reciprocal table here, by the envelope there)
            assert rel_l2(out["chunked"], out["frame"]) < max(3 * e0, 1e-4), (method, rel_l2(out["chunked"], out["frame"]))
            np.testing.assert_allclose(sums["chunked"], sums["frame"], rtol=1e-4)
            assert rel_l2(state["chunked"], state["frame"]) < max(10 * e0, 1e-4)


@pytest.mark.parametrize("method", ["admm", "griffin_lim"])
def test_keep_state_toggled_in_the_middle_of_a_two_sided_run(method):
    """On a two-sided float32 plan `keep_state` selects the KERNELS (frame kernel carrying Y alone / coverage kernels with X and U):
    the choice is latched by *_init, so a toggle between init and iterate - either way - neither switches to buffers the other
    path never reserved nor changes the iterates; it takes effect at the next init (round-5 advice)."""
    n_fft, hop, frames = 512, 128, 12
    rng = np.random.default_rng(77)
    mag = (rng.random((2, n_fft, frames)) + 0.05).astype(np.float32)
    init = (mag * np.exp(1j * rng.uniform(-np.pi, np.pi, mag.shape))).astype(np.complex64)
    a = args_helper(T(init), hop_length=hop, onesided=False, window=torch.from_numpy(hann(n_fft)))
    out = {}
    for first in (False, True):
        for toggle in (False, True):
            plan = Plan(a, 2, frames, torch.float32, dev())
            plan.keep_state(first)
            (plan.admm_init if method == "admm" else plan.gla_init)(T(init), T(mag), 0.7)
            code = plan.path_code
            assert code == (0 if first else 2)
            if toggle:
                plan.keep_state(not first)
            assert plan.path_code == code                          # the running method keeps its kernels
            plan.iterate(2)
            plan.iterate(1, eval_last=True)
            out[first, toggle] = (N(plan.wave()), N(plan.state_spec(2 if method == "admm" else 0)))
            if first and method == "admm":
                assert np.isfinite(N(plan.state_spec(0))).all() and np.isfinite(N(plan.state_spec(1))).all()
            # ... and the next init reads the flag as it stands now
            (plan.admm_init if method == "admm" else plan.gla_init)(T(init), T(mag), 0.7)
            assert plan.path_code == (0 if (first != toggle) else 2)
    for first in (False, True):
        assert np.array_equal(out[first, True][0], out[first, False][0])
        assert np.array_equal(out[first, True][1], out[first, False][1])
    assert rel_l2(out[True, False][0], out[False, False][0]) < 1e-4


@pytest.mark.parametrize("kernels", ["workgroup", "default"])
@pytest.mark.parametrize("n_fft,dtype,tol", [(16384, np.float32, 2e-5), (8192, np.float64, 1e-10)])
def test_transforms_beyond_two_lds_buffers(monkeypatch, n_fft, dtype, tol, kernels):
    """The reference derives n_fft from the spectrogram with no bound (torch_specinv/methods.py:65-68).  The generic kernels keep a
    frame pair's transform in LDS: with the in-place form (one buffer of n_fft complex points, kernels_generic.h) 16384 points in
    float32 and 8192 in float64 - refused until round 3 - run: Griffin-Lim and ADMM against the oracle.  (Since round 6 these two
    sizes run on teams of eight waves of `k_wave_iter` by default: both.)"""
    from spectrogram_inversion_amd.plan import clear_plan_cache
    if kernels == "workgroup":
        monkeypatch.setenv("SPECINV_GENERIC_WAVE", "0")
    clear_plan_cache()
    rng = np.random.default_rng(n_fft)
    hop, frames = n_fft // 4, 6
    mag = (rng.random((2, n_fft // 2 + 1, frames)) + 0.05).astype(dtype)
    w = hann(n_fft, dtype)
    init = oracle.phase_init(mag, hop_length=hop, window=w)        # (a complex start: the transforms are what is under test)
    # random magnitudes are inconsistent: a few of the 10^5 bins pass close to zero and amplify rounding (the float64 run's 1e-10
    # is 10^6 ulp).  The float32 run is therefore held to the float32 ORACLE's own distance from the float64 oracle.
    w64 = hann(n_fft, np.float64)
    kw64 = dict(tol=0, hop_length=hop, window=w64)
    for method, run, okw in (("griffin_lim", si.griffin_lim, dict(max_iter=3, alpha=0.5)), ("admm", si.ADMM, dict(max_iter=2, rho=1.0))):
        ref = getattr(oracle, method)(init, tol=0, hop_length=hop, window=w, **okw)
        ref64 = getattr(oracle, method)(init.astype(np.complex128), **okw, **kw64)
        y = N(run(T(init), tol=0, verbose=False, hop_length=hop, window=torch.from_numpy(w), **okw))
        assert y.shape == ref.shape
        e, e0 = rel_l2(y, ref64), rel_l2(ref, ref64)
        assert e < max(3 * e0, 5 * tol), (method, e, e0)


@pytest.mark.parametrize("n_fft,dtype,onesided,tol", [
    (32768, np.float32, True, 2e-5),      # 4 rows of 8192 points
    (65536, np.float32, True, 2e-5),      # 8 rows: the largest float32 frame
    (32768, np.float32, False, 2e-5),     # two-sided
    (16384, np.float64, True, 1e-10),     # 4 rows of 4096
    (32768, np.float64, True, 1e-10),     # 8 rows: the largest float64 frame
    (3 * 16384, np.float32, True, 2e-5),  # 8 rows of 6144 = 2^11 * 3 points
])
def test_frames_beyond_the_lds_take_four_steps(n_fft, dtype, onesided, tol):
    """The reference derives n_fft from the spectrogram with no bound (torch_specinv/methods.py:65-68).  A frame above 16384 points
    in float32 / 8192 in float64 - refused until round 5 - is transformed as 2, 4 or 8 rows through device memory (kernels_big.h):
    phase_init, the transforms, Griffin-Lim and ADMM against the oracle (the float32 runs held to the float32 oracle's own distance
    from the float64 one, as in test_transforms_beyond_two_lds_buffers)."""
    rng = np.random.default_rng(n_fft)
    hop, frames = n_fft // 4, 5
    F = n_fft // 2 + 1 if onesided else n_fft
    mag = (rng.random((2, F, frames)) + 0.05).astype(dtype)
    w = hann(n_fft, dtype)
    w64 = hann(n_fft, np.float64)
    kw = dict(hop_length=hop, onesided=onesided)
    init = oracle.phase_init(mag, window=w, **kw)
    got = N(si.phase_init(T(mag), window=torch.from_numpy(w), **kw))
    assert rel_l2(got, init) < 1e-6
    # the building blocks: x -> STFT -> ISTFT is the identity up to rounding
    x = rng.standard_normal((2, (frames - 1) * hop)).astype(dtype)
    a = args_helper(T(mag), window=torch.from_numpy(w), **kw)
    plan = get_plan(a, 2, frames, torch.from_numpy(w).dtype, dev())
    sp = plan.stft(T(x))
    ref_sp = oracle.stft(x.astype(np.float64), oracle.args_helper(F, np.float64, window=w64, **kw))
    assert rel_l2(N(sp), ref_sp) < 20 * tol
    assert rel_l2(N(plan.istft(sp)), x) < 20 * tol
    for method, run, okw in (("griffin_lim", si.griffin_lim, dict(max_iter=3, alpha=0.5)), ("admm", si.ADMM, dict(max_iter=2, rho=1.0))):
        ref = getattr(oracle, method)(init, tol=0, window=w, **okw, **kw)
        ref64 = getattr(oracle, method)(init.astype(np.complex128), tol=0, window=w64, **okw, **kw)
        y = N(run(T(init), tol=0, verbose=False, window=torch.from_numpy(w), **okw, **kw))
        assert y.shape == ref.shape and y.dtype == dtype
        e, e0 = rel_l2(y, ref64), rel_l2(ref, ref64)
        assert e < max(3 * e0, 5 * tol), (method, e, e0)
    # the evaluation of the training loop (methods.py:180-182) on the four-step path: the metric after a short run
    plan.gla_init(T(init), None, 0.3)
    done, evals = plan.run(4, 2, 0.0, "sc")
    tr = []
    oracle.griffin_lim(init, max_iter=4, alpha=0.3, tol=0, eva_iter=2, window=w, trace=tr, **kw)
    assert done == 4 and len(evals) == 2 and len(tr) == 2
    assert abs(10 ** (evals[-1][1] / 20) - 10 ** (tr[-1][1] / 20)) < 1e-4, (evals, tr)


def test_transform_too_large_even_for_four_steps_is_refused():
    """Eight rows of 8192 (float32) / 4096 (float64) points are the largest frame: beyond, SPECINV_EUNSUPPORTED at plan creation, not
    some slower path; so is RTISI-LA where its frame buffers (two per look-ahead frame, the two-buffer transform) do not fit."""
    for n_fft, dtype in ((131072, torch.float32), (65536, torch.float64)):
        mag = torch.rand(1, n_fft // 2 + 1, 4, dtype=dtype, device=dev())
        with pytest.raises(NotImplementedError, match="rows"):
            si.griffin_lim(mag, max_iter=1, verbose=False, hop_length=n_fft // 4)
    mag = torch.rand(1, 8193, 6, dtype=torch.float32, device=dev())
    with pytest.raises(NotImplementedError, match="LDS"):
        si.RTISI_LA(mag, look_ahead=1, max_iter=2, verbose=False, hop_length=4096)


def test_state_spec_parity():
    rng = np.random.default_rng(11)
    mag = rng.random((2, 129, 30), dtype=np.float32)
    w = hann(256)
    init = oracle.phase_init(mag, hop_length=64, window=w)
    _, st = oracle.griffin_lim(init, max_iter=3, alpha=0.5, tol=0, hop_length=64, window=w, return_state=True)
    spec = T(init)
    plan = get_plan(args_helper(spec, hop_length=64, window=torch.from_numpy(w)), 2, 30, torch.float32, dev())
    plan.gla_init(spec, None, 0.5)
    plan.iterate(3)
    assert rel_l2(N(plan.state_spec(0)), st["pre_spec"]) < 1e-5
