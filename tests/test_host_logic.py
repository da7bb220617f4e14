"""Host-side logic that needs no GPU: argument normalisation, shape rules, error behaviour, metric
formulas, the L-BFGS control flow (with a torch-CPU vector backend supplied by the test), mel filterbank."""
import math
import os

import numpy as np
import pytest
import torch

import oracle
import spectrogram_inversion_amd as si
from _util import ROOT, hann, load_golden, sweep_kwargs
from spectrogram_inversion_amd.lbfgs import LBFGS
from spectrogram_inversion_amd.metrics import _from_sums
from spectrogram_inversion_amd.plan import args_helper, require_gpu


def _tk(kw):
    out = dict(kw)
    if out.get("window") is not None:
        out["window"] = torch.from_numpy(np.asarray(out["window"]))
    return out


def test_args_helper_matches_oracle_over_sweep():
    g = load_golden("g3_sweep")
    for row in g["meta"]:
        kw = sweep_kwargs(row)
        n_freq = 257 if kw["onesided"] else 512
        a = args_helper(torch.empty(1, n_freq, 3), **_tk(kw))
        o = oracle.args_helper(n_freq, np.float32, **kw)
        assert (a.n_fft, a.win_length, a.hop_length, a.center, a.pad_mode, a.normalized, a.onesided) == \
               (o.n_fft, o.win_length, o.hop_length, o.center, o.pad_mode, o.normalized, o.onesided)
        np.testing.assert_array_equal(a.window.numpy(), o.window)
        for frames in (1, 7, 40):
            assert a.signal_length(frames) == oracle.signal_length(frames, o)
        assert a.frame_count(4410) == oracle.frame_count(4410, o)


def test_args_helper_rules():
    spec = torch.empty(2, 257, 9)
    a = args_helper(spec)                                   # defaults: methods.py:34-41,70-77
    assert (a.n_fft, a.hop_length, a.win_length, a.onesided, a.center, a.pad_mode) == (512, 128, 512, True, True, "reflect")
    assert torch.equal(a.window, torch.ones(512))
    a = args_helper(spec, win_length=300, maxiter=7, foo="ignored")    # unknown kwargs dropped, :42-46
    assert a.window.numel() == 512 and a.window[:106].sum() == 0 and a.window[106:406].sum() == 300
    a = args_helper(torch.empty(2, 512, 9), onesided=False)
    assert a.n_fft == 512 and a.n_freq == 512
    a = args_helper(torch.empty(2, 257, 9, dtype=torch.complex64))
    assert a.window.dtype == torch.float32                  # complex -> real dtype, :50-57
    with pytest.raises(AssertionError):
        args_helper(spec, win_length=600)                   # n_fft >= win_length, :79


def test_argument_errors_precede_device_use():
    mag = torch.rand(2, 65, 8)
    with pytest.raises(AssertionError):
        si.griffin_lim(mag, alpha=-1)                       # methods.py:223
    with pytest.raises(AssertionError):
        si.griffin_lim(torch.rand(65))                      # rank check, :101
    with pytest.raises(AssertionError):
        si.ADMM(mag, metric="psnr")                         # :443
    with pytest.raises(AssertionError):
        si.ADMM(mag, eva_iter=0)                            # :440
    with pytest.raises(AssertionError):
        si.RTISI_LA(mag.to(torch.complex64))                # :297
    with pytest.raises(AssertionError):
        si.RTISI_LA(mag, max_iter=0)                        # :295
    with pytest.raises(AssertionError):
        si.phase_init(mag.to(torch.complex64))              # :586


def test_complex_window_and_unknown_kwargs_follow_the_reference():
    """What the unmodified reference does (run in the build container, torch 2.10): a complex window makes `onesided`
    default to False (methods.py:59-63) and is accepted by `phase_init`, which never reads it; griffin_lim / ADMM /
    RTISI_LA raise RuntimeError on it (conv_transpose1d of real frames with the complex diag(window) weight, :127).
    RTISI_LA forwards the raw kwargs to torch.stft on the non-asymmetric path (:308-310,385): an unknown name is a
    TypeError there, while the asymmetric path and griffin_lim / ADMM silently drop it (:42-46)."""
    w = torch.hann_window(256).to(torch.complex64) * torch.exp(1j * torch.linspace(0, 1, 256))
    a = args_helper(torch.empty(2, 256, 9), window=w)
    assert a.complex_window and not a.onesided and a.n_fft == 256 and a.window.dtype == torch.float32
    mag = torch.rand(2, 256, 9)
    for fn in (si.griffin_lim, si.ADMM, si.RTISI_LA):
        with pytest.raises(RuntimeError, match="complex windows"):
            fn(mag, window=w, verbose=False)
    with pytest.raises(TypeError, match="unexpected keyword argument 'maxiter'"):
        si.RTISI_LA(torch.rand(2, 129, 9), maxiter=3, verbose=False)
    with pytest.raises(TypeError, match="unexpected keyword argument 'n_fft'"):
        si.RTISI_LA(torch.rand(2, 129, 9), n_fft=256, verbose=False)
    if not torch.cuda.is_available():
        # the asymmetric path drops the unknown name and gets as far as asking for the device
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            si.RTISI_LA(torch.rand(2, 129, 9), asymmetric_window=True, maxiter=3, verbose=False)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_cpu_fallback():
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        require_gpu(torch.device("cpu"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        si.griffin_lim(torch.rand(2, 65, 8), max_iter=1, verbose=False)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        si.sc(torch.rand(4), torch.rand(4))


def test_product_never_imports_the_oracle():
    import os
    import re
    root = os.path.dirname(os.path.abspath(si.__file__))
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f


def test_metric_formulas():
    rng = np.random.default_rng(0)
    a, b = rng.random(1000), rng.random(1000)
    s = [float(((a - b) ** 2).sum()), float((a ** 2).sum()), float((b ** 2).sum()), 1000.0]
    assert math.isclose(_from_sums("SC", s), oracle.sc(a, b), rel_tol=1e-12)
    assert math.isclose(_from_sums("SNR", s), oracle.snr(a, b), rel_tol=1e-12)
    assert math.isclose(_from_sums("SER", s), oracle.ser(a, b), rel_tol=1e-12)


class TorchVecOps:
    """Vector backend used ONLY here, to exercise the optimiser's host control flow without a GPU."""

    def dot(self, a, b):
        return float(torch.dot(a.reshape(-1), b.reshape(-1)))

    def axpy(self, alpha, x, y):
        y.add_(x, alpha=alpha)

    def scaled(self, alpha, x):
        return x * alpha

    def absmax_abssum(self, x):
        return float(x.abs().max()), float(x.abs().sum())


class GramTorchVecOps(TorchVecOps):
    """The same plus the two many-vector passes: switches the optimiser to the recursion on Gram matrices."""

    def multi_dot(self, g, vecs):
        return [float(torch.dot(g.reshape(-1).double(), v.reshape(-1).double())) for v in vecs]

    def lincomb(self, vecs, coefs):
        out = torch.zeros_like(vecs[0], dtype=torch.float64)
        for c, v in zip(coefs, vecs):
            out += float(c) * v.double()
        return out.to(vecs[0].dtype)


class PackedTorchVecOps(GramTorchVecOps):
    """... plus the packed interface of the HIP backend (results parked on a "board", one read per iteration): drives
    `LBFGS._step_packed`, the loop the GPU runs, on CPU tensors.  `reads` counts the synchronisation points."""
    packed = True

    def __init__(self):
        self.reads = 0

    def board(self, n):
        return torch.zeros(n, dtype=torch.float64)

    def eval_into(self, fg, x, board, slot):
        loss, g = fg(x)
        board[slot] = loss
        return g

    def stats_into(self, g, d, board, slot):
        g64, d64 = g.double().reshape(-1), d.double().reshape(-1)
        board[slot:slot + 4] = torch.stack([torch.dot(g64, d64), g64.abs().sum(), g64.abs().max(), d64.abs().max()])

    def pair_into(self, g, g_prev, d, t, board, slot):
        y, s = g - g_prev, d * t
        g64 = g.double().reshape(-1)
        board[slot:slot + 4] = torch.stack([torch.dot(y.double().reshape(-1), s.double().reshape(-1)),
                                            torch.dot(y.double().reshape(-1), y.double().reshape(-1)), torch.dot(g64, g64),
                                            torch.dot(g64, g_prev.double().reshape(-1))])
        return y, s

    def multi_dot_into(self, g, vecs, board, slot):
        board[slot:slot + len(vecs)] = torch.tensor(self.multi_dot(g, vecs), dtype=torch.float64)

    def read(self, board, n):
        self.reads += 1
        return board[:n].tolist()


@pytest.mark.parametrize("tag,kw", [("fixed", dict(max_iter=30, lr=1e-3, history_size=4)),
                                    ("fixed_h2", dict(max_iter=12, lr=1e-3, history_size=2))])
def test_lbfgs_packed_loop_retraces_torch(tag, kw):
    """The one-synchronisation-per-iteration loop (`_step_packed`) against torch.optim.LBFGS itself: same losses, same
    iterate, and exactly one read per objective evaluation."""
    g = load_golden("g9_lbfgs_rosen")
    x = torch.from_numpy(g["x0"].copy())
    xt = torch.nn.Parameter(torch.from_numpy(g["x0"].copy()))
    losses, ref = [], []

    def rosen(v):
        a, b = v[1:] - v[:-1] ** 2, 1.0 - v[:-1]
        return (100.0 * a * a + b * b).sum()

    def fg(v):
        p = v.detach().clone().requires_grad_(True)
        f = rosen(p)
        (gr,) = torch.autograd.grad(f, p)
        losses.append(float(f))
        return float(f), gr

    topt = torch.optim.LBFGS([xt], **kw)

    def closure():
        topt.zero_grad()
        f = rosen(xt)
        f.backward()
        ref.append(float(f))
        return f

    ops = PackedTorchVecOps()
    opt = LBFGS(x, vec_ops=ops, **kw)
    for _ in range(3):
        opt.step(fg)
        topt.step(closure)
    assert ops.reads == len(losses) == len(ref)
    np.testing.assert_allclose(losses, ref, rtol=5e-4, atol=1e-7)
    np.testing.assert_allclose(x.numpy(), xt.detach().numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("backend", [TorchVecOps, GramTorchVecOps])
@pytest.mark.parametrize("tag,kw", [("wolfe", dict(max_iter=40, history_size=5, line_search_fn="strong_wolfe")),
                                    ("wolfe_h100", dict(max_iter=25, line_search_fn="strong_wolfe")),
                                    ("fixed", dict(max_iter=30, lr=1e-3, history_size=4))])
def test_lbfgs_control_flow_retraces_torch(tag, kw, backend):
    g = load_golden("g9_lbfgs_rosen")
    x = torch.from_numpy(g["x0"].copy())
    losses = []

    def fg(v):
        a, b = v[1:] - v[:-1] ** 2, 1.0 - v[:-1]
        f = float((100.0 * a * a + b * b).sum())
        gr = torch.zeros_like(v)
        gr[1:] += 200.0 * a
        gr[:-1] += -400.0 * a * v[:-1] - 2.0 * b
        losses.append(f)
        return f, gr

    opt = LBFGS(x, vec_ops=backend(), **kw)
    assert opt.gram == (backend is GramTorchVecOps)
    for _ in range(2):
        opt.step(fg)
    ref = g[f"losses_{tag}"]
    assert len(losses) == len(ref)
    np.testing.assert_allclose(losses, ref, rtol=5e-4, atol=1e-7)
    np.testing.assert_allclose(x.numpy(), g[f"x_{tag}"], rtol=1e-4, atol=1e-6)


def test_lbfgs_option_validation():
    x = torch.zeros(3)
    with pytest.raises(ValueError):
        LBFGS(x, lr=-1.0, vec_ops=TorchVecOps())
    with pytest.raises(RuntimeError):
        LBFGS(x, line_search_fn="armijo", vec_ops=TorchVecOps())
    assert LBFGS(x, max_iter=20, vec_ops=TorchVecOps()).max_eval == 25       # max_iter * 5 // 4


def test_mel_filterbank():
    fb = si.mel_filterbank(22050, 2048, 80)
    assert fb.shape == (80, 1025) and fb.dtype == np.float32 and (fb >= 0).all()
    assert 0.02 < (fb > 0).mean() < 0.03                     # ~2.4 % non-zeros
    peaks = fb.argmax(1)
    assert (np.diff(peaks) > 0).all()                        # centre frequencies increase
    # Slaney area normalisation: every filter integrates to ~1 over frequency (bin width sr/n_fft)
    area = fb.sum(1) * (22050 / 2048)
    assert np.allclose(area[5:], 1.0, atol=0.08)
    g = load_golden("g6_lbfgs")
    np.testing.assert_array_equal(fb[:, ::64], g["mel_fb_check"])


def test_mel_filterbank_options():
    """HTK scale and un-normalised peaks (the options of the mel function the reference's README calls)."""
    from spectrogram_inversion_amd.mel import hz_to_mel, mel_to_hz
    f = np.array([0.0, 440.0, 1000.0, 4000.0, 11025.0])
    for htk in (False, True):
        np.testing.assert_allclose(mel_to_hz(hz_to_mel(f, htk), htk), f, rtol=1e-12, atol=1e-9)
    assert abs(float(hz_to_mel(1000.0, True)) - 999.9855) < 1e-3                # 2595 log10(1 + f/700)
    assert abs(float(hz_to_mel(1000.0)) - 15.0) < 1e-12                       # Slaney: 1 kHz = mel 15
    peak1 = si.mel_filterbank(16000, 512, 40, norm=None)
    assert peak1.shape == (40, 257) and 0.5 < peak1.max() <= 1.0 + 1e-6
    htk = si.mel_filterbank(16000, 512, 40, htk=True)
    sl = si.mel_filterbank(16000, 512, 40)
    assert (np.diff(htk.argmax(1)) >= 0).all() and not np.allclose(htk, sl)
    area = sl.sum(1) * (16000 / 512)
    assert np.allclose(area[8:], 1.0, atol=0.1)
    lim = si.mel_filterbank(16000, 512, 20, fmin=300.0, fmax=4000.0)
    hz = np.linspace(0, 8000, 257)
    assert lim[:, hz < 290].sum() == 0 and lim[:, hz > 4010].sum() == 0


def test_exact_projection_switch_and_progress_mode(monkeypatch):
    """The module-level switches of round 3 (no GPU involved): `set_exact_projection` overrides SPECINV_EXACT, None follows it;
    RTISI_LA's live progress bar is only taken when somebody can watch it (a terminal) or on request."""
    from spectrogram_inversion_amd import plan as P, methods as M
    monkeypatch.delenv("SPECINV_EXACT", raising=False)
    P.set_exact_projection(None)
    assert P.exact_projection() is True                     # (round 4: the reference's operation order is the default)
    monkeypatch.setenv("SPECINV_EXACT", "0")
    assert P.exact_projection() is False
    P.set_exact_projection(True)
    assert P.exact_projection() is True
    P.set_exact_projection(False)
    monkeypatch.setenv("SPECINV_EXACT", "1")
    assert P.exact_projection() is False
    P.set_exact_projection(None)
    assert P.exact_projection() is True
    monkeypatch.setenv("SPECINV_RTISI_PROGRESS", "blocks")
    assert M._live_progress() is True
    monkeypatch.setenv("SPECINV_RTISI_PROGRESS", "end")
    assert M._live_progress() is False
    monkeypatch.delenv("SPECINV_RTISI_PROGRESS")
    assert M._live_progress() is False                      # (pytest captures stderr: not a terminal)


def test_bench_byte_conventions():
    """bench.py prices `roofline.frac` with SURVEY 8d's bytes of the reference algorithm and reports the bytes the shipped kernels
    have to move beside it: the figures BASELINE.md quotes."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.algorithmic_bytes_per_unit("griffin_lim", 512, 1025, 0.3) == 24596          # C2
    assert bench.algorithmic_bytes_per_unit("griffin_lim", 256, 513, 0.0) == 4100            # C1
    assert bench.algorithmic_bytes_per_unit("ADMM", 256, 513, 0.1) == 20516                  # C4
    assert bench.algorithmic_bytes_per_unit("L_BFGS", 512, 1025, None) == 4416               # C5
    assert bench.restated_bytes_per_unit("griffin_lim", 512, 1025, "k_fused4_td") == 8196
    assert bench.restated_bytes_per_unit("griffin_lim", 512, 1025, "k_fused4") is None
    assert bench.restated_bytes_per_unit("ADMM", 256, 513, "k_fused4") == 12308
    from spectrogram_inversion_amd.build import sources_hash
    assert len(sources_hash()) == 16 and sources_hash() == sources_hash()


def test_bench_spawns_its_own_ranks(monkeypatch):
    """`python3 bench.py --gpus N` with WORLD_SIZE unset starts N ranks under torch.distributed.run as a child process (bench.py:
    spawn_ranks) - the command line the driver itself uses, rendezvous on 127.0.0.1 - and leaves with its exit code; nothing in
    the parent touches the GPU before that."""
    import importlib.util
    import subprocess
    import sys
    spec = importlib.util.spec_from_file_location("bench_spawn", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--workload", "C4"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "3", "--workload", "C4"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
