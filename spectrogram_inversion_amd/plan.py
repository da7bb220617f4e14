"""Python handle on a libspecinv plan: argument normalisation (the reference's
`_args_helper`, torch_specinv/methods.py:21-91) and thin typed wrappers over the
C ABI.  torch is used for device memory and streams only."""
from __future__ import annotations

import ctypes as C
import os
import threading
from collections import OrderedDict
from dataclasses import dataclass

import torch

from . import _lib

_REAL = {torch.complex64: torch.float32, torch.complex128: torch.float64, torch.complex32: torch.float16}
_CPLX = {torch.float32: torch.complex64, torch.float64: torch.complex128}
_RECOGNISED = ("win_length", "window", "hop_length", "center", "pad_mode", "normalized", "onesided",
               "return_complex")


@dataclass
class StftArgs:
    """What `_args_helper` returns (methods.py:85-91), window already centre-padded."""
    n_fft: int
    win_length: int
    hop_length: int
    window: torch.Tensor
    center: bool
    pad_mode: str
    normalized: bool
    onesided: bool
    complex_window: bool = False

    @property
    def n_freq(self):
        return self.n_fft // 2 + 1 if self.onesided else self.n_fft

    @property
    def padding(self):
        return self.n_fft // 2 if self.center else 0

    def signal_length(self, n_frames):
        return (n_frames - 1) * self.hop_length + self.n_fft - 2 * self.padding

    def frame_count(self, length):
        return 1 + (length + 2 * self.padding - self.n_fft) // self.hop_length


def args_helper(spec: torch.Tensor, **stft_kwargs) -> StftArgs:
    """methods.py:21-91.  Unknown kwargs are silently ignored (:42-46)."""
    kw = {k: stft_kwargs[k] for k in _RECOGNISED if k in stft_kwargs}
    win_length = kw.get("win_length", None)
    window = kw.get("window", None)
    hop_length = kw.get("hop_length", None)
    center = kw.get("center", True)
    pad_mode = kw.get("pad_mode", "reflect")
    normalized = kw.get("normalized", False)
    onesided = kw.get("onesided", None)

    dtype = _REAL.get(spec.dtype, spec.dtype)                         # :49-57
    if onesided is None:                                              # :59-63
        onesided = not (window is not None and window.is_complex())
    n_fft = (spec.shape[-2] - 1) * 2 if onesided else spec.shape[-2]  # :65-68
    if not win_length:                                                # :70-71
        win_length = n_fft
    if not hop_length:                                                # :73-74
        hop_length = n_fft // 4
    if window is None:                                                # :76-77
        window = torch.ones(win_length, dtype=dtype)
    assert n_fft >= win_length                                        # :79
    # A complex window only decides `onesided` (:59-63) as far as the reference gets: phase_init never touches the
    # window values (:592-605), and griffin_lim / ADMM / RTISI_LA raise RuntimeError on it (conv_transpose1d of the
    # real frames with the complex diag(window) weight, :95,127; `asym_window1 += window`, :327).  The plan keeps
    # |window| so that phase_init has its shape; the iterative entry points refuse (methods._no_complex_window).
    complex_window = bool(window.is_complex())
    if complex_window:
        window = window.detach().abs()
    window = window.detach().to(device="cpu", dtype=dtype).reshape(-1)
    assert window.numel() == win_length
    if n_fft > win_length:                                            # :80-83
        left = (n_fft - win_length) // 2
        right = (n_fft - win_length + 1) // 2
        window = torch.nn.functional.pad(window, [left, right])
        win_length = n_fft
    return StftArgs(int(n_fft), int(win_length), int(hop_length), window.contiguous(), bool(center),
                    str(pad_mode), bool(normalized), bool(onesided), complex_window)


def require_gpu(device: torch.device | None = None) -> torch.device:
    """The engine only runs on the HIP device; there is no CPU path."""
    if device is not None and device.type == "cuda":
        return device
    if not torch.cuda.is_available():
        raise RuntimeError("spectrogram_inversion_amd needs an MI355X (HIP) device; there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


class Plan:
    """One (device, dtype, batch, frames, stft-args) problem."""

    def __init__(self, args: StftArgs, batch: int, n_frames: int, dtype: torch.dtype, device: torch.device):
        if dtype not in _CPLX:
            raise NotImplementedError(f"dtype {dtype} is not supported on the device path (float32/float64 only)")
        if args.pad_mode not in _lib.PAD_MODES:
            raise AssertionError(f"unknown pad_mode {args.pad_mode!r}")
        self.lib = _lib.load()
        self.args = args
        self.batch, self.n_frames = int(batch), int(n_frames)
        self.dtype, self.cdtype = dtype, _CPLX[dtype]
        self.device = device
        win = args.window.to(dtype=dtype, device="cpu").contiguous()
        cfg = _lib.StftCfg(args.n_fft, args.hop_length, self.n_frames, self.batch, int(args.center),
                           _lib.PAD_MODES[args.pad_mode], int(args.normalized), int(args.onesided),
                           _lib.F32 if dtype == torch.float32 else _lib.F64,
                           device.index if device.index is not None else torch.cuda.current_device(),
                           win.data_ptr())
        handle = C.c_void_p()
        _lib.check(self.lib.specinv_plan_create(C.byref(cfg), C.byref(handle)))
        self._h = handle
        self.n_freq = self.lib.specinv_plan_n_freq(self._h)
        self.length = int(self.lib.specinv_plan_length(self._h))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        destroy = getattr(getattr(self, "lib", None), "specinv_plan_destroy", None)
        if h and destroy is not None:          # (None only while the interpreter is being torn down)
            destroy(h)

    @property
    def device_bytes(self) -> int:
        """Device memory the plan's buffers hold right now (`specinv_plan_device_bytes`)."""
        return int(self.lib.specinv_plan_device_bytes(self._h))

    # -- helpers ------------------------------------------------------------------------
    def _sync_stream(self):
        s = torch.cuda.current_stream(self.device)
        _lib.check(self.lib.specinv_plan_set_stream(self._h, C.c_void_p(s.cuda_stream)))

    def _in(self, t: torch.Tensor, dtype, shape=None):
        # (the optimiser loops call this a few hundred thousand times with tensors that are already in place: skip the three
        # dispatcher round trips then)
        if t.dtype != dtype or t.device != self.device or t.requires_grad or not t.is_contiguous():
            t = t.detach().to(device=self.device, dtype=dtype).contiguous()
        if shape is not None:
            assert tuple(t.shape) == tuple(shape), f"expected shape {tuple(shape)}, got {tuple(t.shape)}"
        return t

    def _spec_shape(self):
        return (self.batch, self.n_freq, self.n_frames)

    @property
    def fast_path(self) -> bool:
        """True when the iteration runs on the wave-level FFT kernels (`path` tells which form)."""
        return self.lib.specinv_plan_fast_path(self._h) > 0

    @property
    def path(self) -> str:
        """"fused" (one launch per iteration), "frame" (wave-level frame kernel + overlap-add) or "generic"."""
        return ("generic", "fused", "frame", "frame")[self.lib.specinv_plan_fast_path(self._h)]

    @property
    def path_code(self) -> int:
        """`specinv_plan_fast_path`: 0 generic, 1 fused, 2 frame kernel + gather overlap-add, 3 frame kernel over chunks
        of frames with the overlap-add in LDS."""
        return self.lib.specinv_plan_fast_path(self._h)

    @property
    def objective_kind(self) -> str:
        """How the last `transform_loss_grad` of this plan ran (`specinv_transform_objective_kind`): "chain" of kernels,
        one launch with the mel contractions on the "matrix" cores, one launch with the filterbank as "bands", or "none" yet."""
        return {0: "chain", 1: "matrix", 2: "bands", 3: "walk"}.get(self.lib.specinv_transform_objective_kind(self._h), "none")

    @property
    def launch_geometry(self) -> dict:
        """Diagnostics: how the iteration kernel is launched (`specinv_plan_launch_geometry`)."""
        out = (C.c_int32 * 4)()
        _lib.check(self.lib.specinv_plan_launch_geometry(self._h, out))
        kernel = ("k_iter_pair", "k_fused4", "k_fused", "k_semi", "k_hop", "k_fused4_td", "k_fused_td", "k_hop_td", "k_wave_iter", "k_wave_iter")[out[3]]
        geo = {"waves_per_workgroup": out[0], "chunks": out[1], "waves": out[2], "kernel": kernel}
        if kernel == "k_wave_iter":      # where its overlap-add runs: the frames buffer + k_ola, the kernel's registers or its LDS ring
            geo["overlap_add"] = "ring" if out[3] == 9 else ("registers" if out[1] < self.n_frames else "frames")
        return geo

    def force_generic(self, on=True):
        _lib.check(self.lib.specinv_plan_force_generic(self._h, int(on)))

    def set_exact(self, on=True):
        """The arithmetic of the magnitude projection and of the envelope division on the float32 wave-level kernels
        (torch_specinv/methods.py:132,246-247).  True (the library's default): the reference's operation order, (S m) r with r the
        correctly rounded 1 / |S|, and a correctly rounded division by the envelope; False: the approximate copies of the kernels
        (S (m rsq(|S|^2)), multiplication by 1 / envelope; 3 % faster on the headline step).  Call before `gla_init` /
        `admm_init`: the next init picks the kernels (a running method keeps its own)."""
        _lib.check(self.lib.specinv_plan_set_exact(self._h, int(on)))

    def keep_state(self, on=True):
        """Make the reference's spectral state readable through `state_spec`: X and U of ADMM (the fast paths carry only
        Y = X + U otherwise), pre_spec of griffin_lim (the hop = n_fft/4 kernel carries its momentum as a signal otherwise).
        Call before `admm_init` / `gla_init`."""
        _lib.check(self.lib.specinv_plan_keep_state(self._h, int(on)))

    # -- building blocks ------------------------------------------------------------------
    def stft(self, x: torch.Tensor) -> torch.Tensor:
        self._sync_stream()
        x = self._in(x, self.dtype)
        assert x.dim() == 2 and x.shape[0] == self.batch
        out = torch.empty(self._spec_shape(), dtype=self.cdtype, device=self.device)
        _lib.check(self.lib.specinv_stft(self._h, x.data_ptr(), x.shape[1], out.data_ptr()))
        return out

    def istft(self, spec: torch.Tensor) -> torch.Tensor:
        self._sync_stream()
        spec = self._in(spec, self.cdtype, self._spec_shape())
        out = torch.empty((self.batch, self.length), dtype=self.dtype, device=self.device)
        _lib.check(self.lib.specinv_istft(self._h, spec.data_ptr(), out.data_ptr()))
        return out

    def envelope(self) -> torch.Tensor:
        self._sync_stream()
        out = torch.empty((self.length,), dtype=self.dtype, device=self.device)
        _lib.check(self.lib.specinv_envelope(self._h, out.data_ptr()))
        return out

    def phase_init(self, mag: torch.Tensor) -> torch.Tensor:
        self._sync_stream()
        mag = self._in(mag, self.dtype, self._spec_shape())
        out = torch.empty(self._spec_shape(), dtype=self.cdtype, device=self.device)
        _lib.check(self.lib.specinv_phase_init(self._h, mag.data_ptr(), out.data_ptr()))
        return out

    def metric_sums(self, a: torch.Tensor, b: torch.Tensor):
        self._sync_stream()
        a = self._in(a, self.dtype)
        b = self._in(b, self.dtype, a.shape)
        sums = (C.c_double * 4)()
        _lib.check(self.lib.specinv_metric_sums(self._h, a.data_ptr(), b.data_ptr(), a.numel(), sums))
        return list(sums)

    # -- iterative methods ----------------------------------------------------------------
    def _init(self, fn, init_spec, mag, coef):
        self._sync_stream()
        keep = []
        ip = mp = None
        if init_spec is not None:
            keep.append(self._in(init_spec, self.cdtype, self._spec_shape()))
            ip = keep[-1].data_ptr()
        if mag is not None:
            keep.append(self._in(mag, self.dtype, self._spec_shape()))
            mp = keep[-1].data_ptr()
        _lib.check(fn(self._h, ip, mp, float(coef)))

    def gla_init(self, init_spec, mag, alpha):
        self._init(self.lib.specinv_gla_init, init_spec, mag, alpha)
        self._method = "gla"

    def admm_init(self, init_spec, mag, rho):
        self._init(self.lib.specinv_admm_init, init_spec, mag, rho)
        self._method = "admm"

    def iterate(self, n_iter: int, eval_last: bool = False):
        self._sync_stream()
        fn = self.lib.specinv_gla_iterate if self._method == "gla" else self.lib.specinv_admm_iterate
        sums = (C.c_double * 4)()
        _lib.check(fn(self._h, int(n_iter), int(eval_last), sums))
        return list(sums) if eval_last else None

    def iterate_dev(self, n_iter: int, out: torch.Tensor):
        """`iterate(n_iter, eval_last=True)` with the evaluation's four sums left in `out` (4 float64 on the plan's device): no
        host synchronisation - `distributed.run_loop_global` all-reduces them where they are."""
        assert out.dtype == torch.float64 and out.numel() >= 4 and out.is_contiguous() and out.device == self.device
        self._sync_stream()
        _lib.check(self.lib.specinv_iterate_eval_dev(self._h, int(n_iter), C.c_void_p(out.data_ptr())))

    def run(self, max_iter, eva_iter=10, tol=0.0, metric="sc", callback=None):
        """The reference's `_training_loop` (methods.py:153-190) executed by the library.
        Returns (iterations_done, [(iteration, metric, loss), ...])."""
        self._sync_stream()
        assert isinstance(metric, str) and metric.upper() in _lib.METRICS          # :167-168
        fn = self.lib.specinv_gla_run if self._method == "gla" else self.lib.specinv_admm_run
        cap = max(1, int(max_iter) // max(1, int(eva_iter)) + 1)
        evals = (_lib.Eval * cap)()
        n_ev, done = C.c_int(0), C.c_int(0)
        if callback is not None:
            def _cb(evp, _user):
                e = evp.contents
                return int(bool(callback(e.iteration, e.metric, e.loss)))
            cb = _lib.EVAL_CB(_cb)
        else:
            cb = _lib.EVAL_CB()
        _lib.check(fn(self._h, int(max_iter), int(eva_iter), float(tol), _lib.METRICS[metric.upper()],
                      evals, C.byref(n_ev), C.byref(done), cb, None))
        return done.value, [(evals[i].iteration, evals[i].metric, evals[i].loss) for i in range(n_ev.value)]

    def wave(self) -> torch.Tensor:
        self._sync_stream()
        out = torch.empty((self.batch, self.length), dtype=self.dtype, device=self.device)
        _lib.check(self.lib.specinv_get_wave(self._h, out.data_ptr()))
        return out

    def state_spec(self, which=0) -> torch.Tensor:
        self._sync_stream()
        out = torch.empty(self._spec_shape(), dtype=self.cdtype, device=self.device)
        _lib.check(self.lib.specinv_get_state_spec(self._h, int(which), out.data_ptr()))
        return out

    # -- differentiation building blocks ------------------------------------------------------------
    def gla_update(self, R, P, mag, lr):
        """S = R - lr*P ; Q = S*mag/(|S| + 1e-16)  (methods.py:243-247).  Returns (S, Q)."""
        self._sync_stream()
        S, Q = torch.empty_like(R), torch.empty_like(R)
        _lib.check(self.lib.specinv_gla_update(self._h, R.data_ptr(), P.data_ptr(), mag.data_ptr(), float(lr),
                                               S.data_ptr(), Q.data_ptr()))
        return S, Q

    def gla_update_adjoint(self, gQ, gP_next, S, mag, lr, gmag):
        """Adjoint of `gla_update`; accumulates d/dmag into `gmag`, returns (gR, gP)."""
        self._sync_stream()
        gR, gP = torch.empty_like(gQ), torch.empty_like(gQ)
        _lib.check(self.lib.specinv_gla_update_adjoint(
            self._h, gQ.data_ptr(), None if gP_next is None else gP_next.data_ptr(), S.data_ptr(), mag.data_ptr(),
            float(lr), gR.data_ptr(), gP.data_ptr(), gmag.data_ptr()))
        return gR, gP

    def admm_update(self, R, X, U, mag, rho):
        """methods.py:467-475 without the transforms.  Returns (X', U', V, Y') with V = Z - U' (pre-projection)."""
        self._sync_stream()
        Xn, Un, V, Yn = (torch.empty_like(R) for _ in range(4))
        _lib.check(self.lib.specinv_admm_update(self._h, R.data_ptr(), X.data_ptr(), U.data_ptr(), mag.data_ptr(), float(rho),
                                                Xn.data_ptr(), Un.data_ptr(), V.data_ptr(), Yn.data_ptr()))
        return Xn, Un, V, Yn

    def admm_update_adjoint(self, gYn, gXn, gUn, V, mag, rho, gmag):
        self._sync_stream()
        gR, gX, gU = (torch.empty_like(gYn) for _ in range(3))
        _lib.check(self.lib.specinv_admm_update_adjoint(
            self._h, gYn.data_ptr(), None if gXn is None else gXn.data_ptr(), None if gUn is None else gUn.data_ptr(),
            V.data_ptr(), mag.data_ptr(), float(rho), gR.data_ptr(), gX.data_ptr(), gU.data_ptr(), gmag.data_ptr()))
        return gR, gX, gU

    def istft_adjoint(self, g_x: torch.Tensor) -> torch.Tensor:
        self._sync_stream()
        g_x = self._in(g_x, self.dtype, (self.batch, self.length))
        out = torch.empty(self._spec_shape(), dtype=self.cdtype, device=self.device)
        _lib.check(self.lib.specinv_istft_adjoint(self._h, g_x.data_ptr(), out.data_ptr()))
        return out

    def stft_adjoint(self, g_spec: torch.Tensor, length: int) -> torch.Tensor:
        self._sync_stream()
        g_spec = self._in(g_spec, self.cdtype, self._spec_shape())
        out = torch.empty((self.batch, int(length)), dtype=self.dtype, device=self.device)
        _lib.check(self.lib.specinv_stft_adjoint(self._h, g_spec.data_ptr(), int(length), out.data_ptr()))
        return out

    def phase_init_adjoint(self, mag, g_spec, gmag):
        """gmag += d/dmag <g_spec, phase_init(mag)>."""
        self._sync_stream()
        _lib.check(self.lib.specinv_phase_init_adjoint(self._h, mag.data_ptr(), g_spec.data_ptr(), gmag.data_ptr()))

    def rtisi(self, mag, look_ahead, asymmetric_window, max_iter, alpha) -> torch.Tensor:
        self._sync_stream()
        mag = self._in(mag, self.dtype, self._spec_shape())
        out = torch.empty((self.batch, self.length), dtype=self.dtype, device=self.device)
        _lib.check(self.lib.specinv_rtisi_run(self._h, mag.data_ptr(), int(look_ahead), int(bool(asymmetric_window)),
                                              int(max_iter), float(alpha), out.data_ptr()))
        return out

    def rtisi_recorded(self, mag, look_ahead, asymmetric_window, max_iter, alpha):
        """RTISI_LA on the generic kernel, also returning the record the adjoint needs."""
        self._sync_stream()
        mag = self._in(mag, self.dtype, self._spec_shape())
        n = C.c_int64(0)
        _lib.check(self.lib.specinv_rtisi_record_elems(self._h, int(look_ahead), int(max_iter), C.byref(n)))
        rec = torch.empty(n.value, dtype=self.cdtype, device=self.device)
        out = torch.empty((self.batch, self.length), dtype=self.dtype, device=self.device)
        _lib.check(self.lib.specinv_rtisi_run_recorded(self._h, mag.data_ptr(), int(look_ahead), int(bool(asymmetric_window)),
                                                       int(max_iter), float(alpha), out.data_ptr(), rec.data_ptr()))
        return out, rec

    def rtisi_adjoint(self, mag, rec, g_x, look_ahead, asymmetric_window, max_iter, alpha):
        self._sync_stream()
        g_x = self._in(g_x, self.dtype, (self.batch, self.length))
        gmag = torch.empty(self._spec_shape(), dtype=self.dtype, device=self.device)
        _lib.check(self.lib.specinv_rtisi_adjoint(self._h, mag.data_ptr(), rec.data_ptr(), g_x.data_ptr(), int(look_ahead),
                                                  int(bool(asymmetric_window)), int(max_iter), float(alpha), gmag.data_ptr()))
        return gmag

    # -- streaming RTISI-LA (the plan's n_frames = most frames per push) ------------------------
    def rtisi_stream_begin(self, look_ahead, asymmetric_window, max_iter, alpha):
        self._sync_stream()
        _lib.check(self.lib.specinv_rtisi_stream_begin(self._h, int(look_ahead), int(bool(asymmetric_window)),
                                                       int(max_iter), float(alpha)))

    def rtisi_stream_push(self, mag: torch.Tensor) -> torch.Tensor:
        self._sync_stream()
        k = int(mag.shape[2])
        mag = self._in(mag, self.dtype, (self.batch, self.n_freq, k))
        out = torch.empty((self.batch, k * self.args.hop_length), dtype=self.dtype, device=self.device)
        n = C.c_int64(0)
        _lib.check(self.lib.specinv_rtisi_stream_push(self._h, mag.data_ptr(), k, out.data_ptr(), out.shape[1], C.byref(n)))
        return out[:, :n.value]

    def rtisi_stream_flush(self, look_ahead: int) -> torch.Tensor:
        self._sync_stream()
        cap = look_ahead * self.args.hop_length + self.args.n_fft
        out = torch.empty((self.batch, cap), dtype=self.dtype, device=self.device)
        n = C.c_int64(0)
        _lib.check(self.lib.specinv_rtisi_stream_flush(self._h, out.data_ptr(), cap, C.byref(n)))
        return out[:, :n.value]

    # -- L_BFGS building blocks -------------------------------------------------------------
    def transform_setup(self, kind: int, mel_fb: torch.Tensor | None = None):
        self._sync_stream()
        if mel_fb is not None:
            self._mel = self._in(mel_fb, self.dtype)
            assert self._mel.dim() == 2 and self._mel.shape[1] == self.n_freq
            _lib.check(self.lib.specinv_transform_setup(self._h, kind, self._mel.data_ptr(), self._mel.shape[0]))
            self.n_out = self._mel.shape[0]
        else:
            _lib.check(self.lib.specinv_transform_setup(self._h, kind, None, 0))
            self.n_out = self.n_freq

    def transform_forward(self, x: torch.Tensor) -> torch.Tensor:
        self._sync_stream()
        x = self._in(x, self.dtype)
        out = torch.empty((self.batch, self.n_out, self.n_frames), dtype=self.dtype, device=self.device)
        _lib.check(self.lib.specinv_transform_forward(self._h, x.data_ptr(), x.shape[-1], out.data_ptr()))
        return out

    def transform_loss_grad(self, x: torch.Tensor, target: torch.Tensor):
        self._sync_stream()
        x = self._in(x, self.dtype)
        target = self._in(target, self.dtype, (self.batch, self.n_out, self.n_frames))
        grad = torch.empty_like(x)
        loss = C.c_double(0)
        _lib.check(self.lib.specinv_transform_loss_grad(self._h, x.data_ptr(), x.shape[-1], target.data_ptr(),
                                                        C.byref(loss), grad.data_ptr()))
        return loss.value, grad

    def vec_dot(self, a, b) -> float:
        self._sync_stream()
        out = C.c_double(0)
        _lib.check(self.lib.specinv_vec_dot(self._h, a.data_ptr(), b.data_ptr(), a.numel(), C.byref(out)))
        return out.value

    def vec_axpy(self, alpha, x, y):
        """y += alpha * x (in place)."""
        self._sync_stream()
        _lib.check(self.lib.specinv_vec_axpy(self._h, float(alpha), x.data_ptr(), y.data_ptr(), x.numel()))

    def vec_scale(self, alpha, x, y):
        """y = alpha * x."""
        self._sync_stream()
        _lib.check(self.lib.specinv_vec_scale(self._h, float(alpha), x.data_ptr(), y.data_ptr(), x.numel()))

    def lbfgs_direction(self, g, s_list, y_list, rho, h_diag):
        """d = -H g by the two-loop recursion, all dot products kept on the device."""
        self._sync_stream()
        m = len(s_list)
        d = torch.empty_like(g)
        sp = (C.c_void_p * max(1, m))(*[t.data_ptr() for t in s_list])
        yp = (C.c_void_p * max(1, m))(*[t.data_ptr() for t in y_list])
        rh = (C.c_double * max(1, m))(*[float(r) for r in rho])
        _lib.check(self.lib.specinv_lbfgs_direction(self._h, g.data_ptr(), sp, yp, rh, m, float(h_diag), d.data_ptr(),
                                                    g.numel()))
        return d

    def vec_multi_dot(self, g, vecs):
        """[g . v for v in vecs] in one pass over g and the vectors (float64 accumulation)."""
        self._sync_stream()
        k = len(vecs)
        g = g if g.data_ptr() % 16 == 0 else g.clone()                    # (the kernels load 16 bytes per lane)
        vecs = [t if t.data_ptr() % 16 == 0 else t.clone() for t in vecs]
        vp = (C.c_void_p * k)(*[t.data_ptr() for t in vecs])
        out = (C.c_double * k)()
        _lib.check(self.lib.specinv_vec_multi_dot(self._h, g.data_ptr(), vp, k, g.numel(), out))
        return list(out)

    def vec_lincomb(self, vecs, coefs):
        """sum_j coefs[j] * vecs[j] in one pass (float64 accumulation, rounded once)."""
        self._sync_stream()
        k = len(vecs)
        out = torch.empty_like(vecs[0])
        vecs = [t if t.data_ptr() % 16 == 0 else t.clone() for t in vecs]
        vp = (C.c_void_p * k)(*[t.data_ptr() for t in vecs])
        cf = (C.c_double * k)(*[float(c) for c in coefs])
        _lib.check(self.lib.specinv_vec_lincomb(self._h, vp, cf, k, out.numel(), out.data_ptr()))
        return out

    def vec_lincomb_step(self, vecs, coefs, t, x):
        """d = sum_j coefs[j] * vecs[j] and x += t * d in one pass; returns d."""
        self._sync_stream()
        assert x.is_contiguous() and x.data_ptr() % 16 == 0
        k = len(vecs)
        out = torch.empty_like(vecs[0])
        vecs = [v if v.data_ptr() % 16 == 0 else v.clone() for v in vecs]
        vp = (C.c_void_p * k)(*[v.data_ptr() for v in vecs])
        cf = (C.c_double * k)(*[float(c) for c in coefs])
        _lib.check(self.lib.specinv_vec_lincomb_step(self._h, vp, cf, k, out.numel(), out.data_ptr(), float(t), x.data_ptr()))
        return out

    def lbfgs_pair(self, g, g_prev, d, t):
        """y = g - g_prev, s = t*d in one pass; returns (y, s, y.s, y.y)."""
        self._sync_stream()
        y, s = torch.empty_like(g), torch.empty_like(g)
        out = (C.c_double * 2)()
        _lib.check(self.lib.specinv_lbfgs_pair(self._h, g.data_ptr(), g_prev.data_ptr(), d.data_ptr(), float(t), y.data_ptr(),
                                               s.data_ptr(), g.numel(), out))
        return y, s, out[0], out[1]

    def lbfgs_stats(self, g, d):
        """(g.d, sum|g|, max|g|, max|d|) in one pass."""
        self._sync_stream()
        out = (C.c_double * 4)()
        _lib.check(self.lib.specinv_lbfgs_stats(self._h, g.data_ptr(), d.data_ptr(), g.numel(), out))
        return out[0], out[1], out[2], out[3]

    # -- the same passes with device-resident results (one L-BFGS iteration = one host synchronisation) ----------------
    def transform_loss_grad_dev(self, x: torch.Tensor, target: torch.Tensor, loss_ptr: int) -> torch.Tensor:
        self._sync_stream()
        x = self._in(x, self.dtype)
        target = self._in(target, self.dtype, (self.batch, self.n_out, self.n_frames))
        grad = torch.empty_like(x)
        _lib.check(self.lib.specinv_transform_loss_grad_dev(self._h, x.data_ptr(), x.shape[-1], target.data_ptr(), loss_ptr,
                                                            grad.data_ptr()))
        return grad

    def transform_loss_grad_stats_dev(self, x: torch.Tensor, target: torch.Tensor, d, out_ptr: int) -> torch.Tensor:
        """gradient now; {loss, g.d, sum|g|, max|g|, max|d|} left in device memory at `out_ptr` (d None: d = g)."""
        self._sync_stream()
        x = self._in(x, self.dtype)
        target = self._in(target, self.dtype, (self.batch, self.n_out, self.n_frames))
        grad = torch.empty_like(x)
        if d is not None:
            assert d.is_contiguous() and d.dtype == x.dtype and d.numel() == x.numel() and d.device == x.device
        _lib.check(self.lib.specinv_transform_loss_grad_stats_dev(self._h, x.data_ptr(), x.shape[-1], target.data_ptr(),
                                                                  d.data_ptr() if d is not None else None, out_ptr, grad.data_ptr()))
        return grad

    def vec_multi_dot_dev(self, g, vecs, out_ptr: int):
        self._sync_stream()
        k = len(vecs)
        assert g.data_ptr() % 16 == 0 and all(t.data_ptr() % 16 == 0 for t in vecs)
        vp = (C.c_void_p * k)(*[t.data_ptr() for t in vecs])
        _lib.check(self.lib.specinv_vec_multi_dot_dev(self._h, g.data_ptr(), vp, k, g.numel(), out_ptr))

    def lbfgs_pair_dev(self, g, g_prev, d, t, out_ptr: int):
        """y = g - g_prev, s = t*d; {y.s, y.y, g.g, g.g_prev} to device memory.  Returns (y, s)."""
        self._sync_stream()
        y, s = torch.empty_like(g), torch.empty_like(g)
        _lib.check(self.lib.specinv_lbfgs_pair_dev(self._h, g.data_ptr(), g_prev.data_ptr(), d.data_ptr(), float(t), y.data_ptr(),
                                                   s.data_ptr(), g.numel(), out_ptr))
        return y, s

    def lbfgs_pair_stats_dev(self, g, g_prev, d, t, out_ptr: int):
        """Pair and statistics in one pass: {g.d, sum|g|, max|g|, max|d|, y.s, y.y, g.g, g.g_prev} to device memory."""
        self._sync_stream()
        y, s = torch.empty_like(g), torch.empty_like(g)
        _lib.check(self.lib.specinv_lbfgs_pair_stats_dev(self._h, g.data_ptr(), g_prev.data_ptr(), d.data_ptr(), float(t),
                                                         y.data_ptr(), s.data_ptr(), g.numel(), out_ptr))
        return y, s

    def lbfgs_stats_dev(self, g, d, out_ptr: int):
        """{g.d, sum|g|, max|g|, max|d|} to device memory."""
        self._sync_stream()
        _lib.check(self.lib.specinv_lbfgs_stats_dev(self._h, g.data_ptr(), d.data_ptr(), g.numel(), out_ptr))

    # ---- the device-resident optimiser (csrc/lbfgs_dev.h) --------------------------------------------------------------
    def lbfgs_dev_create(self, n: int, lr, max_iter, max_eval, tolerance_grad, tolerance_change, history_size,
                         time_objective=False) -> int:
        self._sync_stream()
        o = _lib.LbfgsOpts(float(lr), float(tolerance_grad), float(tolerance_change), int(max_iter),
                           int(max_eval) if max_eval is not None else 0, int(history_size), int(time_objective))
        h = C.c_int32(-1)
        _lib.check(self.lib.specinv_lbfgs_dev_create(self._h, int(n), C.byref(o), C.byref(h)))
        return h.value

    def lbfgs_dev_step(self, handle: int, x: torch.Tensor, target: torch.Tensor):
        """One optimizer.step on x (in place), everything enqueued, one synchronisation.  Returns the info record."""
        self._sync_stream()
        assert x.is_contiguous() and x.dtype == torch.float32 and x.device == self.device
        target = self._in(target, self.dtype, (self.batch, self.n_out, self.n_frames))
        info = _lib.LbfgsInfo()
        _lib.check(self.lib.specinv_lbfgs_dev_step(self._h, handle, x.data_ptr(), x.shape[-1], target.data_ptr(), C.byref(info)))
        return info

    def lbfgs_dev_destroy(self, handle: int):
        if self._h:
            _lib.check(self.lib.specinv_lbfgs_dev_destroy(self._h, handle))

    def board_alloc(self, n: int):
        """n doubles of pinned host memory the device writes directly: (host ctypes array, device address)."""
        hp, dp = C.c_void_p(), C.c_void_p()
        _lib.check(self.lib.specinv_board_alloc(self._h, int(n), C.byref(hp), C.byref(dp)))
        return (C.c_double * n).from_address(hp.value), dp.value

    def stream_wait(self):
        self._sync_stream()
        _lib.check(self.lib.specinv_stream_wait(self._h))

    def read_doubles(self, ptr: int, n: int):
        self._sync_stream()
        out = (C.c_double * n)()
        _lib.check(self.lib.specinv_read_doubles(self._h, ptr, n, out))
        return list(out)

    def vec_absmax_abssum(self, x):
        self._sync_stream()
        out = (C.c_double * 2)()
        _lib.check(self.lib.specinv_vec_absmax_abssum(self._h, x.data_ptr(), x.numel(), out))
        return out[0], out[1]


# small LRU of plans so that repeated calls with one configuration reuse device state.  A plan carries the state of one
# run, so it must not be shared by two threads: the cache is per thread.  It is bounded by count and by the device
# memory the cached plans hold (a BASELINE C2 plan is 1.2 GB): least recently used plans go first, the plan just
# asked for always stays.
_TLS = threading.local()
_CACHE_MAX = 4
_CACHE_MAX_BYTES = int(float(os.environ.get("SPECINV_PLAN_CACHE_GB", "4")) * (1 << 30))


def _cache() -> "OrderedDict[tuple, Plan]":
    c = getattr(_TLS, "plans", None)
    if c is None:
        c = _TLS.plans = OrderedDict()
    return c


def get_plan(args: StftArgs, batch: int, n_frames: int, dtype: torch.dtype, device: torch.device) -> Plan:
    key = (str(device), dtype, batch, n_frames, args.n_fft, args.hop_length, args.center, args.pad_mode,
           args.normalized, args.onesided, args.window.to(torch.float64).numpy().tobytes())
    cache = _cache()
    plan = cache.get(key)
    if plan is None:
        plan = Plan(args, batch, n_frames, dtype, device)
        cache[key] = plan
    else:
        cache.move_to_end(key)
        plan.keep_state(False)                 # (a per-run request: the next user of a cached plan gets the default kernels)
    plan.set_exact(exact_projection())
    trim_plan_cache()
    return plan


_EXACT = [None]


def set_exact_projection(on: bool | None):
    """Module-level switch for the drop-in functions (their signatures are the reference's, so the choice cannot be an
    argument): True = the reference's operation order in the projection and a true envelope division on every plan the public
    functions create from now on (the default), False = the approximate kernels (3 % faster on the headline step),
    None = follow the environment (SPECINV_EXACT=0 selects the approximations)."""
    _EXACT[0] = on


def has_approx() -> bool:
    """Does the loaded library carry the approximate-projection kernels (built with SPECINV_BUILD_APPROX=1)?  Without them
    `set_exact(False)` / `set_exact_projection(False)` / SPECINV_EXACT=0 are accepted and every plan keeps the reference's operation
    order."""
    return bool(_lib.load().specinv_has_approx())


def exact_projection() -> bool:
    if _EXACT[0] is not None:
        return bool(_EXACT[0])
    return os.environ.get("SPECINV_EXACT", "1") != "0"


def trim_plan_cache():
    """Enforce the cache bounds.  Plans reserve their buffers as they are used, so the public functions call this
    again when they are done with a plan; the most recently used plan always stays."""
    cache = _cache()
    while len(cache) > 1 and (len(cache) > _CACHE_MAX or sum(p.device_bytes for p in cache.values()) > _CACHE_MAX_BYTES):
        cache.popitem(last=False)


def clear_plan_cache():
    """Drop the calling thread's cached plans (their device memory is released with them)."""
    _cache().clear()
