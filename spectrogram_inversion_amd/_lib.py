"""ctypes binding of libspecinv.so (the C ABI declared in include/specinv.h).

There is no CPU fallback: if the HIP library is missing or cannot be loaded this
module raises immediately.
"""
from __future__ import annotations

import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
# SPECINV_LIB selects a tuning variant built by tools/sweep_variants.sh (development only)
LIB_PATH = os.environ.get("SPECINV_LIB") or os.path.join(_PKG, "libspecinv.so")

OK, EINVAL, EHIP, EUNSUPPORTED, ENOMEM, ESTATE = 0, -1, -2, -3, -4, -5
F32, F64 = 0, 1
PAD_MODES = {"reflect": 0, "constant": 1, "replicate": 2, "circular": 3}
METRICS = {"SC": 0, "SNR": 1, "SER": 2}
TF_MAG, TF_LOGMEL = 0, 1


class StftCfg(C.Structure):
    _fields_ = [("n_fft", C.c_int32), ("hop_length", C.c_int32), ("n_frames", C.c_int32),
                ("batch", C.c_int32), ("center", C.c_int32), ("pad_mode", C.c_int32),
                ("normalized", C.c_int32), ("onesided", C.c_int32), ("dtype", C.c_int32),
                ("device", C.c_int32), ("window_host", C.c_void_p)]


class Eval(C.Structure):
    _fields_ = [("iteration", C.c_int32), ("metric", C.c_double), ("loss", C.c_double)]


class LbfgsOpts(C.Structure):
    _fields_ = [("lr", C.c_double), ("tolerance_grad", C.c_double), ("tolerance_change", C.c_double),
                ("max_iter", C.c_int32), ("max_eval", C.c_int32), ("history_size", C.c_int32), ("time_objective", C.c_int32)]


class LbfgsInfo(C.Structure):
    _fields_ = [("first_loss", C.c_double), ("loss", C.c_double), ("t", C.c_double),
                ("total_iters", C.c_int32), ("func_evals", C.c_int32), ("n_iter", C.c_int32), ("history_len", C.c_int32),
                ("pairs_accepted", C.c_int32), ("pairs_rejected", C.c_int32), ("objective_launches", C.c_int32),
                ("objective_timed", C.c_int32), ("objective_ms", C.c_double),
                ("lean_iterations", C.c_int32), ("full_iterations", C.c_int32), ("suspensions", C.c_int32), ("reserved_", C.c_int32)]


EVAL_CB = C.CFUNCTYPE(C.c_int, C.POINTER(Eval), C.c_void_p)

_P = C.c_void_p
_I64 = C.c_int64
_D = C.c_double
_DP = C.POINTER(C.c_double)
_IP = C.POINTER(C.c_int)

# name -> (restype, argtypes); must list every function of include/specinv.h
SIGNATURES = {
    "specinv_last_error": (C.c_char_p, []),
    "specinv_abi_version": (C.c_int, []),
    "specinv_has_approx": (C.c_int, []),
    "specinv_iterate_eval_dev": (C.c_int, [_P, C.c_int, _P]),
    "specinv_plan_create": (C.c_int, [C.POINTER(StftCfg), C.POINTER(_P)]),
    "specinv_plan_destroy": (C.c_int, [_P]),
    "specinv_plan_set_stream": (C.c_int, [_P, _P]),
    "specinv_plan_n_freq": (C.c_int, [_P]),
    "specinv_plan_length": (_I64, [_P]),
    "specinv_plan_fast_path": (C.c_int, [_P]),
    "specinv_transform_objective_kind": (C.c_int, [_P]),
    "specinv_plan_device_bytes": (_I64, [_P]),
    "specinv_plan_launch_geometry": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "specinv_plan_force_generic": (C.c_int, [_P, C.c_int]),
    "specinv_plan_set_exact": (C.c_int, [_P, C.c_int]),
    "specinv_plan_keep_state": (C.c_int, [_P, C.c_int]),
    "specinv_stft": (C.c_int, [_P, _P, _I64, _P]),
    "specinv_istft": (C.c_int, [_P, _P, _P]),
    "specinv_envelope": (C.c_int, [_P, _P]),
    "specinv_phase_init": (C.c_int, [_P, _P, _P]),
    "specinv_metric_sums": (C.c_int, [_P, _P, _P, _I64, _DP]),
    "specinv_gla_init": (C.c_int, [_P, _P, _P, _D]),
    "specinv_gla_iterate": (C.c_int, [_P, C.c_int, C.c_int, _DP]),
    "specinv_gla_run": (C.c_int, [_P, C.c_int, C.c_int, _D, C.c_int, C.POINTER(Eval), _IP, _IP, EVAL_CB, _P]),
    "specinv_admm_init": (C.c_int, [_P, _P, _P, _D]),
    "specinv_admm_iterate": (C.c_int, [_P, C.c_int, C.c_int, _DP]),
    "specinv_admm_run": (C.c_int, [_P, C.c_int, C.c_int, _D, C.c_int, C.POINTER(Eval), _IP, _IP, EVAL_CB, _P]),
    "specinv_get_wave": (C.c_int, [_P, _P]),
    "specinv_get_state_spec": (C.c_int, [_P, C.c_int, _P]),
    "specinv_gla_update": (C.c_int, [_P, _P, _P, _P, _D, _P, _P]),
    "specinv_gla_update_adjoint": (C.c_int, [_P, _P, _P, _P, _P, _D, _P, _P, _P]),
    "specinv_admm_update": (C.c_int, [_P, _P, _P, _P, _P, _D, _P, _P, _P, _P]),
    "specinv_admm_update_adjoint": (C.c_int, [_P, _P, _P, _P, _P, _P, _D, _P, _P, _P, _P]),
    "specinv_istft_adjoint": (C.c_int, [_P, _P, _P]),
    "specinv_stft_adjoint": (C.c_int, [_P, _P, _I64, _P]),
    "specinv_phase_init_adjoint": (C.c_int, [_P, _P, _P, _P]),
    "specinv_rtisi_run": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _D, _P]),
    "specinv_rtisi_record_elems": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    "specinv_rtisi_run_recorded": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _D, _P, _P]),
    "specinv_rtisi_adjoint": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _D, _P]),
    "specinv_rtisi_stream_begin": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _D]),
    "specinv_rtisi_stream_push": (C.c_int, [_P, _P, C.c_int, _P, _I64, C.POINTER(C.c_int64)]),
    "specinv_rtisi_stream_flush": (C.c_int, [_P, _P, _I64, C.POINTER(C.c_int64)]),
    "specinv_transform_setup": (C.c_int, [_P, C.c_int, _P, C.c_int]),
    "specinv_transform_forward": (C.c_int, [_P, _P, _I64, _P]),
    "specinv_transform_loss_grad": (C.c_int, [_P, _P, _I64, _P, _DP, _P]),
    "specinv_vec_dot": (C.c_int, [_P, _P, _P, _I64, _DP]),
    "specinv_vec_axpy": (C.c_int, [_P, _D, _P, _P, _I64]),
    "specinv_vec_scale": (C.c_int, [_P, _D, _P, _P, _I64]),
    "specinv_vec_absmax_abssum": (C.c_int, [_P, _P, _I64, _DP]),
    "specinv_lbfgs_pair": (C.c_int, [_P, _P, _P, _P, _D, _P, _P, _I64, _DP]),
    "specinv_lbfgs_stats": (C.c_int, [_P, _P, _P, _I64, _DP]),
    "specinv_lbfgs_direction": (C.c_int, [_P, _P, C.POINTER(_P), C.POINTER(_P), _DP, C.c_int, _D, _P, _I64]),
    "specinv_vec_multi_dot": (C.c_int, [_P, _P, C.POINTER(_P), C.c_int, _I64, _DP]),
    "specinv_vec_lincomb": (C.c_int, [_P, C.POINTER(_P), _DP, C.c_int, _I64, _P]),
    "specinv_board_alloc": (C.c_int, [_P, C.c_int, C.POINTER(_P), C.POINTER(_P)]),
    "specinv_stream_wait": (C.c_int, [_P]),
    "specinv_lbfgs_dev_create": (C.c_int, [_P, _I64, _P, C.POINTER(C.c_int32)]),
    "specinv_lbfgs_dev_step": (C.c_int, [_P, C.c_int32, _P, _I64, _P, _P]),
    "specinv_lbfgs_dev_destroy": (C.c_int, [_P, C.c_int32]),
    "specinv_vec_lincomb_step": (C.c_int, [_P, C.POINTER(_P), _DP, C.c_int, _I64, _P, C.c_double, _P]),
    "specinv_transform_loss_grad_dev": (C.c_int, [_P, _P, _I64, _P, _P, _P]),
    "specinv_transform_loss_grad_stats_dev": (C.c_int, [_P, _P, _I64, _P, _P, _P, _P]),
    "specinv_vec_multi_dot_dev": (C.c_int, [_P, _P, C.POINTER(_P), C.c_int, _I64, _P]),
    "specinv_lbfgs_pair_dev": (C.c_int, [_P, _P, _P, _P, _D, _P, _P, _I64, _P]),
    "specinv_lbfgs_stats_dev": (C.c_int, [_P, _P, _P, _I64, _P]),
    "specinv_lbfgs_pair_stats_dev": (C.c_int, [_P, _P, _P, _P, _D, _P, _P, _I64, _P]),
    "specinv_read_doubles": (C.c_int, [_P, _P, C.c_int, _DP]),
}

_lib = None


class SpecinvError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libspecinv error {code}: {msg}")
        self.code = code


def load():
    """Load libspecinv.so once; fail loudly if the HIP extension is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -m spectrogram_inversion_amd.build). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    ver = lib.specinv_abi_version()
    if ver != 1:
        raise RuntimeError(f"libspecinv ABI version {ver} != 1 (stale build?)")
    _lib = lib
    return lib


def check(code: int):
    """Map a non-zero return code to an exception the way the reference would raise:
    argument errors are bare asserts there (methods.py:79,101,163-168,223,...)."""
    if code == OK:
        return
    msg = load().specinv_last_error().decode("utf-8", "replace")
    if code == EINVAL:
        raise AssertionError(msg)
    if code == EUNSUPPORTED:
        raise NotImplementedError(msg)
    raise SpecinvError(code, msg)
