"""The C-ABI shared library builds, loads and exports every symbol include/specinv.h declares.
CPU only: no compute entry point is called (argument-error paths only)."""
import ctypes as C
import os
import re

import pytest

from _util import ROOT
from spectrogram_inversion_amd import _lib, build

HEADER = os.path.join(ROOT, "include", "specinv.h")


@pytest.fixture(scope="module")
def lib():
    build.build_lib()           # hipcc cross-compiles gfx950 without a GPU
    return _lib.load()


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(specinv_[a-z0-9_]+)\s*\(", text))
    names.discard("specinv_eval_cb")
    return sorted(names)


def test_header_and_binding_agree():
    assert set(declared_functions()) == set(_lib.SIGNATURES), \
        set(declared_functions()) ^ set(_lib.SIGNATURES)


def test_exports_every_declared_symbol(lib):
    raw = C.CDLL(_lib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(raw, name), f"{name} is declared in specinv.h but not exported"
    assert lib.specinv_abi_version() == 1


def test_code_object_is_gfx950():
    with open(_lib.LIB_PATH, "rb") as fh:
        blob = fh.read()
    assert b"gfx950" in blob


def test_argument_errors_do_not_need_a_gpu(lib):
    handle = C.c_void_p()
    assert lib.specinv_plan_create(None, C.byref(handle)) == _lib.EINVAL
    assert b"null" in lib.specinv_last_error()
    cfg = _lib.StftCfg(n_fft=512, hop_length=128, n_frames=4, batch=1, center=1, pad_mode=0, normalized=0,
                       onesided=1, dtype=7, device=0, window_host=None)
    assert lib.specinv_plan_create(C.byref(cfg), C.byref(handle)) == _lib.EINVAL
    assert lib.specinv_plan_n_freq(None) == _lib.EINVAL
    assert lib.specinv_gla_iterate(None, 1, 0, None) == _lib.EINVAL
    assert lib.specinv_plan_destroy(None) == _lib.OK
    with pytest.raises(AssertionError):
        _lib.check(_lib.EINVAL)
    with pytest.raises(NotImplementedError):
        _lib.check(_lib.EUNSUPPORTED)
