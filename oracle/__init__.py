"""CPU oracle for the spectrogram-inversion hot path (TEST INFRASTRUCTURE ONLY).

This package is a NumPy/SciPy restatement of the algorithms in the reference
`torch_specinv` (methods.py / metrics.py).  It is the *checker* for the HIP
path, never the product:

  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
    leg may import it;
  * the product package (`spectrogram_inversion_amd`) must never import it and
    has no CPU fallback.

Parity status: PINNED.  The reference's own tests hold no numeric vectors
(shape-only assertions, test/test_griffin.py:9-68), so the oracle is pinned
against golden fixtures produced by importing the unmodified reference in the
build container (`tests/golden/make_golden.py` -> `tests/golden/*.npz`, torch
2.10.0 CPU path).  `tests/test_oracle_golden.py` checks every fixture.

Every function cites the reference lines it restates (paths relative to the
reference checkout).
"""
from .stftlib import (StftArgs, args_helper, stft, istft, ola_envelope, frame_count,
                   signal_length)
from .metrics import sc, snr, ser, mse
from .methods import phase_init, griffin_lim, admm, rtisi_la
from .lbfgs import lbfgs_minimize, l_bfgs

__all__ = [
    "StftArgs", "args_helper", "stft", "istft", "ola_envelope", "frame_count",
    "signal_length", "sc", "snr", "ser", "mse", "phase_init", "griffin_lim",
    "admm", "rtisi_la", "lbfgs_minimize", "l_bfgs",
]
