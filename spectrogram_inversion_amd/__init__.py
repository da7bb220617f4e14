"""spectrogram_inversion_amd - MI355X-native spectrogram inversion.

Drop-in for the public surface of `torch_specinv` v0.2.1
(torch_specinv/__init__.py:6: L_BFGS, RTISI_LA, griffin_lim, ADMM, phase_init; plus
torch_specinv.metrics: sc, snr, ser), implemented as HIP kernels for gfx950 behind the C ABI
of include/specinv.h.  Importing the package does not load the HIP library; the first call
does, and fails loudly if it is missing (no CPU fallback).

    import spectrogram_inversion_amd as torch_specinv
    y = torch_specinv.griffin_lim(mag, max_iter=100, alpha=0.3, hop_length=512, window=w)
"""
name = "spectrogram_inversion_amd"
__version__ = "0.1.0"

from .methods import L_BFGS, RTISI_LA, griffin_lim, ADMM, phase_init   # noqa: E402,F401
from . import metrics                                                   # noqa: E402,F401
from .metrics import sc, snr, ser                                       # noqa: E402,F401
from .transforms import MagSTFT, LogMelSTFT                             # noqa: E402,F401
from .streaming import RTISIStream                                      # noqa: E402,F401
from .mel import mel_filterbank                                         # noqa: E402,F401
from .plan import set_exact_projection, has_approx                      # noqa: E402,F401
