"""Streaming RTISI-LA (`RTISIStream`) against the whole-signal `RTISI_LA`: pushing the spectrogram in pieces must
give the same waveform.  Needs an MI355X: `-m gpu`."""
import numpy as np
import pytest
import torch

from _util import hann, load_golden, rel_l2

pytestmark = pytest.mark.gpu

import spectrogram_inversion_amd as si                        # noqa: E402
from spectrogram_inversion_amd._lib import SpecinvError       # noqa: E402

DEV = torch.device("cuda", 0)


def N(t):
    return t.detach().cpu().numpy()


def _stream(mag, pieces, **kw):
    B, F, Tn = mag.shape
    s = si.RTISIStream(F, batch=B, dtype=mag.dtype, device=DEV, **kw)
    out, t = [], 0
    i = 0
    while t < Tn:
        k = pieces[i % len(pieces)]
        y = s.push(mag[:, :, t:t + k])
        assert y.shape[0] == B and y.shape[1] <= min(k, Tn - t) * s.args.hop_length
        out.append(y)
        t += k
        i += 1
    out.append(s.flush())
    y = torch.cat(out, 1)
    assert s.samples_out == y.shape[1] and s.frames_in == Tn
    return y


CASES = [
    # n_fft, hop, frames, look_ahead, asym, dtype, extra stft kwargs, pieces
    (256, 64, 23, -1, True, torch.float32, {}, [1]),
    (256, 64, 23, -1, False, torch.float32, {}, [3, 1, 5]),
    (256, 64, 23, 0, True, torch.float32, {}, [2]),
    (256, 64, 23, 2, True, torch.float64, {}, [7, 4]),
    (256, 100, 17, -1, True, torch.float32, {}, [1, 2]),                      # hop does not divide n_fft
    (256, 200, 9, 0, True, torch.float32, {}, [1]),                           # hop > n_fft/2, nothing to look ahead
    (128, 32, 19, 5, False, torch.float64, dict(center=False), [4]),          # look_ahead > kept frames
    (128, 32, 19, 2, True, torch.float64, dict(onesided=False, normalized=True), [6, 1]),
    (120, 30, 15, -1, True, torch.float64, {}, [2, 3]),                       # mixed-radix n_fft
    (256, 64, 3, -1, True, torch.float32, {}, [1]),                           # fewer frames than the look-ahead
    (256, 64, 1, 2, True, torch.float32, {}, [1]),                            # a single frame
]


@pytest.mark.parametrize("n_fft,hop,frames,la,asym,dtype,extra,pieces", CASES)
def test_stream_equals_whole_signal_bitwise(n_fft, hop, frames, la, asym, dtype, extra, pieces):
    """Below n_fft 1024 both forms run the same kernel: not one bit may differ."""
    g = torch.Generator().manual_seed(n_fft + hop + frames)
    F = n_fft if extra.get("onesided") is False else n_fft // 2 + 1
    mag = (torch.rand((2, F, frames), generator=g, dtype=dtype) + 0.05).to(DEV)
    kw = dict(look_ahead=la, asymmetric_window=asym, max_iter=3, alpha=0.99, hop_length=hop,
              window=torch.from_numpy(hann(n_fft, np.float64)).to(dtype), **extra)
    if frames == 1 and extra.get("center", True):
        with pytest.raises(AssertionError):      # a one-frame centred signal has no samples left after trimming
            si.RTISI_LA(mag, verbose=False, **kw)
        return
    y_ref = si.RTISI_LA(mag, verbose=False, **kw)
    y = _stream(mag, pieces, max_push=8, **kw)
    assert y.shape == y_ref.shape
    # (centre=False with a Hann window divides 0 by 0 at the first sample, like the reference: methods.py:132)
    assert torch.equal(torch.isnan(y), torch.isnan(y_ref))
    assert torch.equal(torch.nan_to_num(y), torch.nan_to_num(y_ref)), rel_l2(N(y), N(y_ref))


def test_stream_against_reference_fixture():
    """Straight against the reference's output (g5): same tolerance as the whole-signal test."""
    g = load_golden("g5_rtisi")
    for i, meta in enumerate(g["meta"]):
        hop, la, asym, alpha = str(meta).split("|")
        if not int(asym):
            continue
        mag = torch.from_numpy(g[f"mag_h{hop}"]).to(DEV)
        y = _stream(mag, [2, 1], max_push=4, look_ahead=int(la), asymmetric_window=True, max_iter=3, alpha=float(alpha),
                    hop_length=int(hop), window=torch.from_numpy(g["window"]))
        ref, ref64 = g[f"wave{i}"], g[f"wave64_{i}"]
        assert rel_l2(N(y), ref) < max(5 * rel_l2(ref, ref64), 1e-5)


@pytest.mark.parametrize("n_fft,hop,la,pieces", [(2048, 512, 3, [5]), (1024, 128, -1, [1, 3]), (512, 256, 2, [2]),
                                                 (1024, 256, 7, [4, 1]), (2048, 1024, 0, [1])])
def test_stream_on_the_wave_level_kernel(n_fft, hop, la, pieces):
    """Shapes the wave-level RTISI kernel covers: the stream resumes that kernel from its saved frame ring and registers
    and must reproduce the whole-signal call bit for bit; the generic kernels agree on the first frames (random
    magnitudes make the recursion amplify rounding differences from frame to frame)."""
    from spectrogram_inversion_amd.plan import args_helper, get_plan
    mag = torch.rand((2, n_fft // 2 + 1, 24), generator=torch.Generator().manual_seed(5)).to(DEV) + 0.05
    w = torch.hann_window(n_fft)
    kw = dict(look_ahead=la, asymmetric_window=True, max_iter=4, alpha=0.99, hop_length=hop, window=w)
    y = _stream(mag, pieces, max_push=5, **kw)
    y_fast = si.RTISI_LA(mag, verbose=False, **kw)
    assert torch.equal(y, y_fast), rel_l2(N(y), N(y_fast))
    plan = get_plan(args_helper(mag, hop_length=hop, window=w), 2, 24, torch.float32, DEV)
    plan.force_generic(True)
    try:
        y_gen = plan.rtisi(mag, la, True, 4, 0.99)
    finally:
        plan.force_generic(False)
    assert rel_l2(N(y[:, :n_fft]), N(y_gen[:, :n_fft])) < 2e-4


def test_stream_latency_and_counts():
    hop, n_fft, la = 64, 256, 2
    s = si.RTISIStream(129, batch=1, look_ahead=la, max_iter=2, hop_length=hop, window=torch.hann_window(n_fft), device=DEV)
    assert s.latency_frames == la
    got = [s.push(torch.rand(129, 1)).shape[-1] for _ in range(6)]
    # frame c is committed at push c + la; centre trimming eats the first n_fft/2 samples
    assert got == [0, 0, 0, 0, 64, 64]
    tail = s.flush()
    assert sum(got) + tail.shape[-1] == (6 - 1) * hop            # L of a 6-frame centred signal


def test_stream_2d_input_and_long_block():
    s = si.RTISIStream(65, look_ahead=1, max_iter=2, max_push=4, hop_length=32, window=torch.hann_window(128), device=DEV)
    mag = torch.rand(65, 11)
    y = torch.cat([s.push(mag), s.flush()])                     # 11 frames > max_push: sliced internally
    ref = si.RTISI_LA(mag, look_ahead=1, max_iter=2, verbose=False, hop_length=32, window=torch.hann_window(128))
    assert y.device.type == "cpu" and torch.equal(y, ref)


def test_stream_restart_and_errors():
    s = si.RTISIStream(65, look_ahead=1, max_iter=2, hop_length=32, window=torch.hann_window(128), device=DEV)
    with pytest.raises(SpecinvError):
        s.plan.rtisi_stream_flush(1)                            # nothing pushed yet
    s.push(torch.rand(65, 2))
    s.flush()
    with pytest.raises(AssertionError):
        s.push(torch.rand(65, 1))
    with pytest.raises(AssertionError):
        s.flush()
    with pytest.raises(AssertionError):
        s2 = si.RTISIStream(65, look_ahead=1, hop_length=32, device=DEV)
        s2.push(torch.rand(64, 2))                              # wrong number of bins
    with pytest.raises(AssertionError):
        si.RTISIStream(65, max_iter=0, device=DEV)
