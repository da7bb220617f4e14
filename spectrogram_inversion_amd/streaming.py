"""Frame-by-frame RTISI-LA.

The reference's `RTISI_LA` (torch_specinv/methods.py:273-412) is the off-line form of a real-time algorithm
(:275-278): step i only ever looks at target frames i-look_ahead .. i.  `RTISIStream` exposes exactly that
recursion incrementally: push magnitude frames as they arrive, get the samples that became final; the
concatenation of everything returned equals `RTISI_LA` on the whole spectrogram (same kernel, same order of
operations - bit for bit against the generic kernel, to rounding against the wave-level one).

    s = RTISIStream(n_freq=1025, batch=1, look_ahead=3, max_iter=25, hop_length=512, window=w)
    for frames in source:                 # (B, F, k) or (F, k), k <= max_push
        audio.append(s.push(frames))      # (B, n) finished samples, n <= k * hop
    audio.append(s.flush())
"""
from __future__ import annotations

import torch

from .plan import Plan, args_helper, require_gpu


class RTISIStream:
    def __init__(self, n_freq, batch=1, look_ahead=-1, asymmetric_window=False, max_iter=25, alpha=0.99,
                 max_push=16, dtype=torch.float32, device=None, **stft_kwargs):
        assert max_iter > 0                                        # methods.py:295
        assert alpha >= 0                                          # :296
        assert max_push >= 1
        probe = torch.empty((1, int(n_freq), 1), dtype=dtype)
        self.args = args_helper(probe, **stft_kwargs)              # :305
        self.device = require_gpu(device)
        keep = (self.args.n_fft - 1) // self.args.hop_length       # :322 (win_length is n_fft after :80-83)
        self.look_ahead = keep if look_ahead < 0 else int(look_ahead)   # :323-324
        self.batch, self.max_push = int(batch), max(2, int(max_push))
        self.plan = Plan(self.args, self.batch, self.max_push, dtype, self.device)   # owned: state lives in it
        self.plan.rtisi_stream_begin(self.look_ahead, asymmetric_window, max_iter, alpha)
        self.frames_in = 0
        self.samples_out = 0
        self._done = False
        self._squeeze, self._out_device = False, None

    @property
    def latency_frames(self):
        """A sample is final once this many later frames have been pushed."""
        return self.look_ahead

    def push(self, mag: torch.Tensor) -> torch.Tensor:
        assert not self._done, "push after flush"
        assert not mag.is_complex()                                # methods.py:297
        squeeze = mag.dim() == 2
        m3 = mag.unsqueeze(0) if squeeze else mag
        assert m3.dim() == 3 and m3.shape[0] == self.batch and m3.shape[1] == self.plan.n_freq
        pieces = []
        for j in range(0, m3.shape[2], self.max_push):             # longer blocks go in max_push slices
            pieces.append(self.plan.rtisi_stream_push(m3[:, :, j:j + self.max_push]))
        self.frames_in += int(m3.shape[2])
        y = pieces[0] if len(pieces) == 1 else torch.cat(pieces, 1)
        self.samples_out += int(y.shape[1])
        self._squeeze, self._out_device = squeeze and self.batch == 1, mag.device
        y = y.to(mag.device)
        return y[0] if self._squeeze else y

    def flush(self) -> torch.Tensor:
        assert not self._done, "flush twice"
        y = self.plan.rtisi_stream_flush(self.look_ahead)
        self._done = True
        self.samples_out += int(y.shape[1])
        if self._out_device is not None:
            y = y.to(self._out_device)                            # same device / rank as what push returned
        return y[0] if self._squeeze else y
