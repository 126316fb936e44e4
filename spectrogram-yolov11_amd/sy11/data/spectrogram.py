"""IQ -> STFT -> power -> triangular "mel" bank -> dB -> min-max -> (B, 3, 640, 640) image producer on the MI355X.

The reference contains NO implementation of this stage (README.md:7 prose only); it plugs in where
``DetectionTrainer.preprocess_batch`` (models/yolo/detect/train.py:57-74) / ``BasePredictor.preprocess``
(engine/predictor.py:118-136) produce the float image.  Spec (build-defined): n_fft 1024, hop 256,
periodic Hann, two-sided (complex IQ), 640 frames, fftshift, |X|^2, 640 triangular filters on a log-warped
two-sided axis (<= 8 bins per filter), 10*log10(p + 1e-10), per-image min-max, 3 identical channels.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .. import ops


class SpectrogramProducer:
    def __init__(self, device="cuda", n_fft=1024, hop=256, n_frames=640, n_mel=640, warp_alpha=1.25, mel_taps=8):
        self.n_fft, self.hop, self.n_frames, self.n_mel, self.mel_taps = n_fft, hop, n_frames, n_mel, mel_taps
        self.n_samples = n_fft + (n_frames - 1) * hop
        start, wts = self.filter_bank(n_mel, n_fft, warp_alpha, mel_taps)
        self.device = torch.device(device)
        self.window = torch.hann_window(n_fft, periodic=True, dtype=torch.float32).to(self.device)
        self.mel_start = torch.from_numpy(start).to(self.device)
        self.mel_w = torch.from_numpy(wts).to(self.device)

    @staticmethod
    def filter_bank(n_mel, n_fft, alpha, taps):
        """Gather-form triangular bank: first FFT bin and `taps` weights per filter (zero padded)."""
        m = np.linspace(-1.0, 1.0, n_mel + 2, dtype=np.float64)
        u = np.sign(m) * np.expm1(np.abs(m) * math.log1p(alpha)) / alpha
        p = (n_fft / 2) * (1.0 + u) * (n_fft - 1) / n_fft
        start = np.zeros(n_mel, dtype=np.int32)
        wts = np.zeros((n_mel, taps), dtype=np.float32)
        for j in range(1, n_mel + 1):
            lo, c, hi = p[j - 1], p[j], p[j + 1]
            ks = [k for k in range(int(math.ceil(lo)), int(math.floor(hi)) + 1) if 0 <= k < n_fft]
            if len(ks) > taps:
                raise ValueError(f"filter {j} spans {len(ks)} bins > mel_taps={taps}")
            start[j - 1] = ks[0] if ks else 0
            for t, k in enumerate(ks):
                wts[j - 1, t] = max((k - lo) / (c - lo) if k <= c else (hi - k) / (hi - c), 0.0)
        return start, wts

    def logmel_db(self, iq: torch.Tensor):
        """(B, n_samples) complex64 on device -> (db (B, frames, n_mel), minmax (B, 2))."""
        if iq.shape[1] < self.n_samples:
            raise ValueError(f"need >= {self.n_samples} IQ samples per image, got {iq.shape[1]}")
        return ops.stft_logmel(iq, self.window, self.mel_start, self.mel_w, self.n_fft, self.hop, self.n_frames, self.n_mel)

    def __call__(self, iq: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        """(B, n_samples) complex64 -> (B, 3, n_mel, n_frames) f32 in [0, 1] (NCHW, what the model's stem reads).
        ``out``: write the image there (the trainer passes the captured graph's static input: no 315 MB copy per step)."""
        db, mm = self.logmel_db(iq)
        return ops.stft_normalize(db, mm, out)
