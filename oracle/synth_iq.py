"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product package).

Synthetic 5G/LTE-style scenes for the accuracy gate (BASELINE configs[4] / north_star "mAP@0.5 within 0.1 of reference on the
held-out set").  The reference's dataset is not distributed (README.md:19-20, cfg/datasets/Spectrogram.yaml), so the held-out
set is generated: complex noise plus 1-3 emitters per capture whose time-frequency extent is known exactly —
  class 0  an OFDM-like burst: many tones with random phases filling a band for a time interval ("LTE-like")
  class 1  a linear chirp sweeping a band over a time interval ("NR / radar-like")
and the box of an emitter in the spectrogram IMAGE (x = STFT frame, y = row of the log-warped "mel" axis, both normalised to
[0, 1]) follows from the producer's own spec (n_fft / hop / frames, oracle.stft_ref.mel_edges).  Labels are YOLO rows
(cls, cx, cy, w, h).  Parity status: build-defined data, nothing to pin."""
from __future__ import annotations

import math

import numpy as np
import torch


def freq_to_row(f, n_fft, n_mel, alpha):
    """Normalised frequency f in [-0.5, 0.5) -> (fractional) row of the n_mel-filter bank (inverse of stft_ref.mel_edges)."""
    b = (np.asarray(f, dtype=np.float64) + 0.5) * n_fft                     # bin on the fftshift-ed axis
    u = np.clip(b / ((n_fft / 2) * (n_fft - 1) / n_fft) - 1.0, -1.0, 1.0)
    m = np.sign(u) * np.log1p(alpha * np.abs(u)) / math.log1p(alpha)
    return (m + 1.0) / 2.0 * (n_mel + 1) - 1.0                              # edge index j has centre p[j]; filter j-1 is row j-1


def scene(seed, n_fft=512, hop=128, n_frames=320, n_mel=320, alpha=1.25):
    """-> (iq complex64 (L,), labels float32 (k, 5))."""
    rng = np.random.default_rng(seed)
    L = n_fft + (n_frames - 1) * hop
    iq = (rng.standard_normal(L) + 1j * rng.standard_normal(L)) * (0.05 / math.sqrt(2))
    t = np.arange(L, dtype=np.float64)
    labels = []
    for _ in range(int(rng.integers(1, 4))):
        cls = int(rng.integers(0, 2))
        s0 = int(L * rng.uniform(0.05, 0.55))
        s1 = min(L - 1, s0 + int(L * rng.uniform(0.15, 0.35)))
        fc, bw = rng.uniform(-0.33, 0.33), rng.uniform(0.05, 0.14)
        f0, f1 = fc - bw / 2, fc + bw / 2
        amp = rng.uniform(0.4, 1.0)
        seg = t[s0:s1] - s0
        if cls == 0:
            tones = np.linspace(f0, f1, 32)
            ph = rng.uniform(0, 2 * math.pi, 32)
            sig = np.exp(1j * (2 * math.pi * tones[:, None] * seg[None, :] + ph[:, None])).sum(0) / math.sqrt(32)
        else:
            sig = np.exp(1j * 2 * math.pi * (f0 * seg + 0.5 * (f1 - f0) / max(len(seg), 1) * seg * seg))
        iq[s0:s1] += amp * sig
        # a frame j covers samples [j*hop, j*hop + n_fft): the emitter shows in frames (s0 - n_fft)/hop .. s1/hop
        x0 = max((s0 - n_fft / 2) / hop, 0.0) / n_frames
        x1 = min((s1 - n_fft / 2) / hop, n_frames - 1.0) / n_frames
        y0, y1 = (freq_to_row(f, n_fft, n_mel, alpha) / n_mel for f in (f0, f1))
        y0, y1 = max(float(y0), 0.0), min(float(y1), 1.0)
        if x1 - x0 > 0.02 and y1 - y0 > 0.01:
            labels.append([cls, (x0 + x1) / 2, (y0 + y1) / 2, x1 - x0, y1 - y0])
    return torch.from_numpy(iq.astype(np.complex64)), torch.tensor(labels, dtype=torch.float32).reshape(-1, 5)


def dataset(n, seed0=0, **kw):
    """-> (iq (n, L) complex64, batch dict pieces: batch_idx (m,), cls (m, 1), bboxes (m, 4)) for scenes seed0 .. seed0 + n - 1."""
    iqs, bi, cl, bb = [], [], [], []
    for i in range(n):
        iq, lab = scene(seed0 + i, **kw)
        iqs.append(iq)
        for row in lab:
            bi.append(float(i))
            cl.append([float(row[0])])
            bb.append(row[1:].tolist())
    return torch.stack(iqs), torch.tensor(bi), torch.tensor(cl).reshape(-1, 1), torch.tensor(bb).reshape(-1, 4)


def take(batch_idx, cls, bboxes, lo, hi):
    """Labels of images lo .. hi-1 with image ids renumbered from 0 (one mini-batch)."""
    sel = (batch_idx >= lo) & (batch_idx < hi)
    return {"batch_idx": batch_idx[sel] - lo, "cls": cls[sel], "bboxes": bboxes[sel]}
