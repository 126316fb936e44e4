"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product package).

CPU statement of the IQ -> STFT -> power -> "mel" -> log -> min-max -> 3-channel image producer.

The reference holds NO implementation of this stage (README.md:7 prose only), so there is nothing
to restate and nothing to pin against: **PARITY UNPINNED** by construction.  The spec below is the build's
own; the FFT is anchored on ``torch.stft`` (Hann periodic window, center=False, two-sided).

Spec: n_fft=1024, hop=256, frames=640 (L = 1024 + 639*256 = 164608 complex samples), fftshift, |X|^2,
640 triangular filters on a mildly log-warped two-sided frequency axis (each FFT bin feeds <=2 filters),
10*log10(p + 1e-10), per-image min-max to [0,1], image[b, c, f, t] replicated to 3 channels.
"""
from __future__ import annotations

import math

import numpy as np
import torch

N_FFT = 1024
HOP = 256
N_FRAMES = 640
N_MEL = 640
N_SAMPLES = N_FFT + (N_FRAMES - 1) * HOP
WARP_ALPHA = 1.25
MEL_TAPS = 8          # max FFT bins under one triangular filter (checked in mel_table)
LOG_EPS = 1e-10


def mel_edges(n_mel=N_MEL, n_fft=N_FFT, alpha=WARP_ALPHA):
    """n_mel+2 edge positions (float bin coordinates on the fftshift-ed axis), monotone, symmetric about DC."""
    m = np.linspace(-1.0, 1.0, n_mel + 2, dtype=np.float64)
    u = np.sign(m) * np.expm1(np.abs(m) * math.log1p(alpha)) / alpha
    return (n_fft / 2) * (1.0 + u) * (n_fft - 1) / n_fft


def mel_table(n_mel=N_MEL, n_fft=N_FFT, alpha=WARP_ALPHA):
    """Gather form: for filter j, first bin ``start[j]`` and MEL_TAPS weights (zero padded), float32."""
    p = mel_edges(n_mel, n_fft, alpha)
    start = np.zeros(n_mel, dtype=np.int32)
    wts = np.zeros((n_mel, MEL_TAPS), dtype=np.float32)
    for j in range(1, n_mel + 1):
        lo, c, hi = p[j - 1], p[j], p[j + 1]
        k0 = int(math.ceil(lo))
        k1 = int(math.floor(hi))
        ks = [k for k in range(k0, k1 + 1) if 0 <= k < n_fft]
        assert len(ks) <= MEL_TAPS, (j, len(ks))
        start[j - 1] = ks[0] if ks else 0
        for t, k in enumerate(ks):
            w = (k - lo) / (c - lo) if k <= c else (hi - k) / (hi - c)
            wts[j - 1, t] = max(w, 0.0)
    return start, wts


def mel_matrix(n_mel=N_MEL, n_fft=N_FFT, alpha=WARP_ALPHA):
    start, wts = mel_table(n_mel, n_fft, alpha)
    M = np.zeros((n_mel, n_fft), dtype=np.float32)
    for j in range(n_mel):
        for t in range(MEL_TAPS):
            k = start[j] + t
            if k < n_fft and wts[j, t] != 0:
                M[j, k] = wts[j, t]
    return M


def logmel_db(iq: torch.Tensor) -> torch.Tensor:
    """(B, N_SAMPLES) complex64 -> (B, N_MEL, N_FRAMES) float32 dB (before normalisation)."""
    win = torch.hann_window(N_FFT, periodic=True, dtype=torch.float32)
    X = torch.stft(iq, N_FFT, hop_length=HOP, win_length=N_FFT, window=win, center=False,
                   onesided=False, return_complex=True)            # (B, 1024, frames)
    X = torch.fft.fftshift(X, dim=1)
    P = X.real ** 2 + X.imag ** 2
    M = torch.from_numpy(mel_matrix())
    mel = torch.einsum("jk,bkt->bjt", M, P)
    return 10.0 * torch.log10(mel + LOG_EPS)


def spectrogram_image(iq: torch.Tensor) -> torch.Tensor:
    """(B, N_SAMPLES) complex64 -> (B, 3, 640, 640) float32 in [0, 1]."""
    db = logmel_db(iq)
    lo = db.amin(dim=(1, 2), keepdim=True)
    hi = db.amax(dim=(1, 2), keepdim=True)
    img = (db - lo) / (hi - lo).clamp(min=1e-12)
    return img.unsqueeze(1).expand(-1, 3, -1, -1).contiguous()


def synthetic_iq(batch: int, seed: int = 1) -> torch.Tensor:
    """SURVEY §8(d) synthetic IQ: complex white noise + one band-limited OFDM-like burst + one chirp."""
    g = torch.Generator().manual_seed(seed)
    n = N_SAMPLES
    noise = (torch.randn(batch, n, generator=g) + 1j * torch.randn(batch, n, generator=g)) / math.sqrt(2)
    t = torch.arange(n, dtype=torch.float32)
    out = noise.to(torch.complex64) * 0.1
    for b in range(batch):
        # OFDM-like burst: 64 random-phase subcarriers in a band, gated in time
        f0 = -0.25 + 0.5 * ((b * 37) % 64) / 64.0
        k = torch.arange(64, dtype=torch.float32)
        ph = torch.rand(64, generator=g) * 2 * math.pi
        freqs = f0 + (k - 32) * (0.08 / 64)
        t0, t1 = int(n * 0.2), int(n * 0.55)
        seg = t[t0:t1]
        burst = torch.exp(1j * (2 * math.pi * freqs[:, None] * seg[None, :] + ph[:, None])).sum(0) / 8.0
        out[b, t0:t1] += burst.to(torch.complex64)
        # linear chirp
        c0, c1 = 0.1, 0.35
        tt = t[int(n * 0.6):]
        tt = tt - tt[0]
        phase = 2 * math.pi * (c0 * tt + 0.5 * (c1 - c0) / tt.numel() * tt * tt)
        out[b, int(n * 0.6):] += (0.7 * torch.exp(1j * phase)).to(torch.complex64)
    return out
