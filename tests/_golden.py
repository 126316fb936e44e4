"""Helpers to compare tensors with the committed golden fixtures (tests/golden/*.npz)."""
from pathlib import Path

import numpy as np
import torch

GOLD = Path(__file__).resolve().parent / "golden"
_cache = {}


def load(name):
    if name not in _cache:
        _cache[name] = dict(np.load(GOLD / name, allow_pickle=False))
    return _cache[name]


def check(store, key, t, rtol=1e-4, atol=1e-5, what=""):
    """Compare ``t`` against the stored strided sample and float64 moments of the reference's tensor."""
    t = t.detach().to("cpu", torch.float32).contiguous()
    shape = tuple(store[key + ".shape"].tolist())
    assert tuple(t.shape) == shape, f"{what}{key}: shape {tuple(t.shape)} != golden {shape}"
    stride = int(store[key + ".stride"])
    got = t.flatten()[::stride].numpy()
    ref = store[key + ".s"]
    scale = max(float(np.abs(ref).max()), 1e-30) if ref.size else 1.0
    err = np.abs(got - ref)
    tol = atol * max(scale, 1.0) + rtol * np.abs(ref)
    bad = err > tol
    assert not bad.any(), (f"{what}{key}: {int(bad.sum())}/{ref.size} sampled values off; max err "
                           f"{float(err.max()):.3e} at ref={float(ref[err.argmax()]):.5e} scale={scale:.3e}")
    m = store[key + ".m"]
    d = t.flatten().double()
    n = max(d.numel(), 1)
    abs_sum = float(d.abs().sum())
    # moments: sum tolerance scales with Σ|x| (cancellation), Σ|x| and Σx² relative
    assert abs(float(d.sum()) - m[0]) <= (rtol * 10) * max(m[1], 1e-12) + atol * n, f"{what}{key}: sum"
    assert abs(abs_sum - m[1]) <= (rtol * 10) * max(m[1], 1e-12) + atol * n, f"{what}{key}: abs-sum"
    assert abs(float((d * d).sum()) - m[2]) <= (rtol * 20) * max(m[2], 1e-12) + atol * n, f"{what}{key}: sq-sum"
