"""Flat parameter / buffer storage for the trainer.

The reference updates ~255 parameter tensors and ~160 BatchNorm buffers one by one (optimizer, GradScaler.unscale_,
clip_grad_norm_, ModelEMA: engine/trainer.py:585-593, utils/torch_utils.py:495-531).  On an MI355X those are
hundreds of launch-bound micro-kernels (3.6 ms per step measured) for ~300 MB of traffic (50 us at HBM speed).
Here every trainable parameter becomes a VIEW of one flat f32 buffer, laid out as [decay weights | norm weights |
biases] — the three optimizer groups of trainer.py:776-813 are three contiguous slices, aligned element for element
with the GradStore's flat gradient buffer.  Module / state_dict / checkpoint layout is unchanged.
"""
from __future__ import annotations

import torch
import torch.nn as nn


def param_groups(model):
    """(decay, norm, bias) parameter lists by the reference's name/type rules."""
    g = [], [], []
    bn = tuple(v for k, v in nn.__dict__.items() if "Norm" in k)
    for module_name, module in model.named_modules():
        for param_name, param in module.named_parameters(recurse=False):
            if not param.requires_grad:
                continue
            fullname = f"{module_name}.{param_name}" if module_name else param_name
            if "bias" in fullname:
                g[2].append(param)
            elif isinstance(module, bn):
                g[1].append(param)
            else:
                g[0].append(param)
    return g


def padded(n: int) -> int:
    """Every parameter starts on a 32-byte (f32) / 16-byte (f16 working copy) boundary: the kernels' vector loads need
    it, and e.g. the fusion variant's 18-element spatial-attention filter would otherwise misalign everything after it.
    The pad elements are zero in the parameter, gradient and EMA buffers and stay zero under SGD / AdamW."""
    return (n + 7) // 8 * 8


def _view_like(seg: torch.Tensor, p: torch.Tensor) -> torch.Tensor:
    if p.dim() == 4:                                   # conv filter: OIHW shape over [O][KH][KW][I] memory
        o, i, kh, kw = p.shape
        return seg.view(o, kh, kw, i).permute(0, 3, 1, 2)
    return seg.view(p.shape)


class FlatState:
    """Re-homes a model's trainable parameters (and float buffers) into flat buffers; keeps slices per group."""

    def __init__(self, model: nn.Module):
        groups = param_groups(model)
        self.order = [p for g in groups for p in g]
        dev = self.order[0].device
        total = sum(padded(p.numel()) for p in self.order)
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.group_slices = []
        off = 0
        for g in groups:
            start = off
            for p in g:
                n = p.numel()
                v = _view_like(self.flat[off:off + n], p)
                v.copy_(p.data)
                p.data = v
                p._sy11_owner = self.flat          # in-place updates of the flat buffer invalidate per-parameter caches (conv.working_filter)
                off += padded(n)
            self.group_slices.append((start, off))
        bufs = [b for b in model.buffers() if b.dtype.is_floating_point]
        nb = sum(b.numel() for b in bufs)
        self.flat_buf = torch.empty(nb, dtype=torch.float32, device=dev)
        off = 0
        for b in bufs:
            n = b.numel()
            v = self.flat_buf[off:off + n].view(b.shape)
            v.copy_(b.data)
            b.data = v
            off += n

        self.offsets = {}
        off = 0
        for p in self.order:
            self.offsets[id(p)] = off
            off += padded(p.numel())
        self._wt_desc = None

    def group_tensors(self, flat: torch.Tensor):
        return [flat[a:b] for a, b in self.group_slices]

    # ---- per-step derived weight banks (one launch each instead of one per layer)
    def working_views(self, dtype):
        """{id(param): [O][KH][KW][I] view} over ONE cast of the flat buffer (f32: the flat buffer itself)."""
        bank = self.flat if dtype == torch.float32 else self.flat.to(dtype)
        out = {}
        for p in self.order:
            if p.dim() == 4:
                o, i, kh, kw = p.shape
                a = self.offsets[id(p)]
                out[id(p)] = bank[a:a + p.numel()].view(o, kh, kw, i)
        return bank, out

    def transposed_views(self, bank: torch.Tensor):
        """{id(param): [I][KH][KW][O] view} of every groups==1 conv filter, produced by one batched transpose launch."""
        import ctypes as C
        import numpy as np
        from .. import _lib, ops
        if self._wt_desc is None:
            rec = np.dtype([("N", "<i4"), ("T", "<i4"), ("C", "<i4"), ("pad", "<i4"), ("src", "<i8"), ("dst", "<i8"),
                            ("first", "<i4"), ("tc", "<i4"), ("tn", "<i4"), ("pad2", "<i4")])
            rows, tiles, self._wt_params = [], 0, []
            for p in self.order:
                if p.dim() == 4 and getattr(p, "_sy11_groups", 1) == 1:
                    o, i, kh, kw = p.shape
                    tc, tn = -(-i // 32), -(-o // 32)
                    a = self.offsets[id(p)]
                    rows.append((o, kh * kw, i, 0, a, a, tiles, tc, tn, 0))
                    tiles += kh * kw * tc * tn
                    self._wt_params.append(p)
            arr = np.array(rows, dtype=rec)
            self._wt_desc = torch.from_numpy(arr.view(np.uint8).copy()).to(self.flat.device)
            self._wt_tiles = tiles
        dst = torch.empty_like(bank)
        _lib.call("sy11_weight_transpose_multi", ops.dt_code(bank.dtype), len(self._wt_params), self._wt_tiles,
                  C.c_void_p(self._wt_desc.data_ptr()), C.c_void_p(bank.data_ptr()), C.c_void_p(dst.data_ptr()), ops._stream())
        out = {}
        for p in self._wt_params:
            o, i, kh, kw = p.shape
            a = self.offsets[id(p)]
            out[id(p)] = dst[a:a + p.numel()].view(i, kh, kw, o)
        return out


class FlatEMA:
    """ModelEMA semantics (decay 0.9999 * (1 - exp(-updates / tau)) over every float state entry) in 4 launches."""

    def __init__(self, model, state: FlatState, decay=0.9999, tau=2000, updates=0):
        import math
        from copy import deepcopy
        self.ema = deepcopy(model).eval()
        for k in list(self.ema.__dict__):
            if k.startswith("_sy11_"):
                del self.ema.__dict__[k]
        for p, q in zip(model.parameters(), self.ema.parameters()):
            q.requires_grad_(p.requires_grad)           # same membership / order as the source model's flat layout
        self.ema_state = FlatState(self.ema)
        for p in self.ema.parameters():
            p.requires_grad_(False)
        self.src = state
        self.updates = updates
        self.decay = lambda x: decay * (1 - math.exp(-x / tau))
        self.enabled = True
        # frozen parameters (DFL arange) are not in the flat buffers; they never change, nothing to average

    def update(self, model=None):
        if not self.enabled:
            return
        self.updates += 1
        d = self.decay(self.updates)
        self.ema_state.flat.mul_(d).add_(self.src.flat, alpha=1 - d)
        self.ema_state.flat_buf.mul_(d).add_(self.src.flat_buf, alpha=1 - d)
