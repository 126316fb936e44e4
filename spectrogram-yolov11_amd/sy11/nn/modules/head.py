"""Detect head with the reference's name / signature / state_dict keys (ultralytics/nn/modules/head.py:21-172),
executed by libsy11.  Training returns the three raw maps; eval returns (decoded (B,4+nc,A), maps)."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from ... import ops
from ...engine import Act, Ctx, run_module
from .block import DFL
from .conv import Conv, DWConv, conv2d_bias_run

__all__ = ("Detect",)


class Detect(nn.Module):
    """YOLO Detect head for detection models."""

    dynamic = False
    export = False
    format = None
    end2end = False
    max_det = 300
    shape = None
    anchors = torch.empty(0)
    strides = torch.empty(0)
    legacy = False

    def __init__(self, nc=80, ch=()):
        super().__init__()
        self.nc = nc
        self.nl = len(ch)
        self.reg_max = 16
        self.no = nc + self.reg_max * 4
        self.stride = torch.zeros(self.nl)
        c2, c3 = max((16, ch[0] // 4, self.reg_max * 4)), max(ch[0], min(self.nc, 100))
        self.cv2 = nn.ModuleList(
            nn.Sequential(Conv(x, c2, 3), Conv(c2, c2, 3), nn.Conv2d(c2, 4 * self.reg_max, 1)) for x in ch
        )
        self.cv3 = (
            nn.ModuleList(nn.Sequential(Conv(x, c3, 3), Conv(c3, c3, 3), nn.Conv2d(c3, self.nc, 1)) for x in ch)
            if self.legacy
            else nn.ModuleList(
                nn.Sequential(
                    nn.Sequential(DWConv(x, x, 3), Conv(x, c3, 1)),
                    nn.Sequential(DWConv(c3, c3, 3), Conv(c3, c3, 1)),
                    nn.Conv2d(c3, self.nc, 1),
                )
                for x in ch
            )
        )
        self.dfl = DFL(self.reg_max) if self.reg_max > 1 else nn.Identity()
        for seq in list(self.cv2) + list(self.cv3):
            seq[-1].weight.data = seq[-1].weight.data.contiguous(memory_format=torch.channels_last)

    def forward(self, x):
        """List of 3 feature maps -> list of 3 raw maps (training) or (y, maps) (eval).  NOTE: like the reference
        (head.py:69-70) the incoming list is overwritten with the raw maps."""
        outs = list(run_module(self, x))
        if self.training:
            for i in range(self.nl):
                x[i] = outs[i]
            return x
        for i in range(self.nl):
            x[i] = outs[1 + i]
        return outs[0] if self.export else (outs[0], x)

    def bias_init(self):
        """Initialize Detect() biases, WARNING: requires stride availability (head.py:133-140)."""
        for a, b, s in zip(self.cv2, self.cv3, self.stride):
            a[-1].bias.data[:] = 1.0
            b[-1].bias.data[: self.nc] = math.log(5 / self.nc / (640 / s) ** 2)

    # ---- engine
    def _branch(self, ec: Ctx, seq: nn.Sequential, x: Act, out: Act):
        h = x
        for m in list(seq)[:-1]:
            if isinstance(m, nn.Sequential):
                for mm in m:
                    h = mm._run(ec, h)
            else:
                h = m._run(ec, h)
        conv2d_bias_run(ec, seq[-1], h, out)

    def _level(self, ec: Ctx, i: int, x: Act) -> Act:
        """Both chains of level ``i`` (box: cv2[i], class: cv3[i]) into one (B, H, W, no) map."""
        B, H, W, _ = x.shape
        m = Act(ec.empty(B, H, W, self.no, dtype=torch.float32))      # logits stay f32 (loss / decode)
        self._branch(ec, self.cv2[i], x, m.slice(0, 4 * self.reg_max))
        self._branch(ec, self.cv3[i], x, m.slice(4 * self.reg_max, self.no))
        return m

    def _run(self, ec: Ctx, xs):
        if self.end2end:
            raise ops._lib.Sy11Error("end2end (v10) heads are outside the hot path")
        maps = []
        for i in range(self.nl):                                # the levels are independent: one stream (graph branch) each
            with ec.branch(i):
                maps.append(self._level(ec, i, xs[i]))
        ec.join_branches()
        if ec.training:
            return maps
        if self.reg_max != 16:
            raise ops._lib.Sy11Error("decode kernel assumes reg_max == 16")
        sf = self.__dict__.get("_stride_host")                 # host copy of the strides: no device read inside a graph capture
        if sf is None or len(sf) != len(self.stride):
            sf = self.__dict__["_stride_host"] = [float(v) for v in self.stride.tolist()]
        y = ops.detect_decode([m.data for m in maps], sf, self.nc)   # (B, 4+nc, A)
        return [_Raw(y)] + maps


class _Raw(Act):
    """An output that is already in the reference's layout (B, 4+nc, A): bypasses the NHWC->NCHW view."""

    def __init__(self, t):
        super().__init__(t)
