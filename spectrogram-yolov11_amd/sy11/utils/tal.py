"""Anchor grid and box <-> side-distance codecs under the reference's names (ultralytics/utils/tal.py: make_anchors
:334-346, dist2bbox :349-358, bbox2dist :361-364) — host-side helpers for callers that want the tables as tensors.

The reference's ``TaskAlignedAssigner`` (tal.py:14-296) has no tensor-op counterpart here: assignment runs inside the fused
HIP criterion (csrc/loss.hip ``loss_assign_*``, entered through ``sy11.utils.loss.v8DetectionLoss``), and Detect's decode is
``sy11_detect_decode``.  The CPU restatement of the assigner that the parity tests check against is oracle/loss_ref.py."""
from __future__ import annotations

import torch


def _level_hw(level):
    if torch.is_tensor(level):
        return int(level.shape[-2]), int(level.shape[-1])
    h, w = level
    return int(h), int(w)


def make_anchors(feats, strides, grid_cell_offset=0.5):
    """-> (points (A, 2) as (x, y) cell centres in grid units, stride (A, 1)); levels concatenated in the given order,
    cells row-major inside a level.  ``feats``: maps (.., H, W) or (h, w) pairs."""
    first = feats[0]
    kw = dict(dtype=first.dtype, device=first.device) if torch.is_tensor(first) else dict(dtype=torch.float32)
    points, per_anchor_stride = [], []
    for level, s in zip(feats, strides):
        h, w = _level_hw(level)
        cell = torch.arange(h * w, **{**kw, "dtype": torch.int64})
        xy = torch.stack((cell % w, cell // w), 1).to(kw["dtype"]) + grid_cell_offset
        points.append(xy)
        per_anchor_stride.append(xy.new_full((h * w, 1), float(s)))
    return torch.cat(points), torch.cat(per_anchor_stride)


def dist2bbox(distance, anchor_points, xywh=True, dim=-1):
    """(left, top, right, bottom) distances from an anchor point -> box (xywh centre form, or xyxy corners)."""
    near, far = torch.split(distance, 2, dim)
    lo, hi = anchor_points - near, anchor_points + far
    return torch.cat((0.5 * (lo + hi), hi - lo) if xywh else (lo, hi), dim)


def bbox2dist(anchor_points, bbox, reg_max):
    """xyxy box -> (left, top, right, bottom) distances from the anchor point, limited to [0, reg_max - 0.01]."""
    lo, hi = torch.split(bbox, 2, -1)
    return torch.cat((anchor_points - lo, hi - anchor_points), -1).clamp_(0, reg_max - 0.01)
