"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product package).

CPU restatement (numpy / torch, loops) of the prediction post-processing path.

Reference lines followed (under /root/reference/ultralytics):
  utils/ops.py:181-332 non_max_suppression (best-class and multi_label branches, class offset 7680,
                       max_nms 30000 sort+truncate, max_det) — the wall-clock break at :328 is NOT
                       reproduced (it must never fire in a parity run)
  utils/ops.py:432-449 xywh2xyxy    utils/ops.py:92-127 scale_boxes    utils/ops.py:335-354 clip_boxes

Third-party core: ``torchvision.ops.nms`` (pyproject pins only torchvision>=0.9.0; not vendored, not installed
here).  Restated from its published semantics: stable sort by score descending, greedy keep, suppress j when
IoU(i, j) > iou_threshold (strict), IoU = inter / (area_i + area_j - inter) with w = max(0, x2 - x1),
kept indices returned in descending-score order.  Equal scores: this oracle (and the product) fix the
order "score desc, then original index asc".

Parity status: the wrapper is PINNED by tests/golden/nms_inputs.npz (pre-NMS boxes/scores captured from the
reference's own non_max_suppression); the nms core itself is PARITY UNPINNED (torchvision cannot be executed
here) — SURVEY §8(c).
"""
from __future__ import annotations

import numpy as np
import torch


def nms_core(boxes: np.ndarray, scores: np.ndarray, iou_thres: float) -> np.ndarray:
    """Greedy NMS, float32 arithmetic. Returns kept indices (int64) in descending-score order."""
    boxes = np.asarray(boxes, dtype=np.float32)
    scores = np.asarray(scores, dtype=np.float32)
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), dtype=np.int64)
    order = np.argsort(-scores, kind="stable")
    x1, y1, x2, y2 = (boxes[:, i] for i in range(4))
    areas = (x2 - x1) * (y2 - y1)
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    thr = np.float32(iou_thres)
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(np.float32(0), xx2 - xx1)
        h = np.maximum(np.float32(0), yy2 - yy1)
        inter = w * h
        iou = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[iou > thr]] = True
    return np.asarray(keep, dtype=np.int64)


def xywh2xyxy(x: torch.Tensor) -> torch.Tensor:
    xy, wh = x[..., :2], x[..., 2:] / 2
    return torch.cat((xy - wh, xy + wh), -1)


def pre_nms(pred_img: torch.Tensor, nc: int, conf_thres: float, multi_label: bool, max_nms=30000):
    """Rows handed to the nms core for one image: x = (n, 6) [xyxy, conf, cls]. ``pred_img`` is (A, 4+nc), xywh."""
    xc = pred_img[:, 4:4 + nc].amax(1) > conf_thres
    x = pred_img[xc]
    box = xywh2xyxy(x[:, :4])
    cls = x[:, 4:4 + nc]
    if multi_label and nc > 1:
        i, j = torch.where(cls > conf_thres)
        x = torch.cat((box[i], cls[i, j, None], j[:, None].float()), 1)
    else:
        conf, j = cls.max(1, keepdim=True)
        x = torch.cat((box, conf, j.float()), 1)[conf.view(-1) > conf_thres]
    if x.shape[0] > max_nms:
        x = x[x[:, 4].argsort(descending=True)[:max_nms]]
    return x


def non_max_suppression(prediction: torch.Tensor, conf_thres=0.25, iou_thres=0.45, multi_label=False,
                        max_det=300, nc=0, max_wh=7680, agnostic=False):
    """(B, 4+nc, A) -> list of (k, 6) [x1, y1, x2, y2, conf, cls] plus the kept pre-NMS row indices."""
    bs = prediction.shape[0]
    nc = nc or prediction.shape[1] - 4
    out, kept = [], []
    for xi in range(bs):
        x = pre_nms(prediction[xi].transpose(0, 1), nc, conf_thres, multi_label)
        if x.shape[0] == 0:
            out.append(torch.zeros((0, 6)))
            kept.append(np.zeros((0,), dtype=np.int64))
            continue
        c = x[:, 5:6] * (0 if agnostic else max_wh)
        i = nms_core((x[:, :4] + c).numpy(), x[:, 4].numpy(), iou_thres)[:max_det]
        out.append(x[torch.from_numpy(i)])
        kept.append(i)
    return out, kept


def clip_boxes(boxes, shape):
    boxes[..., 0] = boxes[..., 0].clamp(0, shape[1])
    boxes[..., 1] = boxes[..., 1].clamp(0, shape[0])
    boxes[..., 2] = boxes[..., 2].clamp(0, shape[1])
    boxes[..., 3] = boxes[..., 3].clamp(0, shape[0])
    return boxes


def scale_boxes(img1_shape, boxes, img0_shape):
    gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
    pad = (round((img1_shape[1] - img0_shape[1] * gain) / 2 - 0.1),
           round((img1_shape[0] - img0_shape[0] * gain) / 2 - 0.1))
    boxes = boxes.clone()
    boxes[..., 0] -= pad[0]
    boxes[..., 1] -= pad[1]
    boxes[..., 2] -= pad[0]
    boxes[..., 3] -= pad[1]
    boxes[..., :4] /= gain
    return clip_boxes(boxes, img0_shape)
