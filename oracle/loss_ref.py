"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product package).

CPU restatement (plain PyTorch fp32) of the reference's detection criterion and its helpers.

Reference lines followed (under /root/reference/ultralytics):
  utils/loss.py:65-88    DFLoss            utils/loss.py:91-128   BboxLoss
  utils/loss.py:172-275  v8DetectionLoss   utils/tal.py:14-296    TaskAlignedAssigner
  utils/tal.py:361-364   bbox2dist         utils/metrics.py:171-234 bbox_iou (CIoU branch)
Numerical gotchas kept: SURVEY.md §8(g) items 9, 10, 11.

Parity status: PINNED against reference outputs in tests/golden/ (loss, items, TAL targets, grads).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .yolo11_ref import REG_MAX, dist2bbox, make_anchors


def ciou(b1, b2, eps=1e-7):
    """bbox_iou(xywh=False, CIoU=True): h gets +eps, w does not; alpha under no_grad."""
    b1x1, b1y1, b1x2, b1y2 = b1.chunk(4, -1)
    b2x1, b2y1, b2x2, b2y2 = b2.chunk(4, -1)
    w1, h1 = b1x2 - b1x1, b1y2 - b1y1 + eps
    w2, h2 = b2x2 - b2x1, b2y2 - b2y1 + eps
    inter = (torch.minimum(b1x2, b2x2) - torch.maximum(b1x1, b2x1)).clamp(min=0) * \
            (torch.minimum(b1y2, b2y2) - torch.maximum(b1y1, b2y1)).clamp(min=0)
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    cw = torch.maximum(b1x2, b2x2) - torch.minimum(b1x1, b2x1)
    ch = torch.maximum(b1y2, b2y2) - torch.minimum(b1y1, b2y1)
    c2 = cw ** 2 + ch ** 2 + eps
    rho2 = ((b2x1 + b2x2 - b1x1 - b1x2) ** 2 + (b2y1 + b2y2 - b1y1 - b1y2) ** 2) / 4
    v = (4 / math.pi ** 2) * (torch.atan(w2 / h2) - torch.atan(w1 / h1)) ** 2
    with torch.no_grad():
        alpha = v / (v - iou + (1 + eps))
    return iou - (rho2 / c2 + v * alpha)


def bbox2dist(anchors, bbox, reg_max):
    x1y1, x2y2 = bbox.chunk(2, -1)
    return torch.cat((anchors - x1y1, x2y2 - anchors), -1).clamp(0, reg_max - 0.01)


@torch.no_grad()
def tal_assign(pd_scores, pd_bboxes, anc_points, gt_labels, gt_bboxes, mask_gt,
               topk=10, nc=80, alpha=0.5, beta=6.0, eps=1e-9):
    """TaskAlignedAssigner._forward.  Returns (target_labels, target_bboxes, target_scores, fg_mask, gt_idx)."""
    bs, na, _ = pd_scores.shape
    nmax = gt_bboxes.shape[1]
    if nmax == 0:
        return (torch.full_like(pd_scores[..., 0], nc), torch.zeros_like(pd_bboxes), torch.zeros_like(pd_scores),
                torch.zeros_like(pd_scores[..., 0]), torch.zeros_like(pd_scores[..., 0]))
    # anchors whose centre lies strictly inside the gt box (tal.py:242-263)
    lt, rb = gt_bboxes.view(-1, 1, 4).chunk(2, 2)
    deltas = torch.cat((anc_points[None] - lt, rb - anc_points[None]), 2).view(bs, nmax, na, -1)
    mask_in = (deltas.amin(3) > eps).to(gt_bboxes.dtype)
    # alignment metric (tal.py:132-155)
    m = (mask_in * mask_gt).bool()
    overlaps = torch.zeros(bs, nmax, na, dtype=pd_bboxes.dtype)
    bbox_scores = torch.zeros(bs, nmax, na, dtype=pd_scores.dtype)
    bi = torch.arange(bs).view(-1, 1).expand(-1, nmax)
    ci = gt_labels.squeeze(-1).long()
    bbox_scores[m] = pd_scores[bi, :, ci][m]
    pb = pd_bboxes.unsqueeze(1).expand(-1, nmax, -1, -1)[m]
    gb = gt_bboxes.unsqueeze(2).expand(-1, -1, na, -1)[m]
    overlaps[m] = ciou(gb, pb).squeeze(-1).clamp(min=0)
    align = bbox_scores.pow(alpha) * overlaps.pow(beta)
    # top-k per gt (tal.py:162-191)
    _, idx = torch.topk(align, topk, dim=-1, largest=True)
    idx = idx.masked_fill(~mask_gt.expand(-1, -1, topk).bool(), 0)
    count = torch.zeros(align.shape, dtype=torch.int8)
    ones = torch.ones_like(idx[:, :, :1], dtype=torch.int8)
    for k in range(topk):
        count.scatter_add_(-1, idx[:, :, k:k + 1], ones)
    count = count.masked_fill(count > 1, 0).to(align.dtype)
    mask_pos = count * mask_in * mask_gt
    # one gt per anchor (tal.py:266-296)
    fg = mask_pos.sum(-2)
    if fg.max() > 1:
        multi = (fg.unsqueeze(1) > 1).expand(-1, nmax, -1)
        best = overlaps.argmax(1)
        is_max = torch.zeros_like(mask_pos)
        is_max.scatter_(1, best.unsqueeze(1), 1)
        mask_pos = torch.where(multi, is_max, mask_pos).float()
        fg = mask_pos.sum(-2)
    gt_idx = mask_pos.argmax(-2)
    # targets (tal.py:193-239)
    flat_idx = gt_idx + torch.arange(bs)[..., None] * nmax
    t_labels = gt_labels.long().flatten()[flat_idx].clamp(min=0)
    t_boxes = gt_bboxes.view(-1, 4)[flat_idx]
    t_scores = F.one_hot(t_labels, nc)
    t_scores = torch.where(fg[:, :, None] > 0, t_scores, 0)
    # normalise (tal.py:115-120)
    align = align * mask_pos
    pos_align = align.amax(-1, keepdim=True)
    pos_ov = (overlaps * mask_pos).amax(-1, keepdim=True)
    norm = (align * pos_ov / (pos_align + eps)).amax(-2).unsqueeze(-1)
    return t_labels, t_boxes, t_scores * norm, fg.bool(), gt_idx


def pack_targets(batch_idx, cls, bboxes, batch_size, scale):
    """v8DetectionLoss.preprocess: (n,) image ids + cls + normalised xywh -> (B, maxGT, 5) [cls, xyxy·scale]."""
    t = torch.cat((batch_idx.view(-1, 1).float(), cls.view(-1, 1).float(), bboxes.float()), 1)
    if t.shape[0] == 0:
        return torch.zeros(batch_size, 0, 5)
    counts = torch.bincount(t[:, 0].long(), minlength=batch_size)
    out = torch.zeros(batch_size, int(counts.max()), 5)
    for j in range(batch_size):
        rows = t[t[:, 0] == j, 1:]
        out[j, : rows.shape[0]] = rows
    xywh = out[..., 1:5] * scale
    xy, wh = xywh[..., :2], xywh[..., 2:] / 2
    out[..., 1:5] = torch.cat((xy - wh, xy + wh), -1)
    return out


def detection_loss(maps, batch, nc, strides=(8.0, 16.0, 32.0), box=7.5, cls_gain=0.5, dfl=1.5, return_targets=False, pinned=None):
    """v8DetectionLoss.__call__: returns (sum(loss)·B, detached [box, cls, dfl]).
    ``pinned`` = a (t_labels, t_boxes, t_scores, fg, gt_idx) tuple from an earlier call: the task-aligned assignment (a
    no-grad, DISCRETE function of the predictions, loss.py:250-258) is taken from there instead of being recomputed — used
    by the 16-bit parity tests so that both sides differentiate the same smooth function."""
    B = maps[0].shape[0]
    no = nc + 4 * REG_MAX
    cat = torch.cat([m.reshape(B, no, -1) for m in maps], 2)
    pred_dist, pred_scores = cat.split((4 * REG_MAX, nc), 1)
    pred_scores = pred_scores.permute(0, 2, 1).contiguous()
    pred_dist = pred_dist.permute(0, 2, 1).contiguous()
    imgsz = torch.tensor(maps[0].shape[2:], dtype=torch.float32) * strides[0]
    anchors, stride_t = make_anchors([m.shape[2:] for m in maps], strides)
    targets = pack_targets(batch["batch_idx"], batch["cls"], batch["bboxes"], B, imgsz[[1, 0, 1, 0]])
    gt_labels, gt_bboxes = targets.split((1, 4), 2)
    mask_gt = (gt_bboxes.sum(2, keepdim=True) > 0).float()
    a = pred_dist.shape[1]
    proj = torch.arange(REG_MAX, dtype=torch.float32)
    pred_ltrb = pred_dist.view(B, a, 4, REG_MAX).softmax(3).matmul(proj)
    pred_bboxes = dist2bbox(pred_ltrb, anchors, xywh=False)
    if pinned is not None:
        t_labels, t_boxes, t_scores, fg, gt_idx = pinned
    else:
        t_labels, t_boxes, t_scores, fg, gt_idx = tal_assign(
            pred_scores.detach().sigmoid(), (pred_bboxes.detach() * stride_t), anchors * stride_t,
            gt_labels, gt_bboxes, mask_gt, nc=nc)
    tss = max(t_scores.sum(), 1)
    loss = torch.zeros(3)
    loss[1] = F.binary_cross_entropy_with_logits(pred_scores, t_scores.to(pred_scores.dtype), reduction="none").sum() / tss
    if fg.sum():
        t_boxes_s = t_boxes / stride_t
        w = t_scores.sum(-1)[fg].unsqueeze(-1)
        iou = ciou(pred_bboxes[fg], t_boxes_s[fg])
        loss[0] = ((1.0 - iou) * w).sum() / tss
        tgt = bbox2dist(anchors, t_boxes_s, REG_MAX - 1)[fg].clamp(0, REG_MAX - 1 - 0.01)
        tl = tgt.long()
        tr = tl + 1
        wl = tr - tgt
        wr = 1 - wl
        pd = pred_dist[fg].view(-1, REG_MAX)
        l_dfl = (F.cross_entropy(pd, tl.view(-1), reduction="none").view(tl.shape) * wl
                 + F.cross_entropy(pd, tr.view(-1), reduction="none").view(tl.shape) * wr).mean(-1, keepdim=True)
        loss[2] = (l_dfl * w).sum() / tss
    gains = torch.tensor([box, cls_gain, dfl])
    loss = loss * gains
    if return_targets:
        return loss.sum() * B, loss.detach(), (t_labels, t_boxes, t_scores, fg, gt_idx)
    return loss.sum() * B, loss.detach()
