"""Detection criterion with the reference's names (ultralytics/utils/loss.py: DFLoss :65-88, BboxLoss :91-128,
v8DetectionLoss :172-275).  Consumes the engine's raw maps: NCHW-shaped f32 tensors whose MEMORY is NHWC, so
``(B, H*W, no)`` is a free view (the reference's cat + permute + contiguous round trip disappears)."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .metrics import bbox_iou
from .ops import xywh2xyxy
from .tal import TaskAlignedAssigner, bbox2dist, dist2bbox, make_anchors


class DFLoss(nn.Module):
    def __init__(self, reg_max=16) -> None:
        super().__init__()
        self.reg_max = reg_max

    def __call__(self, pred_dist, target):
        target = target.clamp_(0, self.reg_max - 1 - 0.01)
        tl = target.long()
        tr = tl + 1
        wl = tr - target
        wr = 1 - wl
        return (F.cross_entropy(pred_dist, tl.view(-1), reduction="none").view(tl.shape) * wl
                + F.cross_entropy(pred_dist, tr.view(-1), reduction="none").view(tl.shape) * wr).mean(-1, keepdim=True)


class BboxLoss(nn.Module):
    def __init__(self, reg_max=16):
        super().__init__()
        self.dfl_loss = DFLoss(reg_max) if reg_max > 1 else None

    def forward(self, pred_dist, pred_bboxes, anchor_points, target_bboxes, target_scores, target_scores_sum, fg_mask):
        weight = target_scores.sum(-1)[fg_mask].unsqueeze(-1)
        iou = bbox_iou(pred_bboxes[fg_mask], target_bboxes[fg_mask], xywh=False, CIoU=True)
        loss_iou = ((1.0 - iou) * weight).sum() / target_scores_sum
        if self.dfl_loss:
            target_ltrb = bbox2dist(anchor_points, target_bboxes, self.dfl_loss.reg_max - 1)
            loss_dfl = self.dfl_loss(pred_dist[fg_mask].view(-1, self.dfl_loss.reg_max), target_ltrb[fg_mask]) * weight
            loss_dfl = loss_dfl.sum() / target_scores_sum
        else:
            loss_dfl = torch.tensor(0.0).to(pred_dist.device)
        return loss_iou, loss_dfl


class _FusedLossFn(torch.autograd.Function):
    """Whole criterion as 5 HIP launches forward + 1 backward (csrc/loss.hip); no host synchronisation."""

    @staticmethod
    def forward(ctx, crit, gt, *feats):
        from .. import ops as K
        maps = [f.permute(0, 2, 3, 1) for f in feats]
        maps = [m if m.is_contiguous() else m.contiguous() for m in maps]
        w = K.det_loss_forward(maps, crit.stride_f, crit.nc, gt)
        tot = w.sums.sum(0)                                     # [tss, box, cls, dfl]
        tss = tot[0].clamp(min=1.0)
        items = tot[1:4] / tss
        gains = torch.tensor([crit.hyp.box, crit.hyp.cls, crit.hyp.dfl], dtype=torch.float32).to(items.device, non_blocking=True) \
            if crit._gains is None else crit._gains
        crit._gains = gains
        items = items * gains
        ctx.w, ctx.tss, ctx.crit = w, tss, crit
        loss = items.sum() * feats[0].shape[0]
        ctx.mark_non_differentiable(items)
        return loss, items

    @staticmethod
    def backward(ctx, gloss, gitems):
        from .. import ops as K
        crit = ctx.crit
        up = (gloss.float() / ctx.tss).reshape(1).contiguous()
        dmaps = K.det_loss_backward(ctx.w, up, (crit.hyp.box, crit.hyp.cls, crit.hyp.dfl), out=crit.__dict__.get("grad_out"))
        return (None, None, *[d.permute(0, 3, 1, 2) for d in dmaps])


class v8DetectionLoss:
    """box (CIoU) + cls (BCE) + dfl, returned as (sum * batch_size, detached items).

    On the MI355X the criterion runs as fused HIP kernels (``fused=True``, default when the maps are CUDA f32 tensors with
    reg_max 16 and top-k 10); ``fused=False`` keeps the tensor-op formulation of the reference (device agnostic)."""

    def __init__(self, model, tal_topk=10, fused=True):
        self.fused = fused and tal_topk == 10
        self._gains = None
        device = next(model.parameters()).device
        h = model.args
        m = model.model[-1]
        self.bce = nn.BCEWithLogitsLoss(reduction="none")
        self.hyp = h
        self.stride = m.stride
        self.stride_f = [float(v) for v in m.stride]   # host copy, read ONCE: float(cuda_tensor[i]) per step is a device sync
        self.nc = m.nc
        self.no = m.nc + m.reg_max * 4
        self.reg_max = m.reg_max
        self.device = device
        self.use_dfl = m.reg_max > 1
        self.assigner = TaskAlignedAssigner(topk=tal_topk, num_classes=self.nc, alpha=0.5, beta=6.0)
        self.bbox_loss = BboxLoss(m.reg_max).to(device)
        self.proj = torch.arange(m.reg_max, dtype=torch.float, device=device)

    def _max_targets(self, batch_idx, batch_size, counts):
        """Largest number of targets in one image WITHOUT stalling the launch queue: counted on the host when the labels
        still live there (the dataloader case); for device labels the value is cached per (storage, version) — the
        reference's `counts.max()` (utils/loss.py:201) is a device->host read after every forward pass."""
        if not batch_idx.is_cuda:
            return int(torch.bincount(batch_idx.long().view(-1), minlength=batch_size).max())
        key = (batch_idx.data_ptr(), batch_idx._version, batch_idx.numel(), batch_size)
        cache = self.__dict__.setdefault("_max_gt_cache", {})
        if key not in cache:
            if len(cache) > 64:
                cache.clear()
            cache[key] = int(counts.max())                # one synchronising read, first time this label tensor is seen
        return cache[key]

    def preprocess(self, targets, batch_size, scale_tensor, batch_idx=None):
        nl, ne = targets.shape
        if nl == 0:
            return torch.zeros(batch_size, 0, ne - 1, device=self.device)
        i = targets[:, 0].long()
        # per-image target counts by scatter-add: torch.bincount on a device tensor reads max(i) back to size its output,
        # i.e. it stalls the host until everything queued so far (the previous step's backward) has finished
        counts = torch.zeros(batch_size, dtype=torch.long, device=i.device).scatter_add_(0, i, torch.ones_like(i))
        n_max = self._max_targets(batch_idx, batch_size, counts) if batch_idx is not None else int(counts.max())
        out = torch.zeros(batch_size, n_max, ne - 1, device=self.device)
        # rank of each target inside its image (stable order), no per-image host loop
        order = torch.argsort(i, stable=True)
        starts = torch.cumsum(counts, 0) - counts
        rank = torch.empty_like(i)
        rank[order] = torch.arange(nl, device=i.device) - starts[i[order]]
        out[i, rank] = targets[:, 1:]
        out[..., 1:5] = xywh2xyxy(out[..., 1:5].mul_(scale_tensor))
        return out

    def bbox_decode(self, anchor_points, pred_dist):
        if self.use_dfl:
            b, a, c = pred_dist.shape
            # (softmax * proj).sum == softmax @ proj; the elementwise form avoids a (B*A*4, 16) x 16 rocBLAS gemv (1.4 ms)
            pred_dist = (pred_dist.view(b, a, 4, c // 4).softmax(3) * self.proj.type(pred_dist.dtype)).sum(3)
        return dist2bbox(pred_dist, anchor_points, xywh=False)

    def __call__(self, preds, batch):
        feats = preds[1] if isinstance(preds, tuple) else preds
        B = feats[0].shape[0]
        if self.fused and self.reg_max == 16 and all(f.is_cuda and f.dtype == torch.float32 for f in feats):
            hw = tuple(feats[0].shape[2:])
            scales = self.__dict__.setdefault("_scale_cache", {})
            if hw not in scales:                           # (w, h, w, h) in pixels, uploaded once per input size
                imgsz = torch.tensor(hw, dtype=torch.float32) * self.stride_f[0]
                scales[hw] = imgsz[[1, 0, 1, 0]].to(self.device)
            targets = torch.cat((batch["batch_idx"].view(-1, 1), batch["cls"].view(-1, 1), batch["bboxes"]), 1)
            gt = self.preprocess(targets.to(self.device).float(), B, scale_tensor=scales[hw], batch_idx=batch["batch_idx"])
            loss, items = _FusedLossFn.apply(self, gt, *feats)
            return loss, items.detach()
        if self.fused and self.reg_max == 16:
            # never a silent detour: the hot-path criterion is the HIP one; the tensor-op formulation below is an explicit choice
            from .. import _lib
            raise _lib.Sy11Error("v8DetectionLoss: the fused criterion needs CUDA f32 head maps (got "
                                 f"{[(str(f.device), str(f.dtype)) for f in feats]}); construct it with fused=False for the tensor-op formulation")
        loss = torch.zeros(3, device=self.device)
        # (B, no, H, W) with NHWC memory -> (B, H*W, no) without a copy; fall back to permute for NCHW-contiguous input
        flat = [xi.permute(0, 2, 3, 1).reshape(B, -1, self.no) for xi in feats]
        cat = torch.cat(flat, 1).float()
        pred_distri, pred_scores = cat.split((self.reg_max * 4, self.nc), 2)
        pred_scores = pred_scores.contiguous()
        pred_distri = pred_distri.contiguous()
        dtype = pred_scores.dtype
        batch_size = B
        imgsz = torch.tensor(feats[0].shape[2:], device=self.device, dtype=dtype) * self.stride[0]
        anchor_points, stride_tensor = make_anchors(feats, self.stride, 0.5)
        anchor_points, stride_tensor = anchor_points.to(dtype), stride_tensor.to(dtype)
        targets = torch.cat((batch["batch_idx"].view(-1, 1), batch["cls"].view(-1, 1), batch["bboxes"]), 1)
        targets = self.preprocess(targets.to(self.device).float(), batch_size, scale_tensor=imgsz[[1, 0, 1, 0]])
        gt_labels, gt_bboxes = targets.split((1, 4), 2)
        mask_gt = gt_bboxes.sum(2, keepdim=True).gt_(0.0)
        pred_bboxes = self.bbox_decode(anchor_points, pred_distri)
        _, target_bboxes, target_scores, fg_mask, _ = self.assigner(
            pred_scores.detach().sigmoid(), (pred_bboxes.detach() * stride_tensor).type(gt_bboxes.dtype),
            anchor_points * stride_tensor, gt_labels, gt_bboxes, mask_gt)
        target_scores_sum = max(target_scores.sum(), 1)
        loss[1] = self.bce(pred_scores, target_scores.to(dtype)).sum() / target_scores_sum
        if fg_mask.sum():
            target_bboxes /= stride_tensor
            loss[0], loss[2] = self.bbox_loss(pred_distri, pred_bboxes, anchor_points, target_bboxes, target_scores,
                                              target_scores_sum, fg_mask)
        loss[0] *= self.hyp.box
        loss[1] *= self.hyp.cls
        loss[2] *= self.hyp.dfl
        return loss.sum() * batch_size, loss.detach()
