"""Detection criterion with the reference's name and call contract (ultralytics/utils/loss.py: v8DetectionLoss :172-275,
which there drives DFLoss :65-88, BboxLoss :91-128 and utils/tal.py's TaskAlignedAssigner :14-296).  Here the whole
criterion — decode, task-aligned assignment, BCE / CIoU / DFL terms and the gradient w.r.t. the head maps — is the fused
HIP criterion of csrc/loss.hip; there is no tensor-op formulation in the product (the CPU restatement used by the tests is
oracle/loss_ref.py).  Consumes the engine's raw maps: NCHW-shaped f32 tensors whose MEMORY is NHWC."""
from __future__ import annotations

import torch



class _FusedLossFn(torch.autograd.Function):
    """Whole criterion as 6 HIP launches forward (decode, metrics, resolve, norm, terms, finish) + 1 backward (csrc/loss.hip);
    no host synchronisation, no tensor-op glue."""

    @staticmethod
    def forward(ctx, crit, gt, *feats):
        from .. import ops as K
        maps = [f.permute(0, 2, 3, 1) for f in feats]
        maps = [m if m.is_contiguous() else m.contiguous() for m in maps]
        w = K.det_loss_forward(maps, crit.stride_f, crit.nc, gt)
        out = K.det_loss_finish(w, (crit.hyp.box, crit.hyp.cls, crit.hyp.dfl))       # [loss, box, cls, dfl, 1 / max(tss, 1)]
        ctx.w, ctx.inv_tss, ctx.crit = w, out[4:5], crit
        loss, items = out[0], out[1:4]
        ctx.mark_non_differentiable(items)
        return loss, items

    @staticmethod
    def backward(ctx, gloss, gitems):
        from .. import ops as K
        crit = ctx.crit
        up = gloss if gloss.dtype == torch.float32 else gloss.float()
        dmaps = K.det_loss_backward(ctx.w, up.reshape(1), (crit.hyp.box, crit.hyp.cls, crit.hyp.dfl), out=crit.__dict__.get("grad_out"),
                                    inv_tss=ctx.inv_tss)
        return (None, None, *[d.permute(0, 3, 1, 2) for d in dmaps])


class v8DetectionLoss:
    """box (CIoU) + cls (BCE) + dfl, returned as (sum * batch_size, detached items); fused HIP kernels, CUDA f32 maps only."""

    def __init__(self, model, tal_topk=10, fused=True):
        if not fused or tal_topk != 10:
            from .. import _lib
            raise _lib.Sy11Error("v8DetectionLoss: only the fused HIP criterion (top-k 10) exists on the hot path; the CPU "
                                 "restatement lives in oracle/loss_ref.py (test infrastructure)")
        self._gains = None
        device = next(model.parameters()).device
        m = model.model[-1]
        self.hyp = model.args
        self.stride = m.stride
        self.stride_f = [float(v) for v in m.stride]   # host copy, read ONCE: float(cuda_tensor[i]) per step is a device sync
        self.nc = m.nc
        self.no = m.nc + m.reg_max * 4
        self.reg_max = m.reg_max
        self.device = device
        if self.reg_max != 16:
            from .. import _lib
            raise _lib.Sy11Error(f"v8DetectionLoss: the HIP criterion is built for reg_max 16 (got {self.reg_max})")

    def _max_targets(self, batch_idx, counts):
        """Largest number of targets in one image.  Host labels (the dataloader case): counted on the host, no device read.
        Device labels: ONE synchronising read per label tensor OBJECT, remembered through a weak reference — an address /
        version key is not an identity (the caching allocator hands the same address to the next batch's labels, version 0)."""
        import weakref
        if not batch_idx.is_cuda:
            return int(counts.max()) if counts.numel() else 0
        cache = self.__dict__.setdefault("_max_gt_cache", {})
        hit = cache.get(id(batch_idx))
        if hit is not None and hit[0]() is batch_idx and hit[1] == batch_idx._version:
            return hit[2]
        for k in [k for k, v in cache.items() if v[0]() is None]:
            del cache[k]
        n = int(counts.max())
        cache[id(batch_idx)] = (weakref.ref(batch_idx), batch_idx._version, n)
        return n

    def preprocess(self, targets, batch_size, scale_tensor, batch_idx=None):
        """(n, 6) [image, cls, xywh normalised] -> (B, max targets per image, 5) [cls, xyxy pixels], zero padded
        (utils/loss.py:194-207): one HIP launch (sy11_det_loss_pack_targets) instead of the per-image host loop."""
        from .. import ops as K
        nl, ne = targets.shape
        if nl == 0:
            return torch.zeros(batch_size, 0, ne - 1, device=self.device)
        src = batch_idx if batch_idx is not None else targets[:, 0]
        if not src.is_cuda:                                  # labels still on the host: count there, nothing to wait for
            n_max = self._max_targets(src, torch.bincount(src.long().view(-1), minlength=batch_size))
        else:
            # scatter-add, not torch.bincount: bincount on a device tensor reads max(i) back to size its output; the count is
            # taken ONCE per label tensor object (_max_targets)
            hit = self.__dict__.setdefault("_max_gt_cache", {}).get(id(src))
            if hit is not None and hit[0]() is src and hit[1] == src._version:
                n_max = hit[2]
            else:
                i = targets[:, 0].long()
                counts = torch.zeros(batch_size, dtype=torch.long, device=i.device).scatter_add_(0, i, torch.ones_like(i))
                n_max = self._max_targets(src, counts)
        t = targets.to(self.device, non_blocking=True)
        sw, sh = (float(v) for v in scale_tensor[:2]) if not torch.is_tensor(scale_tensor) else self._scale_wh(scale_tensor)
        return K.det_loss_pack_targets(t[:, 0], t[:, 1], t[:, 2:6], batch_size, n_max, (sw, sh))

    def _scale_wh(self, scale_tensor):
        key = (scale_tensor.data_ptr(), scale_tensor._version)
        c = self.__dict__.setdefault("_scale_wh_cache", {})
        if key not in c:
            v = scale_tensor.detach().cpu()
            c[key] = (float(v[0]), float(v[1]))
        return c[key]

    def __call__(self, preds, batch):
        feats = preds[1] if isinstance(preds, tuple) else preds
        B = feats[0].shape[0]
        if not all(f.is_cuda and f.dtype == torch.float32 for f in feats):
            from .. import _lib                              # never a silent detour to a tensor-op formulation
            raise _lib.Sy11Error("v8DetectionLoss: the HIP criterion needs CUDA f32 head maps (got "
                                 f"{[(str(f.device), str(f.dtype)) for f in feats]})")
        hw = tuple(feats[0].shape[2:])
        scale_wh = (hw[1] * self.stride_f[0], hw[0] * self.stride_f[0])         # image (w, h) in pixels: loss.py:233
        parts = (batch["batch_idx"].view(-1, 1), batch["cls"].view(-1, 1), batch["bboxes"])
        # host labels (the dataloader case): one host-side cat + ONE upload; device labels: one cat launch
        targets = torch.cat([p.float() for p in parts], 1)
        gt = self.preprocess(targets, B, scale_tensor=scale_wh, batch_idx=batch["batch_idx"])
        loss, items = _FusedLossFn.apply(self, gt, *feats)
        return loss, items.detach()
