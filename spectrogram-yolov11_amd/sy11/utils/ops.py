"""Box utilities and NMS under the reference's names (ultralytics/utils/ops.py: scale_boxes :92-127, make_divisible :130-143,
non_max_suppression :181-332, clip_boxes :335-354, xyxy2xywh :412-429, xywh2xyxy :432-449).

``non_max_suppression`` is re-cut for the device: candidate selection (r04: a HIP kernel pair straight over the (B, 4 + nc, A) tensor
Detect wrote — ``sy11_nms_candidates`` — and ONE stable sort of 64-bit (image, score) keys; the tensor-op form of r03 remains for
apriori labels / class filters / CPU tensors), class offsets and the `max_nms` / `max_det` truncations are done ONCE for the whole
batch (one host read of the per-image candidate counts instead of a Python loop with several reads per image); the greedy
suppression itself runs for all images side by side on the HIP
bit-matrix kernels (``sy11_nms_sorted_batched``) that stands where the reference calls ``torchvision.ops.nms`` (ops.py:312), fed in
(score descending, index ascending) order so the kept set is bit-exact.  The reference's wall-clock break (ops.py:328-330) is
intentionally absent: results never depend on time.
"""
from __future__ import annotations

import math

import torch

from .. import ops as _k


def make_divisible(x, divisor):
    """Smallest multiple of ``divisor`` (an int or a stride tensor: its maximum) that is >= x."""
    d = int(divisor.max()) if isinstance(divisor, torch.Tensor) else divisor
    return d * math.ceil(x / d)


def _need_boxes(x):
    if x.shape[-1] != 4:
        raise AssertionError(f"input shape last dimension expected 4 but input shape is {x.shape}")


def xywh2xyxy(x):
    """(cx, cy, w, h) -> (x1, y1, x2, y2), last dimension 4; a new tensor."""
    _need_boxes(x)
    centre, half = x[..., :2], x[..., 2:] / 2
    return torch.cat((centre - half, centre + half), -1)


def xyxy2xywh(x):
    """(x1, y1, x2, y2) -> (cx, cy, w, h), last dimension 4; a new tensor."""
    _need_boxes(x)
    lo, hi = x[..., :2], x[..., 2:]
    return torch.cat(((lo + hi) / 2, hi - lo), -1)


def clip_boxes(boxes, shape):
    """Clamp xyxy boxes into an image of ``shape`` = (h, w), in place: x columns to [0, w], y columns to [0, h]."""
    h, w = shape[0], shape[1]
    boxes[..., 0:4:2].clamp_(0, w)          # strided views of columns (0, 2) and (1, 3): in place, no gather
    boxes[..., 1:4:2].clamp_(0, h)
    return boxes


def scale_boxes(img1_shape, boxes, img0_shape, ratio_pad=None, padding=True, xywh=False):
    """Boxes of the letterboxed ``img1_shape`` back to the original ``img0_shape``, in place: remove the border, undo the gain, clip."""
    if ratio_pad is not None:
        gain, (pad_x, pad_y) = ratio_pad[0][0], ratio_pad[1]
    else:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad_x = round((img1_shape[1] - img0_shape[1] * gain) / 2 - 0.1)
        pad_y = round((img1_shape[0] - img0_shape[0] * gain) / 2 - 0.1)
    if padding:
        last = 2 if xywh else 4             # xywh: only the centre moves
        boxes[..., 0:last:2] -= pad_x
        boxes[..., 1:last:2] -= pad_y
    boxes[..., :4] /= gain
    return clip_boxes(boxes, img0_shape)


def nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    """Drop-in for torchvision.ops.nms: kept indices (int64) in descending-score order; ties -> lower index first."""
    if boxes.shape[0] == 0:
        return torch.zeros((0,), dtype=torch.int64, device=boxes.device)
    order = torch.sort(scores, descending=True, stable=True).indices
    keep = _k.nms_sorted(boxes[order].contiguous(), float(iou_threshold))
    return order[keep]


def _suppress(boxes_sorted, scores_sorted, counts, iou_threshold, max_keep):
    """Greedy suppression of a batch: image i owns counts[i] consecutive rows, already in (score descending, candidate order) order
    -> bool keep mask with at most ``max_keep`` survivors per image (ONE launch pair for all images: sy11_nms_sorted_batched).
    (``scores_sorted`` is not needed by the kernel: the order carries it.  Kept in the signature for the parity tests, which
    intercept this call to compare the rows with what the reference hands to torchvision.ops.nms.)"""
    return _k.nms_sorted_batched(boxes_sorted, counts, float(iou_threshold), int(max_keep))


def _prior_rows(labels, nc, nm, dtype, device):
    """The reference's apriori-label rows (ops.py:272-278): one (4 + nc + nm) row per given label — its box in xyxy, a one-hot
    class vector — appended after an image's predictions.  -> (image index, rows) or None."""
    img, rows = [], []
    for xi, lb in enumerate(labels or ()):
        if lb is None or len(lb) == 0:
            continue
        lb = torch.as_tensor(lb, dtype=dtype, device=device)
        v = torch.zeros((lb.shape[0], 4 + nc + nm), dtype=dtype, device=device)
        v[:, :4] = xywh2xyxy(lb[:, 1:5])
        v[torch.arange(lb.shape[0], device=device), lb[:, 0].long() + 4] = 1.0
        img.append(torch.full((lb.shape[0],), xi, dtype=torch.long, device=device))
        rows.append(v)
    return (torch.cat(img), torch.cat(rows)) if rows else None


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False,
                        labels=(), max_det=300, nc=0, max_time_img=0.05, max_nms=30000, max_wh=7680, in_place=True,
                        rotated=False, end2end=False):
    """(B, 4 + nc + nm, A) predictions -> list of B tensors (n_i, 6 + nm) [x1, y1, x2, y2, score, class, mask coefficients],
    rows in descending-score order, at most ``max_det`` per image.  Same arguments and results as the reference's."""
    for name, v in (("Confidence threshold", conf_thres), ("IoU", iou_thres)):
        if not 0 <= v <= 1:
            raise AssertionError(f"Invalid {name} {v}, valid values are between 0.0 and 1.0")
    if rotated:
        raise NotImplementedError("rotated boxes are outside the hot path")
    if isinstance(prediction, (list, tuple)):                 # (inference output, raw maps)
        prediction = prediction[0]
    dev = prediction.device
    wanted = None if classes is None else torch.as_tensor(classes, device=dev)
    if prediction.shape[-1] == 6 or end2end:                  # already one row per detection: threshold only
        out = []
        for rows in prediction:
            rows = rows[rows[:, 4] > conf_thres][:max_det]
            out.append(rows if wanted is None else rows[(rows[:, 5:6] == wanted).any(1)])
        return out

    B, D, A = prediction.shape
    nc = nc or D - 4
    nm = D - 4 - nc
    multi_label = bool(multi_label) and nc > 1
    prior = _prior_rows(labels, nc, nm, prediction.dtype, dev)
    if (prior is None and wanted is None and prediction.is_cuda and prediction.dtype == torch.float32 and prediction.is_contiguous()
            and conf_thres >= 0):
        # ---- device path (r04): thresholding, compaction and the candidate order come from ONE pass over the tensor as Detect wrote it
        # (sy11_nms_candidates); the per-image score order is ONE stable sort of 64-bit keys (image << 32 | ~bits(score)) instead of
        # two sorts + gathers; no (B, A, D) transposed copy, no boolean mask, no torch.nonzero over B x A x nc scores
        half = prediction[:, 2:4] / 2
        xy = torch.cat((prediction[:, 0:2] - half, prediction[:, 0:2] + half), 1)       # (B, 4, A): xywh2xyxy, element for element
        if in_place:
            prediction[:, :4] = xy                            # the caller's tensor holds corner boxes afterwards, as in the reference
        # one segment per (image, class) unless agnostic: shifted by class * max_wh, boxes of different classes never intersect, so
        # the image-wide greedy sweep keeps exactly what per-class sweeps keep — at n^2 / (2 nc) instead of n^2 / 2 IoU tests.  That
        # holds while no box reaches across a class's max_wh-wide band: checked on the device (x extent of ALL boxes < max_wh; Detect's
        # decoded boxes span at most the image + 2 x 15 strides), read back with the candidate counts; otherwise one sweep per image
        span_ok = (xy[:, 2].amax() - xy[:, 0].amin()) < max_wh
        key, anchor, cidx, counts, by_class = _k.nms_candidates(prediction, nc, conf_thres, multi_label, not agnostic, max_nms, span_ok)
        order = torch.sort(key, stable=True).indices
        key = key[order]
        seg = key >> 32
        conf = ((~key) & 0xFFFFFFFF).to(torch.int32).view(torch.float32)
        a_idx, c = anchor[order].long(), cidx[order].long()
        img = seg // nc if by_class else seg
        if not by_class and counts and max(counts) > max_nms:  # keep each image's max_nms best (the keys are per image here)
            counts_t = torch.tensor(counts, device=dev)
            starts = torch.cumsum(counts_t, 0) - counts_t
            sel = (torch.arange(img.numel(), device=dev) - starts[img]) < max_nms
            a_idx, c, conf, img, order = a_idx[sel], c[sel], conf[sel], img[sel], order[sel]
            counts = [min(n, max_nms) for n in counts]
        box = xy[img, :, a_idx]
        cls_f = c.to(prediction.dtype)
        shifted = box if agnostic else box + (cls_f * max_wh).unsqueeze(1)
        if by_class:
            keep = _k.nms_sorted_segments(shifted.contiguous(), seg, B * nc, iou_thres, max_det)
            kept = torch.nonzero(keep, as_tuple=True)[0]
            # back to the reference's order — per image, score descending, ties in candidate order: survivors into candidate order
            # (their positions in the unsorted candidate list), then one stable sort by (image, score)
            kept = kept[torch.sort(order[kept]).indices]
            kept = kept[torch.sort((img[kept] << 32) | (key[kept] & 0xFFFFFFFF), stable=True).indices]
        else:
            keep = _suppress(shifted.contiguous(), conf, counts, iou_thres, max_det)
            kept = torch.nonzero(keep, as_tuple=True)[0]
        kimg = img[kept]
        kcount = torch.bincount(kimg, minlength=B)
        krank = torch.arange(kept.numel(), device=dev) - (torch.cumsum(kcount, 0) - kcount)[kimg]
        kept = kept[krank < max_det]
        final = torch.cat((box[kept], conf[kept, None], cls_f[kept, None], prediction[img[kept], 4 + nc:, a_idx[kept]]), 1)
        sizes = torch.bincount(img[kept], minlength=B).tolist()    # host read: survivors per image
        return list(torch.split(final, sizes))
    box, c, conf, img, counts, extra, dtype = _candidates_by_tensor_ops(prediction, prior, wanted, B, D, A, nc, nm, conf_thres, multi_label,
                                                                      in_place, max_nms)
    cls_f = c.to(dtype)
    shifted = box if agnostic else box + (cls_f * max_wh).unsqueeze(1)      # classes never overlap: one NMS for all of them

    # ---- greedy suppression, all images side by side (HIP), then the first max_det survivors of each image
    keep = _suppress(shifted.contiguous(), conf, counts, iou_thres, max_det)
    kept = torch.nonzero(keep, as_tuple=True)[0]
    kimg = img[kept]
    kcount = torch.bincount(kimg, minlength=B)
    krank = torch.arange(kept.numel(), device=dev) - (torch.cumsum(kcount, 0) - kcount)[kimg]
    kept = kept[krank < max_det]                               # (the kernel already stops at max_det survivors; a stub may not)
    final = torch.cat((box[kept], conf[kept, None], cls_f[kept, None], extra(kept)), 1)
    sizes = torch.bincount(img[kept], minlength=B).tolist()    # host read 2 of 2: survivors per image
    return list(torch.split(final, sizes))


def _candidates_by_tensor_ops(prediction, prior, wanted, B, D, A, nc, nm, conf_thres, multi_label, in_place, max_nms):
    """The r03 candidate selection with tensor operations: taken with apriori `labels`, a `classes` filter, CPU tensors or a
    non-f32 / non-contiguous prediction.  -> (box (n, 4), class index, score, image index, per-image counts, mask-coefficient getter, dtype)."""
    dev = prediction.device
    rows = prediction.transpose(1, 2)                         # (B, A, D) view
    xyxy = xywh2xyxy(rows[..., :4])
    if in_place:
        rows[..., :4] = xyxy                                  # the caller's tensor holds corner boxes afterwards, as in the reference
    rows = torch.cat((xyxy, rows[..., 4:]), -1).reshape(B * A, D)
    img_of = torch.arange(B, device=dev).repeat_interleave(A)
    if prior is not None:                                     # an image's apriori rows follow its predictions (stable sort below)
        img_of = torch.cat((img_of, prior[0]))
        rows = torch.cat((rows, prior[1]))
        order = torch.sort(img_of, stable=True).indices
        img_of, rows = img_of[order], rows[order]

    # ---- candidates, in the reference's order: image, then anchor, then (multi-label) class
    scores = rows[:, 4:4 + nc]
    if multi_label:
        r, c = torch.nonzero(scores > conf_thres, as_tuple=True)
        conf = scores[r, c]
    else:
        conf, c = scores.max(1)
        r = torch.nonzero(conf > conf_thres, as_tuple=True)[0]
        conf, c = conf[r], c[r]
    if wanted is not None:
        sel = (c.unsqueeze(1) == wanted.view(1, -1)).any(1)
        r, c, conf = r[sel], c[sel], conf[sel]
    img = img_of[r]

    # ---- per image: descending score, ties by candidate order (what a stable sort of each image's rows gives; the reference's
    # `argsort(descending=True)[:max_nms]` followed by nms's own ordering selects and orders the same rows)
    by_score = torch.sort(conf, descending=True, stable=True).indices
    by_image = torch.sort(img[by_score], stable=True).indices
    order = by_score[by_image]
    r, c, conf, img = r[order], c[order], conf[order], img[order]
    counts = torch.bincount(img, minlength=B).tolist()         # host read 1 of 2: candidates per image
    if counts and max(counts) > max_nms:                       # keep each image's max_nms best
        counts_t = torch.tensor(counts, device=dev)
        starts = torch.cumsum(counts_t, 0) - counts_t
        sel = (torch.arange(img.numel(), device=dev) - starts[img]) < max_nms
        r, c, conf, img = r[sel], c[sel], conf[sel], img[sel]
        counts = [min(n, max_nms) for n in counts]
    box = rows[r, :4]
    return box, c, conf, img, counts, (lambda kept: rows[r[kept], 4 + nc:]), rows.dtype

