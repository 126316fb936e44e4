"""Validation metrics under the reference's names (ultralytics/utils/metrics.py: box_iou :52-72, smooth :547-552, compute_ap :605-634,
ap_per_class :637-725, Metric :728-851, DetMetrics :898-1000).

The IoU matrix of a validation image runs on the device (``box_iou_device`` -> sy11_box_iou, bit-identical to the reference's
evaluation order); the precision / recall bookkeeping runs once per validation pass over a few thousand rows and stays in numpy.
(The training criterion's CIoU lives inside csrc/loss.hip; there is no tensor-op ``bbox_iou`` in this package.)
"""
from __future__ import annotations

import numpy as np
import torch

_trapezoid = getattr(np, "trapezoid", None) or np.trapz
_RECALL_GRID = np.linspace(0.0, 1.0, 101)          # COCO's 101 recall points
_CONF_GRID = np.linspace(0.0, 1.0, 1000)           # confidence axis of the P / R / F1 curves


def box_iou(box1, box2, eps=1e-7):
    """(N, 4) x (M, 4) xyxy -> (N, M) IoU, tensor formulation (host-side stand-in of ``box_iou_device``; same rounding steps:
    width * height, area sum minus intersection, plus eps)."""
    a, b = box1.float(), box2.float()
    span = (torch.minimum(a[:, None, 2:], b[None, :, 2:]) - torch.maximum(a[:, None, :2], b[None, :, :2])).clamp_(min=0)
    inter = span[..., 0] * span[..., 1]
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / (area_a[:, None] + area_b[None, :] - inter + eps)


def box_iou_device(box1: torch.Tensor, box2: torch.Tensor, eps: float = 1e-7) -> torch.Tensor:
    """box_iou through sy11_box_iou: (N, 4) x (M, 4) xyxy on the GPU -> (N, M) f32."""
    from .. import ops
    a, b = box1.float().contiguous(), box2.float().contiguous()
    ops._need_gpu(a, b)
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    ops.call("sy11_box_iou", a.shape[0], b.shape[0], ops._p(a), ops._p(b), float(eps), ops._p(out), ops._stream())
    return out


def smooth(y, f=0.05):
    """Moving average over a window of about ``f`` of the curve (odd length), the ends extended by their edge values."""
    width = round(len(y) * f * 2) // 2 + 1
    edge = width // 2
    padded = np.concatenate((np.full(edge, y[0], dtype=float), y, np.full(edge, y[-1], dtype=float)))
    return np.convolve(padded, np.full(width, 1.0 / width), mode="valid")


def compute_ap(recall, precision):
    """Average precision of one PR curve: sentinels (0, 1) / (1, 0), the monotone (right-to-left running maximum) precision
    envelope, area under its 101-point interpolation.  -> (ap, envelope, recall with sentinels)."""
    rec = np.concatenate(([0.0], recall, [1.0]))
    env = np.maximum.accumulate(np.concatenate(([1.0], precision, [0.0]))[::-1])[::-1]
    return _trapezoid(np.interp(_RECALL_GRID, rec, env), _RECALL_GRID), env, rec


def _class_curves(tp_c, conf_c, n_labels, eps):
    """One class: cumulative TP / FP over its detections (already in descending confidence) -> recall (n, T), precision (n, T)
    and both sampled on the confidence grid at the first IoU threshold."""
    hits = tp_c.cumsum(0)
    misses = (1 - tp_c).cumsum(0)
    recall = hits / (n_labels + eps)
    precision = hits / (hits + misses)
    # np.interp wants increasing abscissae: walk the confidences negated
    r_grid = np.interp(-_CONF_GRID, -conf_c, recall[:, 0], left=0)
    p_grid = np.interp(-_CONF_GRID, -conf_c, precision[:, 0], left=1)
    return recall, precision, r_grid, p_grid


def ap_per_class(tp, conf, pred_cls, target_cls, eps=1e-16):
    """AP per class and IoU threshold plus the P / R / F1 operating point at the F1-optimal confidence (plotting dropped).
    tp (n, T) bool, conf (n,), pred_cls (n,), target_cls (m,).
    -> (tp count, fp count, p, r, f1, ap (classes, T), classes, p_curve, r_curve, f1_curve, confidence grid, PR curves at T0)."""
    rank = np.argsort(-conf)
    tp, conf, pred_cls = tp[rank], conf[rank], pred_cls[rank]
    classes, n_labels = np.unique(target_cls, return_counts=True)
    n_cls, n_thr = classes.shape[0], tp.shape[1]
    ap = np.zeros((n_cls, n_thr))
    p_curve = np.zeros((n_cls, _CONF_GRID.size))
    r_curve = np.zeros_like(p_curve)
    pr_at_t0 = []
    for k in range(n_cls):
        mine = pred_cls == classes[k]
        if not mine.any() or n_labels[k] == 0:
            continue
        recall, precision, r_curve[k], p_curve[k] = _class_curves(tp[mine], conf[mine], n_labels[k], eps)
        for t in range(n_thr):
            ap[k, t], env, rec = compute_ap(recall[:, t], precision[:, t])
            if t == 0:
                pr_at_t0.append(np.interp(_CONF_GRID, rec, env))
    f1_curve = 2 * p_curve * r_curve / (p_curve + r_curve + eps)
    best = smooth(f1_curve.mean(0), 0.1).argmax()                 # one confidence for all classes: the smoothed mean-F1 peak
    p, r, f1 = p_curve[:, best], r_curve[:, best], f1_curve[:, best]
    n_tp = (r * n_labels).round()
    n_fp = (n_tp / (p + eps) - n_tp).round()
    return n_tp, n_fp, p, r, f1, ap, classes.astype(int), p_curve, r_curve, f1_curve, _CONF_GRID, np.array(pr_at_t0)


class Metric:
    """Per-class precision / recall / F1 / AP of one task with the reference's accessor names."""

    _FITNESS_WEIGHTS = np.array([0.0, 0.0, 0.1, 0.9])     # (P, R, mAP@0.5, mAP@0.5:0.95)

    def __init__(self):
        self.p = self.r = self.f1 = self.all_ap = self.ap_class_index = ()
        self.nc = 0

    def update(self, results):
        """results = (p, r, f1, all_ap, ap_class_index, ...) as ap_per_class()[2:] delivers them."""
        self.p, self.r, self.f1, self.all_ap, self.ap_class_index = results[:5]

    def _ap_columns(self, cols=None):
        """Mean over classes of the AP columns ``cols`` (None: all thresholds); 0.0 before the first update."""
        if not len(self.all_ap):
            return 0.0
        return float(np.mean(self.all_ap if cols is None else self.all_ap[:, cols]))

    # per class
    ap50 = property(lambda self: self.all_ap[:, 0] if len(self.all_ap) else [])
    ap = property(lambda self: self.all_ap.mean(1) if len(self.all_ap) else [])
    # means over classes
    mp = property(lambda self: float(np.mean(self.p)) if len(self.p) else 0.0)
    mr = property(lambda self: float(np.mean(self.r)) if len(self.r) else 0.0)
    map50 = property(lambda self: self._ap_columns(0))
    map75 = property(lambda self: self._ap_columns(5))
    map = property(lambda self: self._ap_columns())

    def mean_results(self):
        return [self.mp, self.mr, self.map50, self.map]

    def class_result(self, i):
        return self.p[i], self.r[i], self.ap50[i], self.ap[i]

    @property
    def maps(self):
        """mAP@0.5:0.95 per class id: the overall value where a class had no labels."""
        out = np.full(self.nc, self.map, dtype=float)
        if len(self.ap_class_index):
            out[np.asarray(self.ap_class_index, int)] = self.ap
        return out

    def fitness(self):
        return float((np.asarray(self.mean_results()) * self._FITNESS_WEIGHTS).sum())


class DetMetrics:
    """Detection metrics facade: ``process`` the accumulated statistics, then read ``results_dict`` / ``fitness`` / ``maps``."""

    keys = ["metrics/precision(B)", "metrics/recall(B)", "metrics/mAP50(B)", "metrics/mAP50-95(B)"]
    task = "detect"

    def __init__(self, names=()):
        self.names = names
        self.box = Metric()
        self.speed = dict.fromkeys(("preprocess", "inference", "loss", "postprocess"), 0.0)

    def process(self, tp, conf, pred_cls, target_cls):
        self.box.nc = len(self.names)
        self.box.update(ap_per_class(tp, conf, pred_cls, target_cls)[2:])

    def mean_results(self):
        return self.box.mean_results()

    def class_result(self, i):
        return self.box.class_result(i)

    maps = property(lambda self: self.box.maps)
    fitness = property(lambda self: self.box.fitness())
    ap_class_index = property(lambda self: self.box.ap_class_index)

    @property
    def results_dict(self):
        values = self.mean_results() + [self.fitness]
        return dict(zip(self.keys + ["fitness"], values))
