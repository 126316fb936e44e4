"""bbox_iou with the reference's signature (ultralytics/utils/metrics.py:171-234): IoU / GIoU / DIoU / CIoU.
Gotcha kept: in xyxy mode h gets +eps, w does not; union += eps; CIoU alpha under no_grad."""
from __future__ import annotations

import math

import torch


def bbox_iou(box1, box2, xywh=True, GIoU=False, DIoU=False, CIoU=False, eps=1e-7):
    if xywh:
        (x1, y1, w1, h1), (x2, y2, w2, h2) = box1.chunk(4, -1), box2.chunk(4, -1)
        b1_x1, b1_x2, b1_y1, b1_y2 = x1 - w1 / 2, x1 + w1 / 2, y1 - h1 / 2, y1 + h1 / 2
        b2_x1, b2_x2, b2_y1, b2_y2 = x2 - w2 / 2, x2 + w2 / 2, y2 - h2 / 2, y2 + h2 / 2
    else:
        b1_x1, b1_y1, b1_x2, b1_y2 = box1.chunk(4, -1)
        b2_x1, b2_y1, b2_x2, b2_y2 = box2.chunk(4, -1)
        w1, h1 = b1_x2 - b1_x1, b1_y2 - b1_y1 + eps
        w2, h2 = b2_x2 - b2_x1, b2_y2 - b2_y1 + eps
    inter = (b1_x2.minimum(b2_x2) - b1_x1.maximum(b2_x1)).clamp_(0) * (b1_y2.minimum(b2_y2) - b1_y1.maximum(b2_y1)).clamp_(0)
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    if CIoU or DIoU or GIoU:
        cw = b1_x2.maximum(b2_x2) - b1_x1.minimum(b2_x1)
        ch = b1_y2.maximum(b2_y2) - b1_y1.minimum(b2_y1)
        if CIoU or DIoU:
            c2 = cw.pow(2) + ch.pow(2) + eps
            rho2 = ((b2_x1 + b2_x2 - b1_x1 - b1_x2).pow(2) + (b2_y1 + b2_y2 - b1_y1 - b1_y2).pow(2)) / 4
            if CIoU:
                v = (4 / math.pi**2) * ((w2 / h2).atan() - (w1 / h1).atan()).pow(2)
                with torch.no_grad():
                    alpha = v / (v - iou + (1 + eps))
                return iou - (rho2 / c2 + v * alpha)
            return iou - rho2 / c2
        c_area = cw * ch + eps
        return iou - (c_area - union) / c_area
    return iou


def box_iou(box1, box2, eps=1e-7):
    """(N,4) x (M,4) xyxy -> (N,M) IoU (metrics.py:52-72)."""
    (a1, a2), (b1, b2) = box1.float().unsqueeze(1).chunk(2, 2), box2.float().unsqueeze(0).chunk(2, 2)
    inter = (torch.min(a2, b2) - torch.max(a1, b1)).clamp_(0).prod(2)
    return inter / ((a2 - a1).prod(2) + (b2 - b1).prod(2) - inter + eps)


# ------------------------------------------------------------------------------------------------ validation metrics
# The reference's mAP machinery (utils/metrics.py:547-552 smooth, :605-634 compute_ap, :637-725 ap_per_class, :728-851
# Metric, :898-1000 DetMetrics).  box_iou runs on the device through the C-ABI; the PR-curve arithmetic is host-side
# numpy exactly as in the reference (it runs once per validation pass over a few thousand rows).
import numpy as np  # noqa: E402


def box_iou_device(box1: torch.Tensor, box2: torch.Tensor, eps: float = 1e-7) -> torch.Tensor:
    """box_iou (metrics.py:52-72) through sy11_box_iou: (N,4) x (M,4) xyxy on the GPU -> (N,M) f32, bit-identical to the
    reference's evaluation order."""
    from .. import ops
    a, b = box1.float().contiguous(), box2.float().contiguous()
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    ops._need_gpu(a, b)
    ops.call("sy11_box_iou", a.shape[0], b.shape[0], ops._p(a), ops._p(b), float(eps), ops._p(out), ops._stream())
    return out


def smooth(y, f=0.05):
    """Box filter of fraction f with edge replication (metrics.py:547-552)."""
    nf = round(len(y) * f * 2) // 2 + 1
    pad = np.ones(nf // 2)
    return np.convolve(np.concatenate((pad * y[0], y, pad * y[-1])), np.ones(nf) / nf, mode="valid")


_trapz = getattr(np, "trapezoid", None) or np.trapz


def compute_ap(recall, precision):
    """101-point interpolated area under the monotone precision envelope, sentinels (0,1) and (1,0) (metrics.py:605-634)."""
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([1.0], precision, [0.0]))
    mpre = np.flip(np.maximum.accumulate(np.flip(mpre)))
    x = np.linspace(0, 1, 101)
    return _trapz(np.interp(x, mrec, mpre), x), mpre, mrec


def ap_per_class(tp, conf, pred_cls, target_cls, eps=1e-16):
    """Per-class AP at every IoU threshold plus the P / R / F1 curves (metrics.py:637-725, plotting dropped).
    Returns (tp, fp, p, r, f1, ap, unique_classes, p_curve, r_curve, f1_curve, x, prec_values)."""
    order = np.argsort(-conf)
    tp, conf, pred_cls = tp[order], conf[order], pred_cls[order]
    classes, n_targets = np.unique(target_cls, return_counts=True)
    nc = classes.shape[0]
    x, prec_values = np.linspace(0, 1, 1000), []
    ap = np.zeros((nc, tp.shape[1]))
    p_curve, r_curve = np.zeros((nc, 1000)), np.zeros((nc, 1000))
    for ci, c in enumerate(classes):
        sel = pred_cls == c
        n_l, n_p = n_targets[ci], sel.sum()
        if n_p == 0 or n_l == 0:
            continue
        fpc = (1 - tp[sel]).cumsum(0)
        tpc = tp[sel].cumsum(0)
        recall = tpc / (n_l + eps)
        r_curve[ci] = np.interp(-x, -conf[sel], recall[:, 0], left=0)      # xp must increase: negate the confidences
        precision = tpc / (tpc + fpc)
        p_curve[ci] = np.interp(-x, -conf[sel], precision[:, 0], left=1)
        for j in range(tp.shape[1]):
            ap[ci, j], mpre, mrec = compute_ap(recall[:, j], precision[:, j])
            if j == 0:
                prec_values.append(np.interp(x, mrec, mpre))
    prec_values = np.array(prec_values)
    f1_curve = 2 * p_curve * r_curve / (p_curve + r_curve + eps)
    i = smooth(f1_curve.mean(0), 0.1).argmax()
    p, r, f1 = p_curve[:, i], r_curve[:, i], f1_curve[:, i]
    tpn = (r * n_targets).round()
    fpn = (tpn / (p + eps) - tpn).round()
    return tpn, fpn, p, r, f1, ap, classes.astype(int), p_curve, r_curve, f1_curve, x, prec_values


class Metric:
    """Per-class P / R / F1 / AP container with the reference's accessors (metrics.py:728-896)."""

    def __init__(self):
        self.p, self.r, self.f1, self.all_ap, self.ap_class_index = [], [], [], [], []
        self.nc = 0

    @property
    def ap50(self):
        return self.all_ap[:, 0] if len(self.all_ap) else []

    @property
    def ap(self):
        return self.all_ap.mean(1) if len(self.all_ap) else []

    @property
    def mp(self):
        return self.p.mean() if len(self.p) else 0.0

    @property
    def mr(self):
        return self.r.mean() if len(self.r) else 0.0

    @property
    def map50(self):
        return self.all_ap[:, 0].mean() if len(self.all_ap) else 0.0

    @property
    def map75(self):
        return self.all_ap[:, 5].mean() if len(self.all_ap) else 0.0

    @property
    def map(self):
        return self.all_ap.mean() if len(self.all_ap) else 0.0

    def mean_results(self):
        return [self.mp, self.mr, self.map50, self.map]

    def class_result(self, i):
        return self.p[i], self.r[i], self.ap50[i], self.ap[i]

    @property
    def maps(self):
        maps = np.zeros(self.nc) + self.map
        for i, c in enumerate(self.ap_class_index):
            maps[c] = self.ap[i]
        return maps

    def fitness(self):
        return (np.array(self.mean_results()) * [0.0, 0.0, 0.1, 0.9]).sum()

    def update(self, results):
        self.p, self.r, self.f1, self.all_ap, self.ap_class_index = results[:5]


class DetMetrics:
    """Detection metrics with the reference's public surface (metrics.py:898-1000): process(), keys, results_dict, fitness."""

    def __init__(self, names=()):
        self.names = names
        self.box = Metric()
        self.speed = {"preprocess": 0.0, "inference": 0.0, "loss": 0.0, "postprocess": 0.0}
        self.task = "detect"

    def process(self, tp, conf, pred_cls, target_cls):
        res = ap_per_class(tp, conf, pred_cls, target_cls)[2:]
        self.box.nc = len(self.names)
        self.box.update(res)

    @property
    def keys(self):
        return ["metrics/precision(B)", "metrics/recall(B)", "metrics/mAP50(B)", "metrics/mAP50-95(B)"]

    def mean_results(self):
        return self.box.mean_results()

    def class_result(self, i):
        return self.box.class_result(i)

    @property
    def maps(self):
        return self.box.maps

    @property
    def fitness(self):
        return self.box.fitness()

    @property
    def ap_class_index(self):
        return self.box.ap_class_index

    @property
    def results_dict(self):
        return dict(zip(self.keys + ["fitness"], self.mean_results() + [self.fitness]))
