"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product package).

CPU restatement of the reference's validation metrics, written with explicit loops so that it is independent of the
vectorised numpy formulation it checks:
  utils/metrics.py:52-72     box_iou
  engine/validator.py:224-264 match_predictions (use_scipy=False)
  utils/metrics.py:605-634   compute_ap (101-point interpolation)      utils/metrics.py:547-552 smooth
  utils/metrics.py:637-725   ap_per_class                              utils/metrics.py:832-851 mean results / fitness

Parity status: PINNED — tests/golden/metrics.npz holds the reference's own outputs for seeded inputs (made by
oracle/gen_golden_metrics.py, which imports /root/reference in the dev container).
"""
from __future__ import annotations

import numpy as np


def box_iou(a, b, eps=1e-7):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    out = np.zeros((len(a), len(b)), np.float32)
    for i, p in enumerate(a):
        for j, q in enumerate(b):
            w = max(np.float32(min(p[2], q[2]) - max(p[0], q[0])), np.float32(0))
            h = max(np.float32(min(p[3], q[3]) - max(p[1], q[1])), np.float32(0))
            inter = np.float32(w * h)
            ua = np.float32((p[2] - p[0]) * (p[3] - p[1])) + np.float32((q[2] - q[0]) * (q[3] - q[1]))
            out[i, j] = inter / np.float32(np.float32(ua - inter) + np.float32(eps))
    return out


def match_predictions(pred_cls, true_cls, iou, iouv):
    """iou: (labels, detections).  Returns (detections, len(iouv)) bool."""
    iou = np.asarray(iou, np.float32) * (np.asarray(true_cls)[:, None] == np.asarray(pred_cls)[None, :])
    n_det = len(pred_cls)
    correct = np.zeros((n_det, len(iouv)), bool)
    for t, thr in enumerate(iouv):
        cand = [(float(iou[l, d]), l, d) for l in range(iou.shape[0]) for d in range(n_det) if iou[l, d] >= thr]
        if not cand:
            continue
        if len(cand) > 1:
            cand.sort(key=lambda c: -c[0])                   # by IoU, best first (inputs are tie-free)
            best_for_det = {}
            for v, l, d in cand:                              # each detection keeps its best label
                best_for_det.setdefault(d, (v, l, d))
            by_det = [best_for_det[d] for d in sorted(best_for_det)]
            first_for_label = {}
            for v, l, d in by_det:                            # each label keeps the lowest-index detection left
                first_for_label.setdefault(l, (v, l, d))
            cand = list(first_for_label.values())
        for _, _, d in cand:
            correct[d, t] = True
    return correct


def compute_ap(recall, precision):
    mrec = [0.0] + list(recall) + [1.0]
    mpre = [1.0] + list(precision) + [0.0]
    for i in range(len(mpre) - 2, -1, -1):                   # monotone envelope from the right
        mpre[i] = max(mpre[i], mpre[i + 1])
    xs = np.linspace(0, 1, 101)
    ys = np.interp(xs, mrec, mpre)
    return float(sum((ys[i] + ys[i + 1]) * (xs[i + 1] - xs[i]) / 2 for i in range(100)))


def smooth(y, f=0.05):
    nf = round(len(y) * f * 2) // 2 + 1
    h = nf // 2
    yp = np.concatenate((np.full(h, y[0]), y, np.full(h, y[-1])))
    return np.array([yp[i:i + nf].mean() for i in range(len(y))])


def ap_per_class(tp, conf, pred_cls, target_cls, eps=1e-16):
    """-> dict(classes, ap (nc,10), p, r, f1 at the max-mean-F1 confidence)."""
    order = np.argsort(-conf)
    tp, conf, pred_cls = tp[order], conf[order], pred_cls[order]
    classes = sorted(set(target_cls.tolist()))
    x = np.linspace(0, 1, 1000)
    ap = np.zeros((len(classes), tp.shape[1]))
    pc, rc = np.zeros((len(classes), 1000)), np.zeros((len(classes), 1000))
    for ci, c in enumerate(classes):
        sel = pred_cls == c
        n_l, n_p = int((target_cls == c).sum()), int(sel.sum())
        if n_p == 0 or n_l == 0:
            continue
        t = tp[sel].astype(np.float64)
        tpc, fpc = np.cumsum(t, 0), np.cumsum(1 - t, 0)
        recall, precision = tpc / (n_l + eps), tpc / (tpc + fpc)
        rc[ci] = np.interp(-x, -conf[sel], recall[:, 0], left=0)
        pc[ci] = np.interp(-x, -conf[sel], precision[:, 0], left=1)
        for j in range(tp.shape[1]):
            ap[ci, j] = compute_ap(recall[:, j], precision[:, j])
    f1c = 2 * pc * rc / (pc + rc + eps)
    i = int(smooth(f1c.mean(0), 0.1).argmax())
    return dict(classes=np.array(classes, int), ap=ap, p=pc[:, i], r=rc[:, i], f1=f1c[:, i])


def summary(res):
    """(mp, mr, map50, map, fitness) — Metric.mean_results / fitness (metrics.py:832-851)."""
    mp, mr = res["p"].mean(), res["r"].mean()
    m50, m = res["ap"][:, 0].mean(), res["ap"].mean()
    return mp, mr, m50, m, 0.1 * m50 + 0.9 * m
