"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product package).

A whole (short) training run and a validation pass on the CPU, assembled from the other restatements: forward + loss
(yolo11_ref / loss_ref, autograd for the backward), the reference trainer's update rule (trainer_ref: clip 10, SGD-nesterov with
the three parameter groups, EMA over every float state entry), and the validator's metric chain (yolo11_ref eval forward +
Detect decode -> nms_ref.non_max_suppression(multi_label=True, conf 0.001, iou 0.7, max_det 300: models/yolo/detect/val.py:93-106)
-> metrics_ref box_iou / match_predictions / ap_per_class -> mAP@0.5).  Used by the accuracy gate to train the same model on the
same data as the HIP trainer.  Parity status: every piece is PINNED on its own (see the modules named above)."""
from __future__ import annotations

import numpy as np
import torch

from . import loss_ref, metrics_ref, nms_ref, trainer_ref as T, yolo11_ref as R


def train(sd0, layers, nc, images, labels, batch, steps, lr=0.01, momentum=0.937, decay=5e-4, name="SGD", log=None):
    """-> (RefTrainerState, [loss per step]).  ``labels`` = (batch_idx, cls, bboxes) of the whole set; mini-batches are taken
    in order, wrapping around, exactly as the device-side loop of the gate does."""
    from .synth_iq import take
    trainable = [k for k, v in sd0.items() if v.dtype.is_floating_point and "running" not in k and ".dfl." not in k]
    norm = {k for k in trainable if k.endswith("bn.weight")}
    st = T.RefTrainerState(sd0, trainable, norm, name=name, lr=lr, momentum=momentum, decay=decay)
    n = images.shape[0]
    losses = []
    for it in range(steps):
        lo = (it * batch) % n
        if lo + batch > n:
            lo = 0
        live = {k: (v.detach().requires_grad_(True) if k in st.wd else v) for k, v in st.sd.items()}
        maps = R.forward(live, layers, images[lo:lo + batch], train=True)           # BN running statistics update in place (st.sd)
        loss, _ = loss_ref.detection_loss(maps, take(*labels, lo, lo + batch), nc=nc)
        loss.backward()
        losses.append(float(loss.detach()))
        if log is not None and (it % 20 == 0 or it == steps - 1):
            log(f"oracle step {it}: loss {losses[-1]:.3f}")
        with torch.no_grad():
            st.optimizer_step({k: live[k].grad for k in trainable if live[k].grad is not None})
    return st, losses


def validate(sd, layers, nc, images, labels, batch=16, conf=0.001, iou=0.7, max_det=300):
    """-> dict(map50, map, precision, recall) of the eval-mode model on (images, labels)."""
    from .synth_iq import take
    iouv = np.linspace(0.5, 0.95, 10)
    H, W = images.shape[2:]
    tps, confs, pcls, tcls = [], [], [], []
    with torch.no_grad():
        for lo in range(0, images.shape[0], batch):
            hi = min(lo + batch, images.shape[0])
            y, _ = R.forward({k: v.clone() for k, v in sd.items()}, layers, images[lo:hi], train=False)
            dets, _ = nms_ref.non_max_suppression(y, conf, iou, multi_label=True, max_det=max_det, nc=nc)
            lab = take(*labels, lo, hi)
            for i, d in enumerate(dets):
                sel = lab["batch_idx"] == i
                gcls = lab["cls"][sel].reshape(-1).numpy()
                gbox = nms_ref.xywh2xyxy(lab["bboxes"][sel]) * torch.tensor([W, H, W, H], dtype=torch.float32)
                tcls.append(gcls)
                if d.shape[0] == 0:
                    continue
                tp = np.zeros((d.shape[0], 10), bool)
                if len(gcls):
                    tp = metrics_ref.match_predictions(d[:, 5].numpy(), gcls, metrics_ref.box_iou(gbox.numpy(), d[:, :4].numpy()), iouv)
                tps.append(tp)
                confs.append(d[:, 4].numpy())
                pcls.append(d[:, 5].numpy())
    if not tps:
        return dict(map50=0.0, map=0.0, precision=0.0, recall=0.0)
    res = metrics_ref.ap_per_class(np.concatenate(tps), np.concatenate(confs), np.concatenate(pcls), np.concatenate(tcls))
    mp, mr, m50, m, _ = metrics_ref.summary(res)
    return dict(map50=float(m50), map=float(m), precision=float(mp), recall=float(mr))
