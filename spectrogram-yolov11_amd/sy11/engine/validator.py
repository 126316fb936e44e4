"""Validation pass with the reference's semantics (ultralytics/engine/validator.py:104-264 and
models/yolo/detect/val.py:17-232), hot path only: inference (DetectionModel on libsy11, eval mode) ->
non_max_suppression(multi_label=True) on the HIP NMS -> per-image TP matrix at iouv = linspace(0.5, 0.95, 10) from the
device IoU matrix (sy11_box_iou) and the reference's IoU-sorted one-to-one matching -> ap_per_class -> DetMetrics.
Dataset / dataloader construction, plots, JSON export stay outside (cv2 / file I/O in the reference)."""
from __future__ import annotations

import numpy as np
import torch

from ..utils import ops
from ..utils.metrics import DetMetrics, box_iou_device


class DetectionValidator:
    """Defaults follow engine/validator.py:101-102 and cfg/default.yaml: conf 0.001, iou 0.7, max_det 300."""

    def __init__(self, model=None, device="cuda", conf=0.001, iou=0.7, max_det=300, single_cls=False, agnostic_nms=False,
                 half=False, producer=None, names=None):
        self.device = torch.device(device)
        self.args = dict(conf=conf, iou=iou, max_det=max_det, single_cls=single_cls, agnostic_nms=agnostic_nms, half=half)
        self.producer = producer
        self.iouv = torch.linspace(0.5, 0.95, 10)              # val.py:41
        self.niou = self.iouv.numel()
        self.model = model
        self.names = names
        self.init_metrics(model) if model is not None else None

    # ---- val.py:69-87
    def init_metrics(self, model):
        self.names = self.names or getattr(model, "names", None) or {i: str(i) for i in range(getattr(model, "nc", 80))}
        self.nc = len(self.names)
        self.metrics = DetMetrics(names=self.names)
        self.seen = 0
        self.stats = dict(tp=[], conf=[], pred_cls=[], target_cls=[], target_img=[])

    # ---- val.py:52-67
    def preprocess(self, batch):
        img = batch["img"]
        if torch.is_complex(img):
            if self.producer is None:
                raise ValueError("raw IQ input needs a SpectrogramProducer")
            img = self.producer(img.to(self.device))
        else:
            img = img.to(self.device)
            if img.dtype == torch.uint8:                    # val.py:54-55: one fused pass instead of .float() then / 255
                from .. import ops as K
                img = K.image_u8_to_float(img.contiguous())
            else:
                img = img.float()
        batch["img"] = img
        for k in ("batch_idx", "cls", "bboxes"):
            batch[k] = batch[k].to(self.device)
        return batch

    # ---- val.py:93-106
    def postprocess(self, preds):
        return ops.non_max_suppression(preds, self.args["conf"], self.args["iou"], nc=self.nc, multi_label=True,
                                       agnostic=self.args["single_cls"] or self.args["agnostic_nms"], max_det=self.args["max_det"])

    # ---- val.py:108-128
    def _prepare_batch(self, si, batch):
        idx = batch["batch_idx"] == si
        cls = batch["cls"][idx].squeeze(-1)
        bbox = batch["bboxes"][idx]
        imgsz = tuple(batch["img"].shape[2:])
        ori_shape = batch["ori_shape"][si] if "ori_shape" in batch else imgsz
        ratio_pad = batch["ratio_pad"][si] if "ratio_pad" in batch else None
        if len(cls):
            bbox = ops.xywh2xyxy(bbox) * torch.tensor(imgsz, device=bbox.device)[[1, 0, 1, 0]]
            ops.scale_boxes(imgsz, bbox, ori_shape, ratio_pad=ratio_pad)          # labels to native image space
        return {"cls": cls, "bbox": bbox, "ori_shape": ori_shape, "imgsz": imgsz, "ratio_pad": ratio_pad}

    def _prepare_pred(self, pred, pbatch):
        predn = pred.clone()
        ops.scale_boxes(pbatch["imgsz"], predn[:, :4], pbatch["ori_shape"], ratio_pad=pbatch["ratio_pad"])
        return predn

    # ---- val.py:130-175
    def update_metrics(self, preds, batch):
        for si, pred in enumerate(preds):
            self.seen += 1
            npr = len(pred)
            dev = pred.device
            stat = dict(conf=torch.zeros(0, device=dev), pred_cls=torch.zeros(0, device=dev),
                        tp=torch.zeros(npr, self.niou, dtype=torch.bool, device=dev))
            pbatch = self._prepare_batch(si, batch)
            cls, bbox = pbatch.pop("cls"), pbatch.pop("bbox")
            nl = len(cls)
            stat["target_cls"] = cls
            stat["target_img"] = cls.unique()
            if npr == 0:
                if nl:
                    for k in self.stats:
                        self.stats[k].append(stat[k])
                continue
            if self.args["single_cls"]:
                pred[:, 5] = 0
            predn = self._prepare_pred(pred, pbatch)
            stat["conf"] = predn[:, 4]
            stat["pred_cls"] = predn[:, 5]
            if nl:
                stat["tp"] = self._process_batch(predn, bbox, cls)
            for k in self.stats:
                self.stats[k].append(stat[k])

    # ---- val.py:205-232
    def _process_batch(self, detections, gt_bboxes, gt_cls):
        iou = box_iou_device(gt_bboxes, detections[:, :4])
        return self.match_predictions(detections[:, 5], gt_cls, iou)

    # ---- engine/validator.py:224-264 (use_scipy=False branch)
    def match_predictions(self, pred_classes, true_classes, iou):
        """(N,10) bool: detection n is a true positive at threshold t.  Candidates (label, detection) with matching class
        and IoU >= t are ranked by IoU; each detection keeps its best label, then each label keeps ONE detection — the
        lowest-index one among those left (the order np.unique leaves), not necessarily its best."""
        correct = np.zeros((pred_classes.shape[0], self.niou), dtype=bool)
        same = true_classes[:, None] == pred_classes
        iou = (iou * same).cpu().numpy()
        for i, thr in enumerate(self.iouv.tolist()):
            lab, det = np.nonzero(iou >= thr)
            if lab.size == 0:
                continue
            m = np.stack((lab, det), 1)
            if m.shape[0] > 1:
                m = m[iou[m[:, 0], m[:, 1]].argsort()[::-1]]
                m = m[np.unique(m[:, 1], return_index=True)[1]]
                m = m[np.unique(m[:, 0], return_index=True)[1]]
            correct[m[:, 1].astype(int), i] = True
        return torch.tensor(correct, dtype=torch.bool, device=pred_classes.device)

    # ---- val.py:182-190
    def get_stats(self):
        stats = {k: torch.cat(v, 0).cpu().numpy() for k, v in self.stats.items()}
        self.nt_per_class = np.bincount(stats["target_cls"].astype(int), minlength=self.nc)
        self.nt_per_image = np.bincount(stats["target_img"].astype(int), minlength=self.nc)
        stats.pop("target_img", None)
        if len(stats) and stats["tp"].any():
            self.metrics.process(**stats)
        return self.metrics.results_dict

    # ---- engine/validator.py:104-222, inference loop only
    @torch.no_grad()
    def __call__(self, model=None, batches=()):
        model = model or self.model
        was_training = model.training
        model = model.to(self.device).eval()
        model._sy11_dtype = torch.float16 if self.args["half"] else torch.float32
        if "_sy11_graph_cfg" not in model.__dict__:          # repeated batch shapes replay a captured forward graph
            from . import enable_graphs
            enable_graphs(model)
        self.init_metrics(model)
        for batch in batches:
            batch = self.preprocess(dict(batch))
            preds = model(batch["img"])
            preds = self.postprocess(preds[0] if isinstance(preds, (tuple, list)) else preds)
            self.update_metrics(preds, batch)
        stats = self.get_stats()
        model.train(was_training)
        return stats
