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

    # ---- val.py:108-128: one image's labels, mapped back to the image as it was before LetterBox
    def _image_geometry(self, si, batch):
        net_hw = tuple(batch["img"].shape[2:])
        native_hw = batch["ori_shape"][si] if "ori_shape" in batch else net_hw
        return net_hw, native_hw, (batch["ratio_pad"][si] if "ratio_pad" in batch else None)

    def _prepare_batch(self, si, batch):
        net_hw, native_hw, ratio_pad = self._image_geometry(si, batch)
        mine = batch["batch_idx"] == si
        cls, box = batch["cls"][mine].squeeze(-1), batch["bboxes"][mine]
        if cls.numel():
            h, w = net_hw
            box = ops.xywh2xyxy(box) * box.new_tensor([w, h, w, h])             # normalised -> network-input pixels
            ops.scale_boxes(net_hw, box, native_hw, ratio_pad=ratio_pad)        # -> native pixels (in place)
        return {"cls": cls, "bbox": box, "ori_shape": native_hw, "imgsz": net_hw, "ratio_pad": ratio_pad}

    def _prepare_pred(self, pred, pbatch):
        native = pred.clone()
        ops.scale_boxes(pbatch["imgsz"], native[:, :4], pbatch["ori_shape"], ratio_pad=pbatch["ratio_pad"])
        return native

    def _record(self, conf, pred_cls, tp, target_cls):
        """One image's contribution to the running statistics (the five lists get_stats concatenates)."""
        for key, val in (("conf", conf), ("pred_cls", pred_cls), ("tp", tp), ("target_cls", target_cls), ("target_img", target_cls.unique())):
            self.stats[key].append(val)

    # ---- val.py:130-175
    def update_metrics(self, preds, batch):
        for si, det in enumerate(preds):
            self.seen += 1
            labels = self._prepare_batch(si, batch)
            gt_cls, gt_box = labels["cls"], labels["bbox"]
            if det.shape[0] == 0:
                if gt_cls.numel():                                   # missed labels still count as targets; an empty image adds nothing
                    none = torch.zeros(0, device=det.device)
                    self._record(none, none, torch.zeros(0, self.niou, dtype=torch.bool, device=det.device), gt_cls)
                continue
            if self.args["single_cls"]:
                det[:, 5] = 0
            native = self._prepare_pred(det, labels)
            tp = (self._process_batch(native, gt_box, gt_cls) if gt_cls.numel()
                  else torch.zeros(det.shape[0], self.niou, dtype=torch.bool, device=det.device))
            self._record(native[:, 4], native[:, 5], tp, gt_cls)

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
