"""Prediction path with the reference's contract (ultralytics/engine/predictor.py:66-167,222-306 and
models/yolo/detect/predict.py:23-73; results containers engine/results.py:187,1015-1122), hot path only:
preprocess (uint8/float image tensors or raw IQ) -> inference (fused DetectionModel on libsy11) -> postprocess
(non_max_suppression on the HIP bit-matrix NMS, scale_boxes) -> Results(boxes=(n,6) [x1,y1,x2,y2,conf,cls]).
Image decoding / LetterBox resizing is cv2-bound CPU work in the reference and stays outside."""
from __future__ import annotations

import threading

import torch

from ..utils import ops


class Boxes:
    """(n, 6) detections in original-image pixels (results.py:1015): xyxy, conf, cls accessors."""

    def __init__(self, boxes: torch.Tensor, orig_shape):
        assert boxes.shape[-1] == 6
        self.data = boxes
        self.orig_shape = orig_shape

    @property
    def xyxy(self):
        return self.data[:, :4]

    @property
    def conf(self):
        return self.data[:, -2]

    @property
    def cls(self):
        return self.data[:, -1]

    @property
    def xywh(self):
        return ops.xyxy2xywh(self.xyxy)

    def __len__(self):
        return self.data.shape[0]


class Results:
    def __init__(self, orig_img, path, names, boxes=None):
        self.orig_img = orig_img
        self.orig_shape = tuple(orig_img.shape[-2:]) if orig_img is not None else None
        self.boxes = Boxes(boxes, self.orig_shape) if boxes is not None else None
        self.names = names
        self.path = path

    def __len__(self):
        return len(self.boxes) if self.boxes is not None else 0


class DetectionPredictor:
    """conf / iou / max_det defaults follow cfg/default.yaml (0.25 / 0.7 / 300)."""

    def __init__(self, model, device="cuda", conf=0.25, iou=0.7, max_det=300, classes=None, agnostic_nms=False, half=False,
                 producer=None):
        self.device = torch.device(device)
        self.model = model.to(self.device).eval()
        self.model.fuse()                                   # predictor.setup_model -> AutoBackend(fuse=True)
        self.model._sy11_dtype = torch.float16 if half else torch.float32
        self.args = dict(conf=conf, iou=iou, max_det=max_det, classes=classes, agnostic_nms=agnostic_nms)
        self.producer = producer
        self._lock = threading.Lock()                       # predictor.py:115: one inference at a time per predictor

    def preprocess(self, im):
        """(B,3,H,W) uint8/float tensor (already letterboxed, RGB) or complex IQ (B, L) -> float image in [0, 1]."""
        if torch.is_complex(im):
            if self.producer is None:
                raise ValueError("raw IQ input needs a SpectrogramProducer")
            return self.producer(im.to(self.device))
        im = im.to(self.device)
        return im.float() / 255 if im.dtype == torch.uint8 else im.float()

    @torch.no_grad()
    def inference(self, im):
        return self.model(im.contiguous())

    def postprocess(self, preds, img, orig_imgs=None, paths=None):
        a = self.args
        preds = ops.non_max_suppression(preds, a["conf"], a["iou"], classes=a["classes"], agnostic=a["agnostic_nms"],
                                        max_det=a["max_det"])
        out = []
        for i, pred in enumerate(preds):
            orig = orig_imgs[i] if orig_imgs is not None else img[i]
            pred = pred.clone()
            pred[:, :4] = ops.scale_boxes(img.shape[2:], pred[:, :4], orig.shape[-2:])
            out.append(Results(orig, paths[i] if paths else None, self.model.names, boxes=pred[:, :6]))
        return out

    def __call__(self, source, orig_imgs=None, paths=None):
        with self._lock:
            im = self.preprocess(source)
            preds = self.inference(im)
            return self.postprocess(preds, im, orig_imgs, paths)
