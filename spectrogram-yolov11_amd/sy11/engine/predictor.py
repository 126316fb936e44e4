"""Prediction path with the reference's contract (ultralytics/engine/predictor.py:66-167,222-306 and
models/yolo/detect/predict.py:23-73; results containers engine/results.py:187,1015-1122), hot path only:
preprocess (a list of raw HWC BGR uint8 images, uint8/float image tensors, or raw IQ) -> inference (fused DetectionModel
on libsy11) -> postprocess (non_max_suppression on the HIP bit-matrix NMS, scale_boxes) -> Results(boxes=(n,6)
[x1,y1,x2,y2,conf,cls]).  pre_transform's LetterBox and preprocess's BGR->RGB / HWC->CHW / /255 (predictor.py:118-163)
are one HIP launch per image writing into the batch tensor; only file decoding stays outside."""
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
    def __init__(self, orig_img, path, names, boxes=None, orig_shape=None):
        self.orig_img = orig_img
        self.orig_shape = orig_shape or (tuple(orig_img.shape[-2:]) if orig_img is not None else None)
        self.boxes = Boxes(boxes, self.orig_shape) if boxes is not None else None
        self.names = names
        self.path = path

    def __len__(self):
        return len(self.boxes) if self.boxes is not None else 0


class DetectionPredictor:
    """conf / iou / max_det defaults follow cfg/default.yaml (0.25 / 0.7 / 300)."""

    def __init__(self, model, device="cuda", conf=0.25, iou=0.7, max_det=300, classes=None, agnostic_nms=False, half=False,
                 producer=None, imgsz=640, graphs=True):
        self.device = torch.device(device)
        self.model = model.to(self.device).eval()
        self.model.fuse()                                   # predictor.setup_model -> AutoBackend(fuse=True)
        self.model._sy11_dtype = torch.float16 if half else torch.float32
        self.args = dict(conf=conf, iou=iou, max_det=max_det, classes=classes, agnostic_nms=agnostic_nms)
        self.producer = producer
        if graphs:                                          # batches of one shape replay a captured forward graph after 2 eager calls
            from . import enable_graphs
            enable_graphs(self.model)
        self.imgsz = (imgsz, imgsz) if isinstance(imgsz, int) else tuple(imgsz)
        self._lock = threading.Lock()                       # predictor.py:115: one inference at a time per predictor

    def pre_transform(self, im, out_dtype=None):
        """predictor.py:147-163 + :127-135 — [(h, w, 3) BGR uint8] * B -> (B, 3, H, W) RGB float/half in [0, 1] on the
        device.  LetterBox(imgsz, auto = all shapes equal, stride = model stride), as the reference's `pt` path."""
        from ..data.augment import LetterBox
        shapes = {tuple(x.shape) for x in im}
        stride = int(max(self.model.stride)) if hasattr(self.model, "stride") else 32
        lb = LetterBox(self.imgsz, auto=len(shapes) == 1, stride=stride, device=self.device)
        sizes = []
        for x in im:
            new_unpad, _, top, bottom, left, right = lb.geometry(x.shape[:2])
            sizes.append((new_unpad[1] + top + bottom, new_unpad[0] + left + right))
        if len(set(sizes)) != 1:                               # np.stack of the reference fails the same way
            raise ValueError(f"all input arrays must have the same shape after LetterBox, got {sorted(set(sizes))}")
        H, W = sizes[0]
        dtype = out_dtype or torch.float32                       # the stem kernel takes f32 NCHW and rounds for itself
        batch = torch.empty((len(im), 3, H, W), dtype=dtype, device=self.device)
        for i, x in enumerate(im):
            lb.into(x, batch[i], reverse_c=True)
        return batch

    def preprocess(self, im):
        """A list of raw (h, w, 3) BGR uint8 images (numpy or device tensors), a (B,3,H,W) uint8/float tensor (already
        letterboxed, RGB), or complex IQ (B, L) -> float image in [0, 1]."""
        if isinstance(im, (list, tuple)):
            return self.pre_transform(im)
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
            hw = tuple(orig.shape[:2]) if orig.shape[-1] == 3 and orig.ndim == 3 else tuple(orig.shape[-2:])   # HWC raw | CHW tensor
            pred = pred.clone()
            pred[:, :4] = ops.scale_boxes(img.shape[2:], pred[:, :4], hw)
            out.append(Results(orig, paths[i] if paths else None, self.model.names, boxes=pred[:, :6], orig_shape=hw))
        return out

    def __call__(self, source, orig_imgs=None, paths=None):
        with self._lock:
            if isinstance(source, (list, tuple)) and orig_imgs is None:
                orig_imgs = source
            im = self.preprocess(source)
            preds = self.inference(im)
            return self.postprocess(preds, im, orig_imgs, paths)
