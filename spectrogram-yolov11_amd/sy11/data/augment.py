"""Image transforms of the data path with the reference's names and call signatures (ultralytics/data/augment.py:
Compose :82-195, LetterBox :1477-1633, Format :1926-2180), re-cut for the MI355X: the pixels never pass through cv2
on the host — an image is uploaded once as raw uint8 HWC and every resampling step is one HIP kernel launch
(sy11_image_letterbox) that can write straight into a slot of the batch tensor the model consumes; only the label
geometry (a few boxes per image) stays in numpy.

Images may be numpy arrays (uploaded, transformed, downloaded: drop-in behaviour for host-side callers and tests) or
uint8 CUDA tensors (stay on the device, nothing synchronises)."""
from __future__ import annotations

import math
import random

import numpy as np
import torch

from .. import ops as K
from ..utils.instance import Instances


def _to_device_u8(img, device):
    """numpy HWC uint8 / torch uint8 (cpu or cuda) -> (cuda tensor, was_numpy)."""
    if isinstance(img, np.ndarray):
        if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 3:
            raise ValueError(f"expected an (h, w, 3) uint8 image, got {img.dtype} {img.shape}")
        return torch.from_numpy(np.ascontiguousarray(img)).to(device, non_blocking=True), True
    if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
        raise ValueError(f"expected an (h, w, 3) uint8 image, got {img.dtype} {tuple(img.shape)}")
    return (img if img.is_cuda else img.to(device, non_blocking=True)).contiguous(), False


class Compose:
    """augment.py:82-195 — a list of transforms applied in order."""

    def __init__(self, transforms):
        self.transforms = transforms if isinstance(transforms, list) else [transforms]

    def __call__(self, data):
        for t in self.transforms:
            data = t(data)
        return data

    def append(self, transform):
        self.transforms.append(transform)

    def insert(self, index, transform):
        self.transforms.insert(index, transform)

    def __getitem__(self, index):
        assert isinstance(index, (int, list)), f"The indices should be either list or int type but got {type(index)}"
        index = [index] if isinstance(index, int) else index
        return Compose([self.transforms[i] for i in index])

    def __setitem__(self, index, value):
        assert isinstance(index, (int, list)), f"The indices should be either list or int type but got {type(index)}"
        if isinstance(index, list):
            assert isinstance(value, list), f"The indices should be the same type as values, but got {type(index)} and {type(value)}"
        if isinstance(index, int):
            index, value = [index], [value]
        for i, v in zip(index, value):
            assert i < len(self.transforms), f"list index {i} out of range {len(self.transforms)}."
            self.transforms[i] = v

    def tolist(self):
        return self.transforms

    def __repr__(self):
        return f"{self.__class__.__name__}({', '.join([f'{t}' for t in self.transforms])})"


class LetterBox:
    """augment.py:1477-1633.  Same constructor, same ``__call__(labels=None, image=None)`` contract (returns the image
    when called without labels, else the updated labels dict), same geometry; the resize + border run on the GPU."""

    def __init__(self, new_shape=(640, 640), auto=False, scaleFill=False, scaleup=True, center=True, stride=32, device="cuda"):
        self.new_shape = new_shape
        self.auto = auto
        self.scaleFill = scaleFill
        self.scaleup = scaleup
        self.stride = stride
        self.center = center
        self.device = device

    def geometry(self, shape, new_shape=None):
        """augment.py:1551-1580 — (new_unpad (w, h), ratio (w, h), top, bottom, left, right) for a (h, w) image."""
        new_shape = self.new_shape if new_shape is None else new_shape
        if isinstance(new_shape, int):
            new_shape = (new_shape, new_shape)
        r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
        if not self.scaleup:
            r = min(r, 1.0)
        ratio = r, r
        new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
        dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
        if self.auto:
            dw, dh = np.mod(dw, self.stride), np.mod(dh, self.stride)
        elif self.scaleFill:
            dw, dh = 0.0, 0.0
            new_unpad = (new_shape[1], new_shape[0])
            ratio = new_shape[1] / shape[1], new_shape[0] / shape[0]
        if self.center:
            dw /= 2
            dh /= 2
        top, bottom = int(round(dh - 0.1)) if self.center else 0, int(round(dh + 0.1))
        left, right = int(round(dw - 0.1)) if self.center else 0, int(round(dw + 0.1))
        return new_unpad, ratio, top, bottom, left, right

    def into(self, image, dst, new_shape=None, reverse_c=True):
        """Letterbox ``image`` straight into ``dst`` — one (3, H, W) slot of a batch tensor, uint8 or float (/255, with
        BGR->RGB when reverse_c): LetterBox + the transpose / flip / divide of predictor.preprocess in a single launch.
        ``dst``'s H x W must equal the letterboxed size.  Returns (ratio, (left, top))."""
        src, _ = _to_device_u8(image, dst.device)
        new_unpad, ratio, top, bottom, left, right = self.geometry(src.shape[:2], new_shape)
        H, W = new_unpad[1] + top + bottom, new_unpad[0] + left + right
        if tuple(dst.shape) != (3, H, W):
            raise ValueError(f"LetterBox.into: dst is {tuple(dst.shape)}, the letterboxed image is (3, {H}, {W})")
        K.image_letterbox(src, dst, (new_unpad[1], new_unpad[0]), top, left, 114, reverse_c=reverse_c, chw=True)
        return ratio, (left, top)

    def __call__(self, labels=None, image=None):
        if labels is None:
            labels = {}
        img = labels.get("img") if image is None else image
        src, was_numpy = _to_device_u8(img, self.device)
        shape = tuple(src.shape[:2])
        new_shape = labels.pop("rect_shape", self.new_shape)
        new_unpad, ratio, top, bottom, left, right = self.geometry(shape, new_shape)
        H, W = new_unpad[1] + top + bottom, new_unpad[0] + left + right
        if (H, W) == shape and shape[::-1] == new_unpad:
            out = src                                               # nothing to resize, nothing to pad
        else:
            out = torch.empty((H, W, 3), dtype=torch.uint8, device=src.device)
            K.image_letterbox(src, out, (new_unpad[1], new_unpad[0]), top, left, 114, reverse_c=False, chw=False)
        out_img = out.cpu().numpy() if was_numpy else out
        if labels.get("ratio_pad"):
            labels["ratio_pad"] = (labels["ratio_pad"], (left, top))
        if len(labels):
            labels = self._update_labels(labels, ratio, left, top, shape)
            labels["img"] = out_img
            labels["resized_shape"] = new_shape if not isinstance(new_shape, int) else (new_shape, new_shape)
            return labels
        return out_img

    @staticmethod
    def _update_labels(labels, ratio, padw, padh, shape=None):
        """augment.py:1600-1633 — boxes to xyxy pixels of the source image, scaled by ratio, shifted by the padding."""
        h, w = shape if shape is not None else labels["img"].shape[:2]
        labels["instances"].convert_bbox(format="xyxy")
        labels["instances"].denormalize(w, h)
        labels["instances"].scale(*ratio)
        labels["instances"].add_padding(padw, padh)
        return labels


class Format:
    """augment.py:1926-2180 for detection: boxes -> `bbox_format` (normalised), image HWC -> CHW with the BGR->RGB flip
    drawn as the reference draws it (``random.uniform(0, 1) > bgr``), plus the empty ``batch_idx`` collate_fn fills."""

    def __init__(self, bbox_format="xywh", normalize=True, return_mask=False, return_keypoint=False, return_obb=False,
                 mask_ratio=4, mask_overlap=True, batch_idx=True, bgr=0.0):
        if return_mask or return_keypoint or return_obb:
            raise NotImplementedError("sy11 Format handles detection labels only (masks/keypoints/obb are out of scope)")
        self.bbox_format = bbox_format
        self.normalize = normalize
        self.batch_idx = batch_idx
        self.bgr = bgr

    def __call__(self, labels):
        img = labels.pop("img")
        h, w = img.shape[:2]
        cls = labels.pop("cls")
        instances = labels.pop("instances")
        instances.convert_bbox(format=self.bbox_format)
        instances.denormalize(w, h)
        nl = len(instances)
        labels["img"] = self._format_img(img)
        labels["cls"] = torch.from_numpy(cls) if nl else torch.zeros(nl)
        labels["bboxes"] = torch.from_numpy(instances.bboxes) if nl else torch.zeros((nl, 4))
        if self.normalize:
            labels["bboxes"][:, [0, 2]] /= w
            labels["bboxes"][:, [1, 3]] /= h
        if self.batch_idx:
            labels["batch_idx"] = torch.zeros(nl)
        return labels

    def _format_img(self, img):
        """augment.py:2070-2107 — HWC -> CHW, channel order reversed unless the bgr coin says keep."""
        flip = random.uniform(0, 1) > self.bgr
        if isinstance(img, np.ndarray):
            img = img.transpose(2, 0, 1)
            return torch.from_numpy(np.ascontiguousarray(img[::-1] if flip else img))
        img = img.permute(2, 0, 1)
        return (img.flip(0) if flip else img).contiguous()
