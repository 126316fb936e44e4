"""Box container of the data path, with the reference's names (ultralytics/utils/instance.py: Bboxes :34-182,
Instances :185-430).  Detection only: boxes are an (n, 4) float32 numpy array in "xyxy" or "xywh"; segments / keypoints
of the reference's other tasks are out of scope and rejected loudly instead of being carried along.

Everything here is label geometry on a handful of boxes per image — host-side numpy by design; the pixels are what the
GPU moves (sy11.data.augment)."""
from __future__ import annotations

import numpy as np

_FORMATS = ("xyxy", "xywh")


def _xywh2xyxy(b):
    out = np.empty_like(b)
    half = b[..., 2:] / 2
    out[..., :2] = b[..., :2] - half
    out[..., 2:] = b[..., :2] + half
    return out


def _xyxy2xywh(b):
    out = np.empty_like(b)
    out[..., 0] = (b[..., 0] + b[..., 2]) / 2
    out[..., 1] = (b[..., 1] + b[..., 3]) / 2
    out[..., 2] = b[..., 2] - b[..., 0]
    out[..., 3] = b[..., 3] - b[..., 1]
    return out


class Instances:
    """instance.py:185-430 restricted to boxes.  `normalized` tracks whether coordinates are fractions of the image."""

    def __init__(self, bboxes, segments=None, keypoints=None, bbox_format="xywh", normalized=True):
        if keypoints is not None or (segments is not None and len(segments)):
            raise NotImplementedError("sy11 Instances carries detection boxes only (segments/keypoints are out of scope)")
        if bbox_format not in _FORMATS:
            raise ValueError(f"bbox_format must be one of {_FORMATS}, got {bbox_format!r}")
        b = np.asarray(bboxes, dtype=np.float32)
        self._bboxes = b.reshape(-1, 4) if b.ndim != 2 else b
        self.format = bbox_format
        self.normalized = normalized

    # -- representation
    @property
    def bboxes(self):
        return self._bboxes

    def __len__(self):
        return len(self._bboxes)

    def __getitem__(self, index):
        return Instances(self._bboxes[index], bbox_format=self.format, normalized=self.normalized)

    def convert_bbox(self, format):
        if format not in _FORMATS:
            raise ValueError(f"bbox format must be one of {_FORMATS}, got {format!r}")
        if format != self.format:
            self._bboxes = _xywh2xyxy(self._bboxes) if format == "xyxy" else _xyxy2xywh(self._bboxes)
            self.format = format

    @property
    def bbox_areas(self):
        b = self._bboxes
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) if self.format == "xyxy" else b[:, 3] * b[:, 2]

    # -- arithmetic (column by column, like Bboxes.mul / add, so float32 rounding matches)
    def _mul(self, s):
        for c in range(4):
            self._bboxes[:, c] *= s[c]

    def scale(self, scale_w, scale_h, bbox_only=False):
        self._mul((scale_w, scale_h, scale_w, scale_h))

    def denormalize(self, w, h):
        if self.normalized:
            self._mul((w, h, w, h))
            self.normalized = False

    def normalize(self, w, h):
        if not self.normalized:
            self._mul((1 / w, 1 / h, 1 / w, 1 / h))
            self.normalized = True

    def add_padding(self, padw, padh):
        assert not self.normalized, "you should add padding with absolute coordinates."
        for c, o in enumerate((padw, padh, padw, padh)):
            self._bboxes[:, c] += o

    def fliplr(self, w):
        if self.format == "xyxy":
            x1, x2 = self._bboxes[:, 0].copy(), self._bboxes[:, 2].copy()
            self._bboxes[:, 0] = w - x2
            self._bboxes[:, 2] = w - x1
        else:
            self._bboxes[:, 0] = w - self._bboxes[:, 0]

    def flipud(self, h):
        if self.format == "xyxy":
            y1, y2 = self._bboxes[:, 1].copy(), self._bboxes[:, 3].copy()
            self._bboxes[:, 1] = h - y2
            self._bboxes[:, 3] = h - y1
        else:
            self._bboxes[:, 1] = h - self._bboxes[:, 1]

    def clip(self, w, h):
        ori = self.format
        self.convert_bbox("xyxy")
        self._bboxes[:, [0, 2]] = self._bboxes[:, [0, 2]].clip(0, w)
        self._bboxes[:, [1, 3]] = self._bboxes[:, [1, 3]].clip(0, h)
        if ori != "xyxy":
            self.convert_bbox(ori)

    def remove_zero_area_boxes(self):
        good = self.bbox_areas > 0
        if not all(good):
            self._bboxes = self._bboxes[good]
        return good

    def update(self, bboxes, segments=None, keypoints=None):
        self._bboxes = np.asarray(bboxes, dtype=np.float32).reshape(-1, 4)

    @classmethod
    def concatenate(cls, instances_list, axis=0):
        assert isinstance(instances_list, (list, tuple))
        if not instances_list:
            return cls(np.empty((0, 4), np.float32))
        fmt, norm = instances_list[0].format, instances_list[0].normalized
        assert all(i.format == fmt for i in instances_list)
        return cls(np.concatenate([i.bboxes for i in instances_list], axis=axis), bbox_format=fmt, normalized=norm)
