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
    """augment.py:82-195 — transforms applied in order; indexable with an int or a list of ints like the reference's."""

    def __init__(self, transforms):
        self.transforms = list(transforms) if isinstance(transforms, (list, tuple)) else [transforms]

    def __call__(self, data):
        for t in self.transforms:
            data = t(data)
        return data

    def append(self, transform):
        self.transforms.append(transform)

    def insert(self, index, transform):
        self.transforms.insert(index, transform)

    @staticmethod
    def _indices(index):
        if isinstance(index, int):
            return [index]
        if isinstance(index, list) and all(isinstance(i, int) for i in index):
            return index
        raise TypeError(f"Compose indices must be an int or a list of ints, got {type(index).__name__}")

    def __getitem__(self, index):
        return Compose([self.transforms[i] for i in self._indices(index)])

    def __setitem__(self, index, value):
        idx = self._indices(index)
        vals = [value] if isinstance(index, int) else list(value)
        if len(idx) != len(vals):
            raise ValueError(f"{len(idx)} positions but {len(vals)} transforms")
        for i, v in zip(idx, vals):
            self.transforms[i] = v                     # IndexError for a position that does not exist

    def tolist(self):
        return self.transforms

    def __repr__(self):
        return f"{type(self).__name__}({', '.join(map(str, self.transforms))})"


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
        lazy = isinstance(img, DeviceImage)
        if lazy:                                                    # a recipe from the dataset: needs its pixels now
            src, was_numpy = img.render(chw=False), img.was_numpy
        else:
            src, was_numpy = _to_device_u8(img, self.device)
        shape = tuple(src.shape[:2])
        new_shape = labels.pop("rect_shape", self.new_shape)
        if isinstance(new_shape, int):
            new_shape = (new_shape, new_shape)
        new_shape = tuple(int(v) for v in new_shape)
        new_unpad, ratio, top, bottom, left, right = self.geometry(shape, new_shape)
        H, W = new_unpad[1] + top + bottom, new_unpad[0] + left + right
        if (H, W) == shape and shape[::-1] == new_unpad:
            out = src                                               # nothing to resize, nothing to pad
        else:
            out = torch.empty((H, W, 3), dtype=torch.uint8, device=src.device)
            K.image_letterbox(src, out, (new_unpad[1], new_unpad[0]), top, left, 114, reverse_c=False, chw=False)
        if lazy:
            out_img = DeviceImage.wrap(out)
            out_img.was_numpy = was_numpy
        else:
            out_img = out.cpu().numpy() if was_numpy else out
        if labels.get("ratio_pad"):
            labels["ratio_pad"] = (labels["ratio_pad"], (left, top))
        if len(labels):
            labels = self._update_labels(labels, ratio, left, top, shape)
            labels["img"] = out_img
            labels["resized_shape"] = new_shape
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
                 mask_ratio=4, mask_overlap=True, batch_idx=True, bgr=0.0, defer=False):
        if return_mask or return_keypoint or return_obb:
            raise NotImplementedError("sy11 Format handles detection labels only (masks/keypoints/obb are out of scope)")
        self.bbox_format = bbox_format
        self.normalize = normalize
        self.batch_idx = batch_idx
        self.bgr = bgr
        self.defer = defer                     # keep a DeviceImage unrendered: collate_fn renders it into its batch slot

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
        if nl:
            b = instances.bboxes
            if self.normalize:                 # the same float32 divisions as the reference's tensor `/=`, done before wrapping
                b[:, [0, 2]] /= np.float32(w)
                b[:, [1, 3]] /= np.float32(h)
            labels["bboxes"] = torch.from_numpy(b)
        else:
            labels["bboxes"] = torch.zeros((nl, 4))
        if self.batch_idx:
            labels["batch_idx"] = torch.zeros(nl)
        return labels

    def _format_img(self, img):
        """augment.py:2070-2107 — HWC -> CHW, channel order reversed unless the bgr coin says keep."""
        flip = random.uniform(0, 1) > self.bgr
        if isinstance(img, DeviceImage):
            if self.defer:
                img.final_reverse_c = flip
                return img
            out = img.render(chw=True, reverse_c=flip)
            return out.cpu() if img.was_numpy else out
        if isinstance(img, np.ndarray):
            img = img.transpose(2, 0, 1)
            return torch.from_numpy(np.ascontiguousarray(img[::-1] if flip else img))
        img = img.permute(2, 0, 1)
        return (img.flip(0) if flip else img).contiguous()


# ------------------------------------------------------------------------------------------------ deferred pixels
class DeviceImage:
    """A uint8 HWC image that exists as a recipe until somebody needs the pixels.

    The reference's training transforms each rewrite the whole image on the host (paste four tiles, warpAffine, two
    colour conversions, two flips, a transpose).  Here they only edit this recipe — tiles on a canvas, one inverted
    affine map, one set of HSV tables, two flip bits, the output layout — and ``render`` produces the final pixels with
    ONE kernel launch (sy11_image_mosaic_warp), directly into the batch slot if one is given.  Steps that cannot be
    folded into the single pass (a second warp, a warp after HSV / flips) render to a temporary first, so any order of
    transforms still gives the sequential result.
    """

    def __init__(self, tiles, canvas_hw, fill=114):
        self.tiles = tiles                    # [(uint8 (h, w, 3) tensor, x1, y1, x2, y2, padw, padh)]
        self.canvas_hw = (int(canvas_hw[0]), int(canvas_hw[1]))
        self.fill = fill
        self.minv = None                      # 6 floats: the inverted 2x3 map, as cv::warpAffine computes it
        self.out_hw = self.canvas_hw
        self.lut = None                       # (3, 256) uint8
        self.flip_ud = False
        self.flip_lr = False
        self.was_numpy = False
        self.final_reverse_c = False          # set by Format(defer=True): the channel flip of _format_img

    # -- construction
    @classmethod
    def wrap(cls, img, device="cuda"):
        if isinstance(img, DeviceImage):
            return img
        was_numpy = isinstance(img, np.ndarray)
        if was_numpy:
            if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 3:
                raise ValueError(f"expected an (h, w, 3) uint8 image, got {img.dtype} {img.shape}")
            t = torch.from_numpy(np.ascontiguousarray(img))
            t = t.to(device, non_blocking=True) if torch.device(device).type == "cuda" else t
        else:
            if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
                raise ValueError(f"expected an (h, w, 3) uint8 image, got {img.dtype} {tuple(img.shape)}")
            t = img.contiguous()
        h, w = t.shape[:2]
        out = cls([(t, 0, 0, w, h, 0, 0)], (h, w))
        out.was_numpy = was_numpy
        return out

    # -- what the transforms look at
    @property
    def shape(self):
        return (self.out_hw[0], self.out_hw[1], 3)

    @property
    def device(self):
        return self.tiles[0][0].device if self.tiles else torch.device("cuda")

    @property
    def pending(self):
        return self.minv is not None or self.lut is not None or self.flip_ud or self.flip_lr

    def plain_tensor(self):
        """The underlying tensor when the recipe is exactly one untouched image, else None."""
        if len(self.tiles) == 1 and not self.pending:
            t, x1, y1, x2, y2, pw, ph = self.tiles[0]
            if (x1, y1, pw, ph) == (0, 0, 0, 0) and (y2, x2) == tuple(t.shape[:2]) == self.canvas_hw:
                return t
        return None

    # -- recipe edits
    def _flatten(self):
        t = self.render(chw=False)
        h, w = t.shape[:2]
        self.tiles, self.canvas_hw, self.out_hw = [(t, 0, 0, w, h, 0, 0)], (h, w), (h, w)
        self.minv, self.lut, self.flip_ud, self.flip_lr = None, None, False, False

    def warp(self, M23, dsize):
        """cv2.warpAffine(img, M23, dsize=(w, h), borderValue=fill) — recorded, not executed."""
        if self.pending:
            self._flatten()
        self.minv = invert_affine(M23)
        self.out_hw = (int(dsize[1]), int(dsize[0]))
        return self

    def hsv(self, luts):
        if self.lut is not None:
            self._flatten()
        self.lut = np.stack([np.asarray(t, np.uint8) for t in luts])
        return self

    def flip(self, ud=False, lr=False):
        self.flip_ud ^= bool(ud)
        self.flip_lr ^= bool(lr)
        return self

    # -- pixels
    def render(self, dst=None, chw=False, reverse_c=False, dtype=torch.uint8):
        H, W = self.out_hw
        if dst is None:
            plain = self.plain_tensor()
            if plain is not None and not chw and not reverse_c and dtype == torch.uint8:
                return plain
            dst = torch.empty((3, H, W) if chw else (H, W, 3), dtype=dtype, device=self.device)
        K.image_mosaic_warp(self.tiles, self.canvas_hw, dst, minv=self.minv, hsv_lut=self.lut, flip_ud=self.flip_ud,
                            flip_lr=self.flip_lr, fill=self.fill, reverse_c=reverse_c, chw=chw)
        return dst

    def unwrap(self):
        """What a transform hands back: numpy in -> numpy out (rendered now), device in -> the recipe itself."""
        return self.render(chw=False).cpu().numpy() if self.was_numpy else self


def invert_affine(M):
    """The inversion at the top of cv::warpAffine (imgwarp.cpp), float64 and in its operation order, so the fixed-point
    coordinate tables built from it are the ones cv2 would build."""
    m = [float(v) for v in np.asarray(M, np.float64).reshape(6)]
    D = m[0] * m[4] - m[1] * m[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[4] * D, m[0] * D
    m[0] = A11
    m[1] *= -D
    m[3] *= -D
    m[4] = A22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    return m


def get_rotation_matrix_2d(center, angle, scale):
    """cv2.getRotationMatrix2D (imgwarp.cpp): degrees, counter-clockwise positive, 2x3 float64."""
    a = angle * math.pi / 180
    alpha, beta = math.cos(a) * scale, math.sin(a) * scale
    cx, cy = center
    return np.array([[alpha, beta, (1 - alpha) * cx - beta * cy], [-beta, alpha, beta * cx + (1 - alpha) * cy]], np.float64)


# ------------------------------------------------------------------------------------------------ training transforms
class Mosaic:
    """augment.py:495-870 (BaseMixTransform.__call__ :352-407 + Mosaic), the 2x2 grid: same draws in the same order
    (apply coin, three buffer picks, the centre), same tile rectangles and label shifts.  The 2s x 2s canvas is never
    allocated — the result is a four-tile DeviceImage."""

    def __init__(self, dataset, imgsz=640, p=1.0, n=4):
        assert 0 <= p <= 1.0, f"The probability should be in range [0, 1], but got {p}."
        if n != 4:
            raise NotImplementedError("sy11 Mosaic implements the 2x2 grid (n=4) the v8 pipeline uses")
        self.dataset = dataset
        self.pre_transform = None
        self.p = p
        self.imgsz = imgsz
        self.border = (-imgsz // 2, -imgsz // 2)
        self.n = n

    def get_indexes(self, buffer=True):
        if buffer:
            return random.choices(list(self.dataset.buffer), k=self.n - 1)
        return [random.randint(0, len(self.dataset) - 1) for _ in range(self.n - 1)]

    def __call__(self, labels):
        if random.uniform(0, 1) > self.p:
            return labels
        indexes = self.get_indexes()
        mix_labels = [self.dataset.get_image_and_label(i) for i in indexes]
        if self.pre_transform is not None:
            mix_labels = [self.pre_transform(d) for d in mix_labels]
        labels["mix_labels"] = mix_labels
        labels = self._mosaic4(labels)
        labels.pop("mix_labels", None)
        return labels

    def _mosaic4(self, labels):
        mosaic_labels = []
        s = self.imgsz
        yc, xc = (int(random.uniform(-x, 2 * s + x)) for x in self.border)
        tiles = []
        was_numpy = False
        for i in range(4):
            patch = labels if i == 0 else labels["mix_labels"][i - 1]
            di = DeviceImage.wrap(patch["img"])
            was_numpy |= di.was_numpy
            h, w = patch.pop("resized_shape")
            if i == 0:      # top left
                x1a, y1a, x2a, y2a = max(xc - w, 0), max(yc - h, 0), xc, yc
                x1b, y1b, x2b, y2b = w - (x2a - x1a), h - (y2a - y1a), w, h
            elif i == 1:    # top right
                x1a, y1a, x2a, y2a = xc, max(yc - h, 0), min(xc + w, s * 2), yc
                x1b, y1b, x2b, y2b = 0, h - (y2a - y1a), min(w, x2a - x1a), h
            elif i == 2:    # bottom left
                x1a, y1a, x2a, y2a = max(xc - w, 0), yc, xc, min(s * 2, yc + h)
                x1b, y1b, x2b, y2b = w - (x2a - x1a), 0, w, min(y2a - y1a, h)
            else:           # bottom right
                x1a, y1a, x2a, y2a = xc, yc, min(xc + w, s * 2), min(s * 2, yc + h)
                x1b, y1b, x2b, y2b = 0, 0, min(w, x2a - x1a), min(y2a - y1a, h)
            padw, padh = x1a - x1b, y1a - y1b
            if x2a > x1a and y2a > y1a:
                tiles.append((di.render(chw=False), x1a, y1a, x2a, y2a, padw, padh))
            patch = self._update_labels(patch, padw, padh, di.shape[:2])
            mosaic_labels.append(patch)
        final = self._cat_labels(mosaic_labels)
        canvas = DeviceImage(tiles, (2 * s, 2 * s))
        canvas.was_numpy = was_numpy
        final["img"] = canvas.unwrap()
        return final

    @staticmethod
    def _update_labels(labels, padw, padh, hw=None):
        nh, nw = hw if hw is not None else labels["img"].shape[:2]
        labels["instances"].convert_bbox(format="xyxy")
        labels["instances"].denormalize(nw, nh)
        labels["instances"].add_padding(padw, padh)
        return labels

    def _cat_labels(self, mosaic_labels):
        if len(mosaic_labels) == 0:
            return {}
        imgsz = self.imgsz * 2
        final = {
            "im_file": mosaic_labels[0].get("im_file"),
            "ori_shape": mosaic_labels[0].get("ori_shape"),
            "resized_shape": (imgsz, imgsz),
            "cls": np.concatenate([l["cls"] for l in mosaic_labels], 0),
            "instances": Instances.concatenate([l["instances"] for l in mosaic_labels], axis=0),
            "mosaic_border": self.border,
        }
        final["instances"].clip(imgsz, imgsz)
        good = final["instances"].remove_zero_area_boxes()
        final["cls"] = final["cls"][good]
        return final


class RandomPerspective:
    """augment.py:873-1300 for the affine case the detection pipeline uses (perspective = 0): the eight draws in the
    reference's order, M = T @ S @ R @ P @ C in float32, boxes through the four corners, the box_candidates filter.
    The warp itself is recorded on the DeviceImage (executed by the fused render)."""

    def __init__(self, degrees=0.0, translate=0.1, scale=0.5, shear=0.0, perspective=0.0, border=(0, 0), pre_transform=None):
        if perspective:
            raise NotImplementedError("sy11 RandomPerspective implements the affine case (perspective=0.0, the default)")
        self.degrees = degrees
        self.translate = translate
        self.scale = scale
        self.shear = shear
        self.perspective = perspective
        self.border = border
        self.pre_transform = pre_transform

    def affine_transform(self, img, border):
        C = np.eye(3, dtype=np.float32)
        C[0, 2] = -img.shape[1] / 2
        C[1, 2] = -img.shape[0] / 2
        P = np.eye(3, dtype=np.float32)
        P[2, 0] = random.uniform(-self.perspective, self.perspective)
        P[2, 1] = random.uniform(-self.perspective, self.perspective)
        R = np.eye(3, dtype=np.float32)
        a = random.uniform(-self.degrees, self.degrees)
        s = random.uniform(1 - self.scale, 1 + self.scale)
        R[:2] = get_rotation_matrix_2d(angle=a, center=(0, 0), scale=s)
        S = np.eye(3, dtype=np.float32)
        S[0, 1] = math.tan(random.uniform(-self.shear, self.shear) * math.pi / 180)
        S[1, 0] = math.tan(random.uniform(-self.shear, self.shear) * math.pi / 180)
        T = np.eye(3, dtype=np.float32)
        T[0, 2] = random.uniform(0.5 - self.translate, 0.5 + self.translate) * self.size[0]
        T[1, 2] = random.uniform(0.5 - self.translate, 0.5 + self.translate) * self.size[1]
        M = T @ S @ R @ P @ C
        if (border[0] != 0) or (border[1] != 0) or (M != np.eye(3)).any():
            img = DeviceImage.wrap(img).warp(M[:2], dsize=self.size).unwrap()
        return img, M, s

    def apply_bboxes(self, bboxes, M):
        n = len(bboxes)
        if n == 0:
            return bboxes
        xy = np.ones((n * 4, 3), dtype=bboxes.dtype)
        xy[:, :2] = bboxes[:, [0, 1, 2, 3, 0, 3, 2, 1]].reshape(n * 4, 2)
        xy = xy @ M.T
        xy = xy[:, :2].reshape(n, 8)
        x, y = xy[:, [0, 2, 4, 6]], xy[:, [1, 3, 5, 7]]
        return np.concatenate((x.min(1), y.min(1), x.max(1), y.max(1)), dtype=bboxes.dtype).reshape(4, n).T

    def __call__(self, labels):
        if self.pre_transform and "mosaic_border" not in labels:
            labels = self.pre_transform(labels)
        labels.pop("ratio_pad", None)
        img = labels["img"]
        cls = labels["cls"]
        instances = labels.pop("instances")
        instances.convert_bbox(format="xyxy")
        instances.denormalize(*img.shape[:2][::-1])
        border = labels.pop("mosaic_border", self.border)
        self.size = img.shape[1] + border[1] * 2, img.shape[0] + border[0] * 2
        img, M, scale = self.affine_transform(img, border)
        bboxes = self.apply_bboxes(instances.bboxes, M)
        new_instances = Instances(bboxes, bbox_format="xyxy", normalized=False)
        new_instances.clip(*self.size)
        instances.scale(scale_w=scale, scale_h=scale, bbox_only=True)
        i = self.box_candidates(box1=instances.bboxes.T, box2=new_instances.bboxes.T, area_thr=0.10)
        labels["instances"] = new_instances[i]
        labels["cls"] = cls[i]
        labels["img"] = img
        labels["resized_shape"] = img.shape[:2]
        return labels

    @staticmethod
    def box_candidates(box1, box2, wh_thr=2, ar_thr=100, area_thr=0.1, eps=1e-16):
        w1, h1 = box1[2] - box1[0], box1[3] - box1[1]
        w2, h2 = box2[2] - box2[0], box2[3] - box2[1]
        ar = np.maximum(w2 / (h2 + eps), h2 / (w2 + eps))
        return (w2 > wh_thr) & (h2 > wh_thr) & (w2 * h2 / (w1 * h1 + eps) > area_thr) & (ar < ar_thr)


class RandomHSV:
    """augment.py:1303-1390 — the three gains from np.random.uniform, the three 256-entry tables built exactly as the
    reference builds them; the colour conversions happen inside the fused render."""

    def __init__(self, hgain=0.5, sgain=0.5, vgain=0.5):
        self.hgain = hgain
        self.sgain = sgain
        self.vgain = vgain

    def __call__(self, labels):
        img = labels["img"]
        if self.hgain or self.sgain or self.vgain:
            r = np.random.uniform(-1, 1, 3) * [self.hgain, self.sgain, self.vgain] + 1
            x = np.arange(0, 256, dtype=r.dtype)
            lut_hue = ((x * r[0]) % 180).astype(np.uint8)
            lut_sat = np.clip(x * r[1], 0, 255).astype(np.uint8)
            lut_val = np.clip(x * r[2], 0, 255).astype(np.uint8)
            labels["img"] = DeviceImage.wrap(img).hsv((lut_hue, lut_sat, lut_val)).unwrap()
        return labels


class RandomFlip:
    """augment.py:1393-1474 — one draw per instance (only the matching direction draws), boxes mirrored in xywh."""

    def __init__(self, p=0.5, direction="horizontal", flip_idx=None):
        assert direction in {"horizontal", "vertical"}, f"Support direction `horizontal` or `vertical`, got {direction}"
        assert 0 <= p <= 1.0, f"The probability should be in range [0, 1], but got {p}."
        self.p = p
        self.direction = direction
        self.flip_idx = flip_idx

    def __call__(self, labels):
        img = labels["img"]
        instances = labels.pop("instances")
        instances.convert_bbox(format="xywh")
        h, w = img.shape[:2]
        h = 1 if instances.normalized else h
        w = 1 if instances.normalized else w
        if self.direction == "vertical" and random.random() < self.p:
            img = DeviceImage.wrap(img).flip(ud=True).unwrap()
            instances.flipud(h)
        if self.direction == "horizontal" and random.random() < self.p:
            img = DeviceImage.wrap(img).flip(lr=True).unwrap()
            instances.fliplr(w)
        labels["img"] = img
        labels["instances"] = instances
        return labels


class _NoMix:
    """MixUp(p=0) / the slots of the v8 pipeline this build does not carry: draws what the reference draws, changes nothing."""

    def __init__(self, p=0.0):
        if p:
            raise NotImplementedError("sy11 carries the default detection pipeline (mixup = copy_paste = 0.0)")

    def __call__(self, labels):
        random.uniform(0, 1)                   # BaseMixTransform.__call__ :386 — the coin is drawn even when p = 0
        return labels


def v8_transforms(dataset, imgsz, hyp, stretch=False):
    """augment.py:2270-2342 for detection defaults: Mosaic -> RandomPerspective (LetterBox pre-transform for the non-mosaic
    samples) -> [CopyPaste: no segments, no draw] -> MixUp(p=0: one draw) -> [Albumentations: not installed, no draw] ->
    RandomHSV -> RandomFlip(vertical) -> RandomFlip(horizontal)."""
    mosaic = Mosaic(dataset, imgsz=imgsz, p=hyp.mosaic)
    affine = RandomPerspective(degrees=hyp.degrees, translate=hyp.translate, scale=hyp.scale, shear=hyp.shear,
                               perspective=hyp.perspective,
                               pre_transform=None if stretch else LetterBox(new_shape=(imgsz, imgsz)))
    if getattr(hyp, "copy_paste", 0.0):
        raise NotImplementedError("copy_paste needs segment labels (out of scope for detection)")
    return Compose([Compose([mosaic, affine]), _NoMix(getattr(hyp, "mixup", 0.0)),
                    RandomHSV(hgain=hyp.hsv_h, sgain=hyp.hsv_s, vgain=hyp.hsv_v),
                    RandomFlip(direction="vertical", p=hyp.flipud), RandomFlip(direction="horizontal", p=hyp.fliplr)])
