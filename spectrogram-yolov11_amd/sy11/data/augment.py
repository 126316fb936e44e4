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
from .recipe import LazyImage, is_lazy, letterbox_image


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

    @staticmethod
    def _border(total, centred):
        """Border pixels before / after the image along one axis.  Centred: half each, rounded apart by -/+ 0.1 exactly as
        augment.py:1576-1577 does (an odd total puts its extra pixel after the image); else everything after it."""
        if not centred:
            return 0, int(round(total + 0.1))
        half = total / 2
        return int(round(half - 0.1)), int(round(half + 0.1))

    def geometry(self, shape, new_shape=None):
        """augment.py:1551-1580 — (new_unpad (w, h), ratio (w, h), top, bottom, left, right) for a (h, w) image."""
        target = self.new_shape if new_shape is None else new_shape
        th, tw = (target, target) if isinstance(target, int) else (target[0], target[1])
        h, w = shape[0], shape[1]
        gain = min(th / h, tw / w)
        if not self.scaleup:
            gain = min(gain, 1.0)
        if self.scaleFill and not self.auto:                  # stretch to the target: no border, per-axis ratios
            inner, ratio, pad_w, pad_h = (tw, th), (tw / w, th / h), 0.0, 0.0
        else:
            inner, ratio = (int(round(w * gain)), int(round(h * gain))), (gain, gain)
            pad_w, pad_h = tw - inner[0], th - inner[1]
            if self.auto:                                     # minimum rectangle: only up to the next stride multiple
                pad_w, pad_h = pad_w % self.stride, pad_h % self.stride
        top, bottom = self._border(pad_h, self.center)
        left, right = self._border(pad_w, self.center)
        return inner, ratio, top, bottom, left, right

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
        elif is_lazy(src):                                          # loader worker process: record the step, no pixels here
            out = letterbox_image(src, (H, W), (new_unpad[1], new_unpad[0]), top, left, 114)
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
        img, cls, inst = labels.pop("img"), labels.pop("cls"), labels.pop("instances")
        h, w = img.shape[:2]
        inst.convert_bbox(format=self.bbox_format)
        inst.denormalize(w, h)
        n = len(inst)
        boxes = inst.bboxes if n else np.zeros((0, 4), np.float32)
        if n and self.normalize:               # the same float32 divisions as the reference's tensor `/=`, done before wrapping
            boxes[:, 0::2] /= np.float32(w)
            boxes[:, 1::2] /= np.float32(h)
        out = {"img": self._format_img(img), "cls": torch.from_numpy(cls) if n else torch.zeros(0),
               "bboxes": torch.from_numpy(boxes) if n else torch.zeros((0, 4))}
        if self.batch_idx:
            out["batch_idx"] = torch.zeros(n)          # collate_fn adds the image's index in its batch
        labels.update(out)
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
        if is_lazy(img):
            t = img
        elif was_numpy:
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

    # -- loader worker processes (data/recipe.py): tiles may be LazyImage nodes — shapes without pixels
    def has_lazy(self):
        return any(is_lazy(t[0]) for t in self.tiles)

    def frozen(self):
        """A detached copy of the recipe (what a "render" LazyImage node carries)."""
        c = DeviceImage(list(self.tiles), self.canvas_hw, self.fill)
        c.minv, c.out_hw, c.lut, c.flip_ud, c.flip_lr = self.minv, self.out_hw, self.lut, self.flip_ud, self.flip_lr
        c.was_numpy, c.final_reverse_c = self.was_numpy, self.final_reverse_c
        return c

    def resolved(self, materialize):
        """The same recipe with every LazyImage tile turned into a device tensor (training process)."""
        if not self.has_lazy():
            return self
        c = self.frozen()
        c.tiles = [(materialize(t[0]), *t[1:]) for t in self.tiles]
        return c

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
        if self.has_lazy():
            if dst is None and not chw and not reverse_c and dtype == torch.uint8:
                return LazyImage((H, W), ("render", self.frozen()))          # a worker asked for intermediate pixels: defer them too
            raise RuntimeError("DeviceImage.render: the recipe still holds LazyImage tiles — resolve() it in the training process first")
        if dst is None:
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
# The host side of the training pipeline is a RECIPE GENERATOR: each transform draws its random numbers (in the reference's order,
# from the reference's two generators — `random` and `np.random` — so that a seeded run consumes the same streams), turns them into
# geometry (tile rectangles, one 3x3 matrix, three 256-entry tables, two flip bits) recorded on the DeviceImage, and moves the few
# boxes of the sample through that geometry in float32.  No pixel is touched here; sy11_image_mosaic_warp renders the recipe.
def mosaic_quadrant(i, xc, yc, w, h, side):
    """Tile ``i`` (bit 0: right column, bit 1: bottom row) of a 2x2 mosaic on a ``side`` x ``side`` canvas whose four tiles meet at
    (xc, yc): the tile's bottom-right / bottom-left / ... corner is pinned to the centre and whatever sticks out of the canvas is
    cropped.  -> ((x1, y1, x2, y2) on the canvas, (padw, padh) = canvas position of the tile's own origin)."""
    right, bottom = i & 1, i >> 1
    x1, x2 = (xc, min(xc + w, side)) if right else (max(xc - w, 0), xc)
    y1, y2 = (yc, min(yc + h, side)) if bottom else (max(yc - h, 0), yc)
    crop_x = 0 if right else w - (x2 - x1)          # first source column / row that lands on the canvas
    crop_y = 0 if bottom else h - (y2 - y1)
    return (x1, y1, x2, y2), (x1 - crop_x, y1 - crop_y)


class Mosaic:
    """augment.py:495-870 (BaseMixTransform.__call__ :352-407 + Mosaic), the 2x2 grid: same draws in the same order (apply coin,
    three buffer picks, the centre), same tile rectangles and label shifts.  The 2s x 2s canvas is never allocated — the result
    is a four-tile DeviceImage."""

    def __init__(self, dataset, imgsz=640, p=1.0, n=4):
        if not 0 <= p <= 1.0:
            raise AssertionError(f"The probability should be in range [0, 1], but got {p}.")
        if n != 4:
            raise NotImplementedError("sy11 Mosaic implements the 2x2 grid (n=4) the v8 pipeline uses")
        self.dataset, self.imgsz, self.p, self.n = dataset, imgsz, p, n
        self.pre_transform = None
        self.border = (-imgsz // 2, -imgsz // 2)

    def get_indexes(self, buffer=True):
        """The three partner samples: from the dataset's recent-sample buffer (default) or from the whole dataset."""
        k = self.n - 1
        if buffer:
            return random.choices(list(self.dataset.buffer), k=k)
        return [random.randint(0, len(self.dataset) - 1) for _ in range(k)]

    def __call__(self, labels):
        if random.uniform(0, 1) > self.p:                          # draw 1: apply at all?
            return labels
        partners = [self.dataset.get_image_and_label(i) for i in self.get_indexes()]        # draws 2-4
        if self.pre_transform is not None:
            partners = [self.pre_transform(d) for d in partners]
        return self._compose([labels] + partners)

    def _mosaic4(self, labels):
        """Reference entry point (labels carrying "mix_labels"); the work is _compose's."""
        return self._compose([labels] + list(labels.pop("mix_labels")))

    def _compose(self, samples):
        side = 2 * self.imgsz
        yc, xc = (int(random.uniform(-b, side + b)) for b in self.border)      # draws 5, 6: centre row, then centre column
        tiles, placed, from_numpy = [], [], False
        for i, sample in enumerate(samples):
            image = DeviceImage.wrap(sample["img"])
            from_numpy |= image.was_numpy
            h, w = sample.pop("resized_shape")
            rect, (padw, padh) = mosaic_quadrant(i, xc, yc, w, h, side)
            if rect[2] > rect[0] and rect[3] > rect[1]:             # a tile pushed entirely off the canvas contributes no pixels
                tiles.append((image.render(chw=False), *rect, padw, padh))
            placed.append(self._update_labels(sample, padw, padh, image.shape[:2]))
        canvas = DeviceImage(tiles, (side, side))
        canvas.was_numpy = from_numpy
        out = self._cat_labels(placed)
        out["img"] = canvas.unwrap()
        return out

    @staticmethod
    def _update_labels(labels, padw, padh, hw=None):
        """A tile's boxes in canvas pixels: corner format, de-normalised by the tile's size, shifted by its canvas position."""
        th, tw = hw if hw is not None else labels["img"].shape[:2]
        inst = labels["instances"]
        inst.convert_bbox(format="xyxy")
        inst.denormalize(tw, th)
        inst.add_padding(padw, padh)
        return labels

    def _cat_labels(self, mosaic_labels):
        """Boxes of all tiles together, clipped to the canvas; boxes clipped to nothing are dropped with their classes."""
        if not mosaic_labels:
            return {}
        side = 2 * self.imgsz
        first = mosaic_labels[0]
        inst = Instances.concatenate([d["instances"] for d in mosaic_labels], axis=0)
        inst.clip(side, side)
        alive = inst.remove_zero_area_boxes()
        return {"im_file": first.get("im_file"), "ori_shape": first.get("ori_shape"), "resized_shape": (side, side),
                "cls": np.concatenate([d["cls"] for d in mosaic_labels], 0)[alive], "instances": inst, "mosaic_border": self.border}


def _mat3(**entries):
    """float32 identity with the given entries, keyed 'rc' (row, column), e.g. _mat3(**{'02': tx})."""
    m = np.eye(3, dtype=np.float32)
    for key, v in entries.items():
        m[int(key[0]), int(key[1])] = v
    return m


def draw_affine(in_hw, out_wh, degrees, translate, scale, shear, perspective=0.0):
    """The eight draws of RandomPerspective.affine_transform in the reference's order (two perspective terms — drawn even though
    they are zero here —, angle, zoom, two shears, two translations) -> (M = T @ S @ R @ P @ C as float32 3x3, zoom).
    C centres the input, R rotates / zooms about the origin, S shears, T moves the origin to a jittered centre of the output."""
    h, w = in_hw
    C = _mat3(**{"02": -w / 2, "12": -h / 2})
    P = _mat3(**{"20": random.uniform(-perspective, perspective), "21": random.uniform(-perspective, perspective)})
    angle = random.uniform(-degrees, degrees)
    zoom = random.uniform(1 - scale, 1 + scale)
    R = _mat3()
    R[:2] = get_rotation_matrix_2d(center=(0, 0), angle=angle, scale=zoom)
    S = _mat3(**{"01": math.tan(random.uniform(-shear, shear) * math.pi / 180), "10": math.tan(random.uniform(-shear, shear) * math.pi / 180)})
    T = _mat3(**{"02": random.uniform(0.5 - translate, 0.5 + translate) * out_wh[0],
                 "12": random.uniform(0.5 - translate, 0.5 + translate) * out_wh[1]})
    M = T
    for nxt in (S, R, P, C):                                       # left to right, each product rounded to float32
        M = M @ nxt
    return M, zoom


def warp_boxes(bboxes, M):
    """xyxy boxes through the affine map ``M``: the axis-aligned hull of the four mapped corners (float32 throughout)."""
    n = len(bboxes)
    if n == 0:
        return bboxes
    corners = np.ones((n, 4, 3), dtype=bboxes.dtype)               # (x1,y1) (x2,y2) (x1,y2) (x2,y1), homogeneous
    corners[:, 0, :2] = bboxes[:, [0, 1]]
    corners[:, 1, :2] = bboxes[:, [2, 3]]
    corners[:, 2, :2] = bboxes[:, [0, 3]]
    corners[:, 3, :2] = bboxes[:, [2, 1]]
    mapped = (corners.reshape(4 * n, 3) @ M.T)[:, :2].reshape(n, 4, 2)
    return np.concatenate((mapped.min(1), mapped.max(1)), 1, dtype=bboxes.dtype)


def surviving_boxes(before, after, wh_thr=2, ar_thr=100, area_thr=0.1, eps=1e-16):
    """Which warped boxes are still usable: more than ``wh_thr`` pixels wide and high, at least ``area_thr`` of the (zoomed)
    original area left after clipping, aspect ratio below ``ar_thr``.  ``before`` / ``after``: (n, 4) xyxy."""
    w0, h0 = before[:, 2] - before[:, 0], before[:, 3] - before[:, 1]
    w1, h1 = after[:, 2] - after[:, 0], after[:, 3] - after[:, 1]
    aspect = np.maximum(w1 / (h1 + eps), h1 / (w1 + eps))
    big_enough = (w1 > wh_thr) & (h1 > wh_thr)
    return big_enough & (w1 * h1 / (w0 * h0 + eps) > area_thr) & (aspect < ar_thr)


class RandomPerspective:
    """augment.py:873-1300 for the affine case the detection pipeline uses (perspective = 0).  ``__call__`` = draw_affine + one
    recorded warp on the DeviceImage (executed by the fused render) + warp_boxes + surviving_boxes."""

    def __init__(self, degrees=0.0, translate=0.1, scale=0.5, shear=0.0, perspective=0.0, border=(0, 0), pre_transform=None):
        if perspective:
            raise NotImplementedError("sy11 RandomPerspective implements the affine case (perspective=0.0, the default)")
        self.degrees, self.translate, self.scale, self.shear, self.perspective = degrees, translate, scale, shear, perspective
        self.border = border
        self.pre_transform = pre_transform

    # the reference's method names, for callers that use the pieces
    def affine_transform(self, img, border):
        M, zoom = draw_affine(img.shape[:2], self.size, self.degrees, self.translate, self.scale, self.shear, self.perspective)
        if any(border) or (M != np.eye(3)).any():                  # the identity on an un-bordered image is skipped, as there
            img = DeviceImage.wrap(img).warp(M[:2], dsize=self.size).unwrap()
        return img, M, zoom

    apply_bboxes = staticmethod(warp_boxes)

    @staticmethod
    def box_candidates(box1, box2, wh_thr=2, ar_thr=100, area_thr=0.1, eps=1e-16):
        """Reference layout: boxes as (4, n)."""
        return surviving_boxes(box1.T, box2.T, wh_thr, ar_thr, area_thr, eps)

    def __call__(self, labels):
        mosaic = "mosaic_border" in labels
        if self.pre_transform and not mosaic:                      # a single image: letterbox it to the network size first
            labels = self.pre_transform(labels)
        labels.pop("ratio_pad", None)
        img = labels["img"]
        h, w = img.shape[:2]
        border = labels.pop("mosaic_border", self.border)
        self.size = (w + 2 * border[1], h + 2 * border[0])         # output (width, height): a mosaic canvas shrinks back to imgsz
        src = labels.pop("instances")
        src.convert_bbox(format="xyxy")
        src.denormalize(w, h)
        img, M, zoom = self.affine_transform(img, border)
        moved = Instances(warp_boxes(src.bboxes, M), bbox_format="xyxy", normalized=False)
        moved.clip(*self.size)
        src.scale(scale_w=zoom, scale_h=zoom, bbox_only=True)      # the un-clipped size the box would have: reference for the area test
        keep = surviving_boxes(src.bboxes, moved.bboxes, area_thr=0.10)
        labels.update(instances=moved[keep], cls=labels["cls"][keep], img=img, resized_shape=img.shape[:2])
        return labels


def hsv_tables(gains):
    """Three 256-entry uint8 tables for (hue, saturation, value) gains: hue wraps at 180 (OpenCV's 8-bit hue range), the other two
    saturate at 255."""
    ramp = np.arange(0, 256, dtype=gains.dtype)
    return ((ramp * gains[0]) % 180).astype(np.uint8), np.clip(ramp * gains[1], 0, 255).astype(np.uint8), \
        np.clip(ramp * gains[2], 0, 255).astype(np.uint8)


class RandomHSV:
    """augment.py:1303-1390 — ONE np.random.uniform(-1, 1, 3) draw scaled by the three gain limits; the colour conversions happen
    inside the fused render, from the tables recorded here."""

    def __init__(self, hgain=0.5, sgain=0.5, vgain=0.5):
        self.hgain, self.sgain, self.vgain = hgain, sgain, vgain

    def __call__(self, labels):
        limits = [self.hgain, self.sgain, self.vgain]
        if any(limits):
            gains = np.random.uniform(-1, 1, 3) * limits + 1
            labels["img"] = DeviceImage.wrap(labels["img"]).hsv(hsv_tables(gains)).unwrap()
        return labels


class RandomFlip:
    """augment.py:1393-1474 — one `random.random()` draw per instance (only the configured direction draws), boxes mirrored in xywh."""

    def __init__(self, p=0.5, direction="horizontal", flip_idx=None):
        if direction not in ("horizontal", "vertical"):
            raise AssertionError(f"Support direction `horizontal` or `vertical`, got {direction}")
        if not 0 <= p <= 1.0:
            raise AssertionError(f"The probability should be in range [0, 1], but got {p}.")
        self.p, self.direction, self.flip_idx = p, direction, flip_idx

    def __call__(self, labels):
        inst = labels.pop("instances")
        inst.convert_bbox(format="xywh")
        if random.random() < self.p:
            img = DeviceImage.wrap(labels["img"])
            h, w = (1, 1) if inst.normalized else img.shape[:2]     # normalised boxes mirror about 1
            if self.direction == "vertical":
                inst.flipud(h)
                labels["img"] = img.flip(ud=True).unwrap()
            else:
                inst.fliplr(w)
                labels["img"] = img.flip(lr=True).unwrap()
        labels["instances"] = inst
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
    if getattr(hyp, "copy_paste", 0.0):
        raise NotImplementedError("copy_paste needs segment labels (out of scope for detection)")
    geometry = Compose([Mosaic(dataset, imgsz=imgsz, p=hyp.mosaic),
                        RandomPerspective(degrees=hyp.degrees, translate=hyp.translate, scale=hyp.scale, shear=hyp.shear,
                                          perspective=hyp.perspective,
                                          pre_transform=None if stretch else LetterBox(new_shape=(imgsz, imgsz)))])
    colour = RandomHSV(hgain=hyp.hsv_h, sgain=hyp.hsv_s, vgain=hyp.hsv_v)
    flips = [RandomFlip(direction=d, p=p) for d, p in (("vertical", hyp.flipud), ("horizontal", hyp.fliplr))]
    return Compose([geometry, _NoMix(getattr(hyp, "mixup", 0.0)), colour, *flips])
