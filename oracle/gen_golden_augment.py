"""ORACLE — TEST INFRASTRUCTURE ONLY.  Golden vectors for the training augmentation chain.

Runs the REFERENCE's own v8_transforms pipeline (ultralytics/data/augment.py: Mosaic -> RandomPerspective -> MixUp(p=0)
-> Albumentations(absent) -> RandomHSV -> RandomFlip x2) + Format on a small in-memory dataset with seeded `random` /
`np.random`, and stores the inputs and the final image / boxes / classes of every sample.  The reference's control flow,
RNG draw order, matrices and label arithmetic are therefore recorded as they are; its cv2 pixel calls (warpAffine,
cvtColor, LUT, resize, copyMakeBorder, getRotationMatrix2D) are bound to oracle.image_ref's restatements because
opencv-python is not installed (pixel parity against a real cv2: unpinned, see oracle/image_ref.py).

Run:  python -m oracle.gen_golden_augment   ->  tests/golden/augment.npz
"""
from __future__ import annotations

import random
import sys

import numpy as np

from oracle import image_ref
from oracle.gen_golden import OUT, ROOT, import_reference

IMGSZ = 64
SHAPES = [(64, 48), (40, 64), (64, 64), (52, 64), (64, 33), (64, 64)]            # long side == imgsz (after load_image)
BASE = dict(hsv_h=0.015, hsv_s=0.7, hsv_v=0.4, degrees=0.0, translate=0.1, scale=0.5, shear=0.0, perspective=0.0,
            flipud=0.0, fliplr=0.5, bgr=0.0, mosaic=1.0, mixup=0.0, copy_paste=0.0, copy_paste_mode="flip", mask_ratio=4,
            overlap_mask=True)
CONFIGS = {
    "default": {},
    "rich": dict(degrees=10.0, shear=2.0, translate=0.2, flipud=0.5),
    "nomosaic": dict(mosaic=0.0, degrees=5.0),
    "halfmosaic": dict(mosaic=0.5, flipud=0.3),
}
N_SAMPLES = 6


def make_dataset_arrays():
    g = np.random.default_rng(99)
    imgs, boxes, clss = [], [], []
    for h, w in SHAPES:
        # smooth-ish content so that interpolation is exercised on non-trivial gradients as well as noise
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack(((xx * 3 + yy) % 256, (yy * 5) % 256, (xx * yy) % 256), -1)
        img = ((base + g.integers(0, 64, (h, w, 3))) % 256).astype(np.uint8)
        n = int(g.integers(1, 4))
        cxy = g.uniform(0.25, 0.75, (n, 2))
        wh = g.uniform(0.1, 0.4, (n, 2))
        imgs.append(img)
        boxes.append(np.concatenate((cxy, wh), 1).astype(np.float32))
        clss.append(g.integers(0, 3, (n, 1)).astype(np.float32))
    return imgs, boxes, clss


def bind_cv2():
    import cv2                                                   # the stub module of import_reference()
    cv2.INTER_LINEAR, cv2.BORDER_CONSTANT, cv2.COLOR_BGR2HSV, cv2.COLOR_HSV2BGR = 1, 0, 40, 54
    cv2.resize = lambda img, dsize, interpolation=None: image_ref.cv2_resize_linear_u8(img, dsize)
    cv2.copyMakeBorder = lambda img, t, b, l, r, kind, value=(114, 114, 114): image_ref.cv2_copy_make_border(img, t, b, l, r, value[0])
    cv2.getRotationMatrix2D = lambda angle, center, scale: image_ref.cv2_get_rotation_matrix_2d(center, angle, scale)
    cv2.warpAffine = lambda img, M, dsize, borderValue=(0, 0, 0): image_ref.cv2_warp_affine_u8(img, M, dsize, borderValue[0])
    cv2.split = lambda m: tuple(m[..., k] for k in range(m.shape[-1]))
    cv2.merge = lambda chans: np.stack(chans, -1)
    cv2.LUT = lambda src, lut: lut[src]

    def cvt(src, code, dst=None):
        out = image_ref.cv2_bgr2hsv_u8(src) if code == cv2.COLOR_BGR2HSV else image_ref.cv2_hsv2bgr_u8(src)
        if dst is not None:
            dst[...] = out
            return dst
        return out
    cv2.cvtColor = cvt


def main():
    import_reference()
    sys.path.insert(0, str(ROOT))
    bind_cv2()
    from ultralytics.data.augment import Format, v8_transforms
    from ultralytics.utils import IterableSimpleNamespace
    from ultralytics.utils.instance import Instances

    imgs, boxes, clss = make_dataset_arrays()

    class FakeDataset:
        """The four things the transforms touch: buffer, len, get_image_and_label, data / use_keypoints."""
        data, use_keypoints = {}, False

        def __init__(self):
            self.buffer = list(range(len(imgs)))

        def __len__(self):
            return len(imgs)

        def get_image_and_label(self, i):
            h, w = imgs[i].shape[:2]
            return {"im_file": f"im{i}", "ori_shape": (h, w), "resized_shape": (h, w), "img": imgs[i].copy(), "cls": clss[i].copy(),
                    "ratio_pad": (1.0, 1.0), "instances": Instances(boxes[i].copy(), np.zeros((0, 1000, 2), np.float32), None, "xywh", True)}

    store = {"n_images": np.asarray(len(imgs)), "configs": np.array(list(CONFIGS))}
    for i, (im, b, c) in enumerate(zip(imgs, boxes, clss)):
        store[f"in.{i}.img"], store[f"in.{i}.boxes"], store[f"in.{i}.cls"] = im, b, c
    for name, over in CONFIGS.items():
        hyp = IterableSimpleNamespace(**{**BASE, **over})
        ds = FakeDataset()
        tf = v8_transforms(ds, IMGSZ, hyp)
        tf.append(Format(bbox_format="xywh", normalize=True, batch_idx=True, bgr=hyp.bgr))
        random.seed(1234)
        np.random.seed(1234)
        for k in range(N_SAMPLES):
            out = tf(ds.get_image_and_label(k % len(imgs)))
            store[f"{name}.{k}.img"] = out["img"].numpy()
            store[f"{name}.{k}.bboxes"] = out["bboxes"].numpy()
            store[f"{name}.{k}.cls"] = out["cls"].numpy()
        store[f"{name}.rng_after"] = np.asarray([random.random(), np.random.uniform()])     # both streams consumed identically
    np.savez_compressed(OUT / "augment.npz", **store)
    print("wrote", OUT / "augment.npz", sum(v.nbytes for v in store.values()), "bytes raw")


if __name__ == "__main__":
    main()
