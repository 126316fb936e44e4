"""ORACLE — TEST INFRASTRUCTURE ONLY.  Golden vectors for the image side of preprocess (SURVEY.md §8(a) row 14 and
§8(f) rank 4): LetterBox, Format, Instances and the predictor's preprocess.

Drives the REFERENCE's own LetterBox / Format / Instances classes (ultralytics/data/augment.py, utils/instance.py).  The
reference calls cv2.resize / cv2.copyMakeBorder for the pixels; opencv-python is absent from this image (and not
vendored), so those two calls are bound to oracle.image_ref's restatement of OpenCV's published 8-bit INTER_LINEAR
algorithm.  The geometry, label arithmetic and control flow recorded here are therefore the reference's; the resampled
pixel values are "restated cv2" (parity unpinned against a real cv2 — see oracle/image_ref.py).

Run:  python -m oracle.gen_golden_image   ->  tests/golden/image.npz
"""
from __future__ import annotations

import random
import sys

import numpy as np

from oracle import image_ref
from oracle.gen_golden import OUT, ROOT, import_reference

# (tag, (h, w), LetterBox kwargs, n boxes)
CASES = [
    ("bus", (108, 81), dict(new_shape=(64, 64), auto=True, stride=32), 3),          # the 1080x810 bus.jpg geometry / 10
    ("square_down", (90, 90), dict(new_shape=(64, 64)), 2),
    ("wide_pad", (50, 120), dict(new_shape=(64, 64)), 4),
    ("half", (128, 96), dict(new_shape=(64, 64)), 2),                                # exact 2x shrink -> box mean
    ("up", (20, 33), dict(new_shape=(64, 96)), 1),                                   # scaleup
    ("no_up", (20, 33), dict(new_shape=(64, 96), scaleup=False), 1),                 # val transform: pad only
    ("fill", (37, 53), dict(new_shape=(64, 64), scaleFill=True), 2),
    ("corner", (70, 41), dict(new_shape=(64, 64), center=False), 0),
    ("same", (64, 64), dict(new_shape=(64, 64)), 2),                                 # nothing to do
    ("rect", (75, 100), dict(new_shape=(96, 96)), 2),                                # labels carry rect_shape (48, 64)
]


def main():
    import_reference()
    sys.path.insert(0, str(ROOT))
    import cv2                                                   # the stub module of import_reference()
    cv2.INTER_LINEAR = 1
    cv2.BORDER_CONSTANT = 0
    cv2.resize = lambda img, dsize, interpolation=None: image_ref.cv2_resize_linear_u8(img, dsize)
    cv2.copyMakeBorder = lambda img, t, b, l, r, kind, value=(114, 114, 114): image_ref.cv2_copy_make_border(img, t, b, l, r, value[0])
    from ultralytics.data.augment import Format, LetterBox
    from ultralytics.utils.instance import Instances

    g = np.random.default_rng(2024)
    store = {"tags": np.array([c[0] for c in CASES])}
    for tag, (h, w), kw, nb in CASES:
        img = g.integers(0, 256, (h, w, 3), dtype=np.uint8)
        cxy = g.uniform(0.2, 0.8, (nb, 2))
        wh = g.uniform(0.05, 0.35, (nb, 2))
        boxes = np.concatenate((cxy, wh), 1).astype(np.float32)
        cls = g.integers(0, 3, (nb, 1)).astype(np.float32)
        store[f"{tag}.img"], store[f"{tag}.boxes"], store[f"{tag}.cls"] = img, boxes, cls
        lb = LetterBox(**kw)
        store[f"{tag}.image_only"] = lb(image=img.copy())                                   # the predictor's call form
        labels = {"img": img.copy(), "cls": cls.copy(), "ratio_pad": (1.0, 1.0),
                  "instances": Instances(boxes.copy(), np.zeros((0, 1000, 2), np.float32), None, "xywh", True)}
        if tag == "rect":
            labels["rect_shape"] = (48, 64)
        out = lb(labels)
        store[f"{tag}.lb_img"] = out["img"]
        store[f"{tag}.lb_boxes"] = out["instances"].bboxes.copy()
        store[f"{tag}.lb_ratio_pad"] = np.asarray([out["ratio_pad"][0][0], out["ratio_pad"][0][1], *out["ratio_pad"][1]], np.float64)
        store[f"{tag}.resized_shape"] = np.asarray(out["resized_shape"])
        random.seed(7)                                                                      # Format draws the bgr coin
        fm = Format(bbox_format="xywh", normalize=True, batch_idx=True, bgr=0.0)(out)
        store[f"{tag}.fm_img"] = fm["img"].numpy()
        store[f"{tag}.fm_boxes"] = fm["bboxes"].numpy()
        store[f"{tag}.fm_cls"] = fm["cls"].numpy()
        store[f"{tag}.fm_batch_idx"] = fm["batch_idx"].numpy()
    np.savez_compressed(OUT / "image.npz", **store)
    print("wrote", OUT / "image.npz", sum(v.nbytes for v in store.values()), "bytes raw")


if __name__ == "__main__":
    main()
