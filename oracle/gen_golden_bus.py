"""ORACLE — TEST INFRASTRUCTURE ONLY.  Golden vectors for BASELINE.json configs[0]: YOLOv11-n predict on bus.jpg through the
reference's CPU path (the plumbing configuration of tests/test_python.py / tests/__init__.py SOURCE = ASSETS/"bus.jpg").

bus.jpg (810 x 1080, a data file the reference's own tests hold) is copied to tests/golden/bus.jpg as a fixture.  The
REFERENCE's LetterBox (auto=True, stride 32, as BasePredictor.pre_transform builds it), its preprocess arithmetic
(engine/predictor.py:118-136) and its DetectionModel (yolo11n.yaml, nc=80, fused, eval) run on it with seeded weights;
stored: the sha256 of the letterboxed uint8 image, samples + moments of the preprocessed tensor and of the decoded
predictions.  cv2 is absent: the JPEG is decoded with PIL (RGB -> BGR) and cv2.resize / copyMakeBorder are bound to
oracle.image_ref (parity of those pixels against a real cv2: unpinned).  torchvision's NMS is absent too: the kept boxes
are pinned in the tests by the oracle's NMS on identical decoded predictions.

Run:  python -m oracle.gen_golden_bus   ->  tests/golden/bus.npz, tests/golden/bus.jpg
"""
from __future__ import annotations

import hashlib
import shutil
import sys

import numpy as np
import torch

from oracle import image_ref
from oracle.gen_golden import OUT, ROOT, import_reference, summarize
from oracle.yolo11_ref import empty_state_dict, resolve_graph, seeded_state_dict


def main():
    import_reference()
    sys.path.insert(0, str(ROOT))
    import cv2
    cv2.INTER_LINEAR, cv2.BORDER_CONSTANT = 1, 0
    cv2.resize = lambda img, dsize, interpolation=None: image_ref.cv2_resize_linear_u8(img, dsize)
    cv2.copyMakeBorder = lambda img, t, b, l, r, kind, value=(114, 114, 114): image_ref.cv2_copy_make_border(img, t, b, l, r, value[0])
    from PIL import Image
    from ultralytics.data.augment import LetterBox
    from ultralytics.nn.tasks import DetectionModel
    src = "/root/reference/bus.jpg"
    shutil.copyfile(src, OUT / "bus.jpg")
    im0 = np.ascontiguousarray(np.asarray(Image.open(src).convert("RGB"))[..., ::-1])          # what cv2.imread returns: BGR HWC
    store = {"orig_shape": np.asarray(im0.shape)}
    lb = LetterBox((640, 640), auto=True, stride=32)(image=im0)                                 # predictor.pre_transform for one pt image
    store["letterbox_shape"] = np.asarray(lb.shape)
    store["letterbox_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(lb).tobytes()).digest(), np.uint8)
    im = np.ascontiguousarray(np.stack([lb])[..., ::-1].transpose((0, 3, 1, 2)))               # predictor.py:127-130
    t = torch.from_numpy(im).float()
    t /= 255
    summarize(store, "im", t)
    model = DetectionModel("yolo11n.yaml", ch=3, nc=80, verbose=False)
    sd = seeded_state_dict(empty_state_dict(resolve_graph("n", nc=80)), seed=7)
    for k in sd:                                                                                # a confident random head (as tests/test_predict_gpu.py)
        if ".cv3." in k and k.endswith("2.bias"):
            sd[k] = sd[k] + 1.0
    model.load_state_dict(sd)
    model.eval().fuse(verbose=False)
    with torch.no_grad():
        y = model(t)
    y = y[0] if isinstance(y, (list, tuple)) else y
    summarize(store, "pred", y)
    np.savez_compressed(OUT / "bus.npz", **store)
    print("wrote", OUT / "bus.npz", {k: v.shape for k, v in store.items() if hasattr(v, "shape")})


if __name__ == "__main__":
    main()
