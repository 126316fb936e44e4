"""CPU: image side of preprocess.  (1) the oracle (oracle/image_ref.py) reproduces the golden LetterBox images recorded
from the REFERENCE's LetterBox control flow (tests/golden/image.npz; the cv2 pixels inside those are the oracle's
restatement — parity unpinned against a real cv2, see oracle/image_ref.py); (2) the product's host-side geometry and
label arithmetic (LetterBox.geometry / _update_labels, Format, Instances — numpy, no kernel) reproduce the reference's
numbers bit for bit; (3) known answers of the restated 8-bit bilinear."""
import random

import numpy as np
import pytest
import torch

from oracle import image_ref as IR
from oracle.gen_golden_image import CASES
from tests._golden import load


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_oracle_letterbox_matches_reference_flow(case):
    tag, _, kw, _ = case
    gold = load("image.npz")
    img, ratio, pad = IR.letterbox(gold[f"{tag}.img"], **kw)
    assert np.array_equal(img, gold[f"{tag}.image_only"])


def test_bus_geometry():
    """bus.jpg is 810 W x 1080 H -> letterboxed 640 H x 480 W on the predictor's auto path."""
    from sy11.data.augment import LetterBox
    new_unpad, ratio, top, bottom, left, right = LetterBox((640, 640), auto=True, stride=32).geometry((1080, 810))
    assert new_unpad == (480, 640) and (top, bottom, left, right) == (0, 0, 0, 0) and ratio == (640 / 1080, 640 / 1080)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_product_geometry_labels_and_format_match_reference(case):
    from sy11.data.augment import Format, LetterBox
    from sy11.utils.instance import Instances
    tag, (h, w), kw, nb = case
    gold = load("image.npz")
    lb = LetterBox(**kw)
    new_shape = (48, 64) if tag == "rect" else None
    new_unpad, ratio, top, bottom, left, right = lb.geometry((h, w), new_shape)
    out_img = gold[f"{tag}.lb_img"]
    assert out_img.shape[:2] == (new_unpad[1] + top + bottom, new_unpad[0] + left + right)
    rp = gold[f"{tag}.lb_ratio_pad"]
    assert (left, top) == (int(rp[2]), int(rp[3]))
    labels = {"cls": gold[f"{tag}.cls"].copy(), "instances": Instances(gold[f"{tag}.boxes"].copy(), bbox_format="xywh", normalized=True)}
    labels = LetterBox._update_labels(labels, ratio, left, top, (h, w))
    assert np.array_equal(labels["instances"].bboxes, gold[f"{tag}.lb_boxes"])
    labels["img"] = out_img                                                    # Format on the reference's letterboxed image
    random.seed(7)
    fm = Format(bbox_format="xywh", normalize=True, batch_idx=True, bgr=0.0)(labels)
    assert np.array_equal(fm["img"].numpy(), gold[f"{tag}.fm_img"])
    assert np.array_equal(fm["bboxes"].numpy(), gold[f"{tag}.fm_boxes"])
    assert np.array_equal(fm["cls"].numpy(), gold[f"{tag}.fm_cls"])
    assert fm["batch_idx"].shape == gold[f"{tag}.fm_batch_idx"].shape


def test_resize_known_answers():
    # constant image stays constant under every path (general, 2x box mean, upscale)
    for shape, dsize in (((9, 7), (5, 4)), ((8, 6), (3, 4)), ((5, 5), (11, 13))):
        img = np.full((*shape, 3), 201, np.uint8)
        assert (IR.cv2_resize_linear_u8(img, dsize) == 201).all()
    # 2x1 upscale of a two-pixel row: taps at src x = -0.25, 0.25, 0.75, 1.25 -> weights pinned at the borders
    img = np.array([[[0, 0, 0], [200, 100, 40]]], np.uint8)
    out = IR.cv2_resize_linear_u8(img, (4, 1))
    assert out[0, :, 0].tolist() == [0, 50, 150, 200] and out[0, :, 2].tolist() == [0, 10, 30, 40]
    # exact 2x shrink is the rounded 2x2 mean
    img = np.arange(4 * 4 * 3, dtype=np.uint8).reshape(4, 4, 3)
    a = img.astype(int)
    assert np.array_equal(IR.cv2_resize_linear_u8(img, (2, 2)), ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2))
    # identity size returns a copy
    assert np.array_equal(IR.cv2_resize_linear_u8(img, (4, 4)), img)


def test_instances_roundtrip_and_flips():
    from sy11.utils.instance import Instances
    b = np.array([[0.5, 0.5, 0.2, 0.4], [0.25, 0.75, 0.1, 0.1]], np.float32)
    ins = Instances(b.copy(), bbox_format="xywh", normalized=True)
    ins.convert_bbox("xyxy")
    ins.denormalize(100, 50)
    assert np.allclose(ins.bboxes[0], [40, 15, 60, 35]) and not ins.normalized
    ins.fliplr(100)
    assert np.allclose(ins.bboxes[0], [40, 15, 60, 35]) and np.allclose(ins.bboxes[1], [70, 35, 80, 40])
    ins.flipud(50)
    assert np.allclose(ins.bboxes[1], [70, 10, 80, 15])
    ins.add_padding(-75, 0)
    ins.clip(100, 50)
    good = ins.remove_zero_area_boxes()
    assert good.tolist() == [False, True] and len(ins) == 1
    ins.normalize(100, 50)
    ins.convert_bbox("xywh")
    assert ins.normalized and ins.bboxes.shape == (1, 4)
    with pytest.raises(NotImplementedError):
        Instances(b, segments=np.zeros((2, 10, 2), np.float32))
    cat = Instances.concatenate([Instances(b, bbox_format="xywh"), Instances(b[:1], bbox_format="xywh")])
    assert len(cat) == 3


def test_multi_scale_oracle_shapes():
    x = np.random.default_rng(0).integers(0, 256, (2, 3, 64, 96), dtype=np.uint8)
    y = IR.preprocess_batch_multi_scale(x, 128)
    assert tuple(y.shape) == (2, 3, 96, 128) and y.dtype == torch.float32
    assert tuple(IR.preprocess_batch_multi_scale(x, 96).shape) == (2, 3, 64, 96)


def test_bus_jpg_letterbox_matches_reference_flow():
    """BASELINE configs[0] plumbing: the reference's LetterBox(auto) on bus.jpg (1080 x 810 -> 640 x 480); the oracle reproduces
    the recorded letterboxed image byte for byte (sha256) and the preprocessed tensor's samples."""
    import hashlib
    from PIL import Image
    from tests._golden import GOLD, check
    gold = load("bus.npz")
    im0 = np.ascontiguousarray(np.asarray(Image.open(GOLD / "bus.jpg").convert("RGB"))[..., ::-1])
    assert tuple(gold["orig_shape"]) == im0.shape == (1080, 810, 3)
    lb, ratio, pad = IR.letterbox(im0, (640, 640), auto=True, stride=32)
    assert tuple(gold["letterbox_shape"]) == lb.shape == (640, 480, 3) and pad == (0, 0)
    assert hashlib.sha256(np.ascontiguousarray(lb).tobytes()).digest() == gold["letterbox_sha256"].tobytes()
    check(gold, "im", IR.predictor_preprocess([im0], (640, 640), stride=32), rtol=1e-9, atol=0)
