"""GPU: image side of preprocess through the C-ABI.  sy11_image_letterbox is bit-identical to the oracle's restated
cv2 LetterBox (and thereby to the golden images recorded from the reference's LetterBox control flow); the fused
predictor form (BGR->RGB, CHW, /255) and sy11_image_u8_to_float are bit-identical in f32; sy11_image_resize_bilinear
matches torch's CPU F.interpolate within 1e-6 (floating point: FMA contraction differs, nothing else)."""
import random

import numpy as np
import pytest
import torch

from oracle import image_ref as IR
from oracle.gen_golden_image import CASES
from tests._golden import load

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_letterbox_matches_golden(case):
    from sy11.data.augment import Format, LetterBox
    from sy11.utils.instance import Instances
    tag, (h, w), kw, nb = case
    gold = load("image.npz")
    lb = LetterBox(**kw)
    out = lb(image=gold[f"{tag}.img"].copy())
    assert isinstance(out, np.ndarray) and np.array_equal(out, gold[f"{tag}.image_only"])
    labels = {"img": gold[f"{tag}.img"].copy(), "cls": gold[f"{tag}.cls"].copy(), "ratio_pad": (1.0, 1.0),
              "instances": Instances(gold[f"{tag}.boxes"].copy(), bbox_format="xywh", normalized=True)}
    if tag == "rect":
        labels["rect_shape"] = (48, 64)
    res = lb(labels)
    assert np.array_equal(res["img"], gold[f"{tag}.lb_img"])
    assert np.array_equal(res["instances"].bboxes, gold[f"{tag}.lb_boxes"])
    assert tuple(res["resized_shape"]) == tuple(gold[f"{tag}.resized_shape"])
    rp = gold[f"{tag}.lb_ratio_pad"]
    assert res["ratio_pad"] == ((rp[0], rp[1]), (int(rp[2]), int(rp[3])))
    random.seed(7)
    fm = Format(bbox_format="xywh", normalize=True, batch_idx=True, bgr=0.0)(res)
    assert np.array_equal(fm["img"].numpy(), gold[f"{tag}.fm_img"]) and np.array_equal(fm["bboxes"].numpy(), gold[f"{tag}.fm_boxes"])


@pytest.mark.parametrize("seed", range(6))
def test_letterbox_random_shapes_bit_exact(seed):
    from sy11.data.augment import LetterBox
    g = np.random.default_rng(seed)
    h, w = int(g.integers(5, 700)), int(g.integers(5, 700))
    new = (int(g.integers(2, 21)) * 32, int(g.integers(2, 21)) * 32)
    kw = dict(new_shape=new, auto=bool(g.integers(0, 2)), scaleup=bool(g.integers(0, 2)), center=bool(g.integers(0, 2)))
    img = g.integers(0, 256, (h, w, 3), dtype=np.uint8)
    want, ratio, pad = IR.letterbox(img, **kw)
    dev_img = torch.from_numpy(img).to(DEV)
    got = LetterBox(**kw)(image=dev_img)                       # device in, device out
    assert got.is_cuda and np.array_equal(got.cpu().numpy(), want)


def test_letterbox_full_size_and_device_tensor_roundtrip():
    """The bus.jpg geometry at full size: 1080 x 810 -> 640 x 480 (auto), and a 1280 x 1280 -> 640 x 640 exact 2x shrink."""
    from sy11.data.augment import LetterBox
    g = np.random.default_rng(3)
    for (h, w), kw, shape in (((1080, 810), dict(new_shape=(640, 640), auto=True), (640, 480, 3)),
                              ((1280, 1280), dict(new_shape=(640, 640)), (640, 640, 3))):
        img = g.integers(0, 256, (h, w, 3), dtype=np.uint8)
        out = LetterBox(**kw)(image=img)
        assert out.shape == shape and np.array_equal(out, IR.letterbox(img, **kw)[0])


@pytest.mark.parametrize("same", [True, False])
def test_predictor_pre_transform_matches_reference_flow(same):
    """pre_transform + preprocess (predictor.py:118-163): list of BGR HWC -> (B, 3, H, W) RGB float / 255, bit-exact."""
    from sy11.engine.predictor import DetectionPredictor
    g = np.random.default_rng(5)
    shapes = [(108, 81)] * 3 if same else [(108, 81), (64, 64), (33, 97)]
    ims = [g.integers(0, 256, (*s, 3), dtype=np.uint8) for s in shapes]
    pred = DetectionPredictor.__new__(DetectionPredictor)      # preprocessing only: no model needed
    pred.device, pred.imgsz, pred.producer = torch.device(DEV), (64, 64), None
    pred.model = type("M", (), {"stride": torch.tensor([8.0, 16.0, 32.0]), "_sy11_dtype": torch.float32})()
    got = pred.preprocess(ims)
    want = IR.predictor_preprocess(ims, (64, 64), stride=32)
    assert got.dtype == torch.float32 and tuple(got.shape) == tuple(want.shape) == (3, 3, 64, 64)
    assert torch.equal(got.cpu(), want)
    half = pred.pre_transform(ims, out_dtype=torch.float16)
    assert torch.equal(half.cpu(), IR.predictor_preprocess(ims, (64, 64), stride=32).half())


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("n", [0, 3, 4, 1021, 64 * 3 * 64 * 64])
def test_u8_to_float_bit_exact(dtype, n):
    from sy11 import ops as K
    x = torch.randint(0, 256, (n,), dtype=torch.uint8, device=DEV)
    got = K.image_u8_to_float(x, dtype)
    assert torch.equal(got.cpu(), (x.cpu().float() / 255).to(dtype))


@pytest.mark.parametrize("src_u8", [True, False])
@pytest.mark.parametrize("ihw,ohw", [((64, 96), (96, 128)), ((64, 64), (32, 32)), ((40, 72), (64, 96)), ((50, 50), (50, 50)), ((7, 5), (32, 64))])
def test_resize_bilinear_vs_torch_cpu(src_u8, ihw, ohw):
    from sy11 import ops as K
    g = torch.Generator().manual_seed(1)
    x8 = torch.randint(0, 256, (2, 3, *ihw), dtype=torch.uint8, generator=g)
    xf = x8.float() / 255
    want = torch.nn.functional.interpolate(xf, size=ohw, mode="bilinear", align_corners=False)
    got = K.image_resize_bilinear((x8 if src_u8 else xf).to(DEV), ohw)
    assert got.dtype == torch.float32 and torch.allclose(got.cpu(), want, rtol=0, atol=1e-6)   # tolerance: fp32 FMA contraction only
    got16 = K.image_resize_bilinear(xf.to(DEV), ohw, dtype=torch.float16)
    assert torch.allclose(got16.cpu().float(), want, rtol=0, atol=1e-3)


def test_trainer_preprocess_batch_multi_scale():
    """preprocess_batch with multi_scale=True (train.py:57-74): the drawn size, the ceil-to-stride shape, the pixels."""
    from sy11.engine.trainer import DetectionTrainer
    tr = DetectionTrainer.__new__(DetectionTrainer)
    tr.device, tr.producer = torch.device(DEV), None
    tr.args = type("A", (), {"multi_scale": True, "imgsz": 64})()
    tr.model = type("M", (), {"stride": torch.tensor([8.0, 16.0, 32.0]), "training": False, "__dict__": {}})()
    x = np.random.default_rng(0).integers(0, 256, (2, 3, 64, 96), dtype=np.uint8)
    seen = set()
    for seed in range(6):
        random.seed(seed)
        sz = random.randrange(32, 128) // 32 * 32                       # same draw as the trainer's
        random.seed(seed)
        got = tr.preprocess_batch({"img": torch.from_numpy(x)})["img"]
        want = IR.preprocess_batch_multi_scale(x, sz, stride=32)
        assert tuple(got.shape) == tuple(want.shape) and torch.allclose(got.cpu(), want, rtol=0, atol=1e-6)
        seen.add(tuple(got.shape[2:]))
    assert len(seen) >= 2
    tr.args.multi_scale = False
    got = tr.preprocess_batch({"img": torch.from_numpy(x)})["img"]
    assert torch.equal(got.cpu(), torch.from_numpy(x).float() / 255)


def test_letterbox_rejects_bad_arguments():
    from sy11 import _lib, ops as K
    src = torch.zeros((8, 8, 3), dtype=torch.uint8, device=DEV)
    with pytest.raises(_lib.Sy11Error):
        K.image_letterbox(src, torch.empty((3, 4, 4), device=DEV), (8, 8), 0, 0)          # does not fit the canvas
    with pytest.raises(_lib.Sy11Error):
        K.image_letterbox(src.float(), torch.empty((3, 8, 8), device=DEV), (8, 8), 0, 0)  # not uint8
    with pytest.raises(_lib.Sy11Error):
        K.image_u8_to_float(torch.zeros(4, device=DEV))
