"""GPU: predict path — fused eval forward, Detect decode, NMS — kept boxes / classes bit-exact vs the oracle's NMS on the
same decoded predictions, and the decoded predictions within 1e-3 of the oracle forward."""
import numpy as np
import pytest
import torch

from oracle import nms_ref, yolo11_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build(nc=80):
    from sy11.nn.tasks import DetectionModel
    m = DetectionModel("yolo11n.yaml", nc=nc, verbose=False)
    sd = R.seeded_state_dict(R.empty_state_dict(R.resolve_graph("n", nc=nc)), seed=7)
    # confident random head: larger cls biases so that some scores clear the 0.25 threshold
    for k in sd:
        if ".cv3." in k and k.endswith("2.bias"):
            sd[k] = sd[k] + 1.0
    m.load_state_dict(sd)
    return m, sd


def test_predict_matches_oracle_and_nms_is_bit_exact():
    from sy11.engine.predictor import DetectionPredictor
    m, sd = build()
    img = R.seeded_image((2, 3, 160, 128), seed=9)
    pred = DetectionPredictor(m, device=DEV, conf=0.25, iou=0.7)
    im = pred.preprocess(img)
    y, _ = pred.inference(im)
    layers = R.resolve_graph("n", nc=80)
    with torch.no_grad():
        yo, _ = R.forward(R.fuse_state_dict({k: v.clone() for k, v in sd.items()}), layers, img, train=False, fused=True)
    scale = yo.abs().max().item()
    assert (y.cpu() - yo).abs().max().item() <= 1e-3 * scale
    # NMS parity on IDENTICAL inputs (the device's own decoded predictions): kept rows bit-exact
    ref_out, _ = nms_ref.non_max_suppression(y.cpu().clone(), 0.25, 0.7, multi_label=False, max_det=300)
    res = pred.postprocess(y.clone(), im)
    assert sum(len(r) for r in res) > 0
    for r, ro in zip(res, ref_out):
        ro = ro.clone()
        ro[:, :4] = nms_ref.scale_boxes(im.shape[2:], ro[:, :4], im.shape[2:])
        assert r.boxes.data.shape == ro.shape
        assert torch.equal(r.boxes.cls.cpu(), ro[:, 5])                       # class ids bit-exact
        assert torch.equal(r.boxes.data.cpu(), ro)                           # boxes + scores bit-exact


def test_multilabel_nms_matches_oracle():
    from sy11.utils.ops import non_max_suppression
    g = torch.Generator().manual_seed(0)
    A, nc = 2000, 12
    pred = torch.zeros(3, 4 + nc, A)
    pred[:, 0:2] = 50 + 500 * torch.rand(3, 2, A, generator=g)
    pred[:, 2:4] = 10 + 150 * torch.rand(3, 2, A, generator=g)
    pred[:, 4:] = torch.rand(3, nc, A, generator=g) ** 4
    ref, _ = nms_ref.non_max_suppression(pred.clone(), 0.05, 0.6, multi_label=True, max_det=300)
    got = non_max_suppression(pred.clone().to(DEV), 0.05, 0.6, multi_label=True, max_det=300)
    for a, b in zip(got, ref):
        assert torch.equal(a.cpu(), b)


def test_predict_from_iq():
    from oracle import stft_ref as S
    from sy11.data.spectrogram import SpectrogramProducer
    from sy11.engine.predictor import DetectionPredictor
    m, _ = build()
    pred = DetectionPredictor(m, device=DEV, conf=0.05, producer=SpectrogramProducer(DEV))
    res = pred(S.synthetic_iq(1, seed=4))
    assert len(res) == 1 and res[0].boxes.data.shape[1] == 6


def test_forward_graph_replay_is_bit_identical_to_eager():
    """The predictor's third call on a shape captures a forward-only hipGraph; replays must reproduce the eager result
    exactly (same kernels, same order), for new inputs too, and a second shape gets its own graph."""
    from sy11.engine.predictor import DetectionPredictor
    m, _ = build()
    eager = DetectionPredictor(m, device=DEV, conf=0.25, iou=0.7, graphs=False)
    imgs = [R.seeded_image((2, 3, 160, 128), seed=20 + i) for i in range(5)]
    want = [eager.inference(eager.preprocess(im))[0].clone() for im in imgs]
    m.__dict__.pop("_sy11_graph_cfg", None)
    pred = DetectionPredictor(m, device=DEV, conf=0.25, iou=0.7, graphs=True)
    got = [pred.inference(pred.preprocess(im))[0].clone() for im in imgs]
    cfg = m.__dict__["_sy11_graph_cfg"]
    assert len(cfg["entries"]) == 1                                   # calls 3..5 replayed one captured graph
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    other = R.seeded_image((1, 3, 96, 96), seed=3)
    outs = [pred.inference(pred.preprocess(other))[0].clone() for _ in range(4)]
    assert len(cfg["entries"]) == 2 and all(torch.equal(outs[0], o) for o in outs[1:])
    res = pred(imgs[0])
    assert len(res) == 2


def test_bus_jpg_predict_matches_reference_cpu_path():
    """BASELINE configs[0]: YOLOv11-n predict on bus.jpg.  Golden = the REFERENCE's LetterBox + preprocess + fused eval
    DetectionModel on the CPU (oracle/gen_golden_bus.py).  Preprocessed tensor bit-exact, decoded predictions within 1e-3 of
    their scale, kept boxes / class ids after NMS identical to the oracle's NMS on the same predictions."""
    from PIL import Image
    from sy11.engine.predictor import DetectionPredictor
    from tests._golden import GOLD, check, load
    gold = load("bus.npz")
    im0 = np.ascontiguousarray(np.asarray(Image.open(GOLD / "bus.jpg").convert("RGB"))[..., ::-1])      # cv2.imread order
    m, _ = build()
    pred = DetectionPredictor(m, device=DEV, conf=0.25, iou=0.7, imgsz=640)
    im = pred.preprocess([im0])
    assert tuple(im.shape) == (1, 3, 640, 480)
    check(gold, "im", im, rtol=1e-9, atol=0)       # samples exact; the f64 moments only see summation order
    y, _ = pred.inference(im)
    check(gold, "pred", y, rtol=1e-3, atol=1e-3)
    ref_out, _ = nms_ref.non_max_suppression(y.cpu().clone(), 0.25, 0.7, multi_label=False, max_det=300)
    res = pred.postprocess(y.clone(), im, orig_imgs=[im0])
    got = res[0].boxes.data.cpu()
    assert got.shape[0] == ref_out[0].shape[0] and got.shape[0] > 0
    assert torch.equal(got[:, 5], ref_out[0][:, 5]) and torch.equal(got[:, 4], ref_out[0][:, 4])          # class ids and scores of the kept rows
    assert res[0].orig_shape == (1080, 810)


@pytest.mark.parametrize("case", ["empty_image", "agnostic", "max_det_5", "one_class", "single_image_max_nms", "classes_filter"])
def test_nms_wrapper_edge_cases_match_oracle(case):
    """non_max_suppression's device path (r04: candidate selection, one key sort, per-(image, class) bit matrices) against the restated
    wrapper on identical inputs, kept rows bit for bit: an image without a candidate between two that have some, class-agnostic
    suppression, a tiny max_det, a one-class model (multi_label is forced off, utils/ops.py:235), more than max_nms = 30 000
    candidates in one image (the reference keeps the 30 000 best, ops.py:291-292), and a class filter (the tensor-op path)."""
    from sy11.utils.ops import non_max_suppression
    g = torch.Generator().manual_seed(11)
    B, A, nc = (1, 8400, 12) if case == "single_image_max_nms" else (3, 1500, 1 if case == "one_class" else 7)
    pred = torch.zeros(B, 4 + nc, A)
    pred[:, 0:2] = 50 + 500 * torch.rand(B, 2, A, generator=g)
    pred[:, 2:4] = 10 + 150 * torch.rand(B, 2, A, generator=g)
    pred[:, 4:] = torch.rand(B, nc, A, generator=g) ** (1 if case == "single_image_max_nms" else 4)
    kw = dict(conf_thres=0.05, iou_thres=0.6, multi_label=True, max_det=300)
    if case == "empty_image":
        pred[1, 4:] *= 0.01                                   # nothing above the threshold in the middle image
    if case == "agnostic":
        kw["agnostic"] = True
    if case == "max_det_5":
        kw["max_det"] = 5
    if case == "single_image_max_nms":
        kw["conf_thres"] = 0.6                                # 12 x 8400 x 0.4 = 40 000 candidate pairs > max_nms
    ref_kw = {k: v for k, v in kw.items()}
    if case == "classes_filter":
        got = non_max_suppression(pred.clone().to(DEV), classes=[1, 3], **kw)
        ref, _ = nms_ref.non_max_suppression(pred[:, [0, 1, 2, 3, 5, 7]].clone(), nc=2, **ref_kw)     # the two wanted classes alone ...
        for r in ref:
            r[:, 5] = torch.where(r[:, 5] == 0, 1.0, 3.0)                                             # ... under their own ids
    else:
        got = non_max_suppression(pred.clone().to(DEV), **kw)
        ref, _ = nms_ref.non_max_suppression(pred.clone(), **ref_kw)
    assert len(got) == B
    if case == "empty_image":
        assert got[1].shape == (0, 6)
    if case == "single_image_max_nms":
        assert int((pred[0, 4:] > 0.6).sum()) > 30000
    for a, b in zip(got, ref):
        assert a.shape == b.shape, (a.shape, b.shape)
        assert torch.equal(a.cpu(), b)
