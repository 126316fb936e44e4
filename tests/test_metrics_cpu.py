"""CPU: validation metrics.  (1) the oracle (oracle/metrics_ref.py, loop restatement) reproduces the REFERENCE's numbers in
tests/golden/metrics.npz; (2) the product's host-side metric code (sy11.utils.metrics / DetectionValidator.match_predictions —
numpy, no kernel involved) reproduces them too."""
import numpy as np
import pytest
import torch

from oracle import metrics_ref as MR
from tests._golden import load

IOUV = np.linspace(0.5, 0.95, 10, dtype=np.float32)


def per_image(gold, tag):
    for k in range(int(gold[f"{tag}.n_img"])):
        yield k, gold[f"{tag}.{k}.gt"], gold[f"{tag}.{k}.gcls"], gold[f"{tag}.{k}.det"], gold[f"{tag}.{k}.tp"]


def collect(gold, tag):
    tps, confs, pcls, tcls = [], [], [], []
    for k, gt, gcls, det, tp in per_image(gold, tag):
        if len(det) or len(gt):
            tps.append(tp); confs.append(det[:, 4]); pcls.append(det[:, 5]); tcls.append(gcls)
    return np.concatenate(tps), np.concatenate(confs), np.concatenate(pcls), np.concatenate(tcls)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_oracle_iou_and_matching_match_reference(tag):
    gold = load("metrics.npz")
    n_checked = 0
    for k, gt, gcls, det, tp in per_image(gold, tag):
        if not (len(det) and len(gt)):
            assert not tp.any()
            continue
        iou = MR.box_iou(gt, det[:, :4])
        assert np.array_equal(iou, gold[f"{tag}.{k}.iou"])                     # bit-exact f32 evaluation order
        assert np.array_equal(MR.match_predictions(det[:, 5], gcls, iou, torch.linspace(0.5, 0.95, 10).tolist()), tp)
        n_checked += 1
    assert n_checked >= 5


@pytest.mark.parametrize("tag", ["a", "b"])
def test_oracle_ap_per_class_matches_reference(tag):
    gold = load("metrics.npz")
    res = MR.ap_per_class(*collect(gold, tag))
    assert np.array_equal(res["classes"], gold[f"{tag}.apc.classes"])
    np.testing.assert_allclose(res["ap"], gold[f"{tag}.apc.ap"], rtol=1e-9, atol=1e-12)
    for k in ("p", "r", "f1"):
        np.testing.assert_allclose(res[k], gold[f"{tag}.apc.{k}"], rtol=1e-9, atol=1e-12)
    mp, mr, m50, m, fit = MR.summary(res)
    np.testing.assert_allclose([mp, mr, m50, m], gold[f"{tag}.mean_results"], rtol=1e-9)
    assert abs(fit - float(gold[f"{tag}.fitness"])) < 1e-12
    assert abs(MR.compute_ap(gold["ap.recall"], gold["ap.precision"]) - float(gold["ap.value"])) < 1e-12


@pytest.mark.parametrize("tag", ["a", "b"])
def test_product_metrics_match_reference(tag):
    from sy11.engine.validator import DetectionValidator
    from sy11.utils.metrics import DetMetrics, ap_per_class, box_iou, compute_ap
    gold = load("metrics.npz")
    v = DetectionValidator(device="cpu")
    for k, gt, gcls, det, tp in per_image(gold, tag):
        if len(det) and len(gt):
            iou = box_iou(torch.from_numpy(gt), torch.from_numpy(det[:, :4]))    # torch formulation kept beside the kernel
            assert np.array_equal(iou.numpy(), gold[f"{tag}.{k}.iou"])
            got = v.match_predictions(torch.from_numpy(det[:, 5]), torch.from_numpy(gcls), iou)
            assert np.array_equal(got.numpy(), tp)
    tp, conf, pc, tc = collect(gold, tag)
    res = ap_per_class(tp, conf, pc, tc)
    for name, val in zip(("tpn", "fpn", "p", "r", "f1", "ap", "classes"), res[:7]):
        np.testing.assert_allclose(np.asarray(val, np.float64), gold[f"{tag}.apc.{name}"], rtol=1e-12, atol=0)
    nc = int(gold[f"{tag}.nc"])
    dm = DetMetrics(names={i: str(i) for i in range(nc)})
    dm.process(tp, conf, pc, tc)
    np.testing.assert_allclose(dm.mean_results(), gold[f"{tag}.mean_results"], rtol=1e-12)
    assert abs(dm.fitness - float(gold[f"{tag}.fitness"])) < 1e-14
    np.testing.assert_allclose(dm.maps, gold[f"{tag}.maps"], rtol=1e-12)
    assert list(dm.results_dict) == ["metrics/precision(B)", "metrics/recall(B)", "metrics/mAP50(B)", "metrics/mAP50-95(B)", "fitness"]
    assert abs(compute_ap(gold["ap.recall"], gold["ap.precision"])[0] - float(gold["ap.value"])) < 1e-14


def test_validator_bookkeeping_without_kernels():
    """update_metrics / get_stats plumbing on CPU tensors with the IoU step stubbed by the torch formulation: images with
    no predictions but labels still count their targets; images with neither are skipped (val.py:141-149)."""
    from sy11.engine import validator as V
    from sy11.utils.metrics import box_iou
    gold = load("metrics.npz")
    v = V.DetectionValidator(device="cpu", names={i: str(i) for i in range(5)})
    v.init_metrics(None)
    orig = V.box_iou_device
    V.box_iou_device = box_iou
    try:
        for k, gt, gcls, det, tp in per_image(gold, "a"):
            imgsz = 160
            xyxy = torch.from_numpy(gt)
            xywh = torch.cat(((xyxy[:, :2] + xyxy[:, 2:]) / 2, xyxy[:, 2:] - xyxy[:, :2]), 1) / imgsz if len(gt) else torch.zeros(0, 4)
            batch = {"img": torch.zeros(1, 3, imgsz, imgsz), "batch_idx": torch.zeros(len(gt)), "cls": torch.from_numpy(gcls).view(-1, 1),
                     "bboxes": xywh}
            v.update_metrics([torch.from_numpy(det).clone()], batch)
    finally:
        V.box_iou_device = orig
    stats = v.get_stats()
    assert v.seen == int(gold["a.n_img"])
    np.testing.assert_allclose([stats[k] for k in v.metrics.keys], gold["a.mean_results"], rtol=1e-5)   # boxes went through xywh round trip
    assert v.nt_per_class.sum() == sum(len(g) for _, g, *_ in [(k, gold[f"a.{k}.gcls"]) for k in range(int(gold["a.n_img"]))])
