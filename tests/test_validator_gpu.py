"""GPU: validation path.  sy11_box_iou is bit-identical to the reference's IoU matrices (tests/golden/metrics.npz); the
validator's postprocess (multi-label NMS on the HIP kernels) + TP assignment reproduces the oracle pipeline
(oracle.nms_ref + oracle.metrics_ref) exactly on the reference's NMS fixture; the full __call__ loop runs on a model."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import metrics_ref as MR, nms_ref, yolo11_ref as R
from tests._golden import load

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_box_iou_kernel_bit_exact_vs_reference():
    from sy11.utils.metrics import box_iou_device
    gold = load("metrics.npz")
    n = 0
    for tag in ("a", "b"):
        for k in range(int(gold[f"{tag}.n_img"])):
            if f"{tag}.{k}.iou" in gold:
                got = box_iou_device(torch.from_numpy(gold[f"{tag}.{k}.gt"]).to(DEV), torch.from_numpy(gold[f"{tag}.{k}.det"][:, :4]).to(DEV))
                assert np.array_equal(got.cpu().numpy(), gold[f"{tag}.{k}.iou"])
                n += 1
    assert n >= 10
    # bigger, random: against the torch formulation evaluated on the CPU
    from sy11.utils.metrics import box_iou
    g = torch.Generator().manual_seed(0)
    a = torch.rand(300, 2, generator=g) * 500
    a = torch.cat((a, a + torch.rand(300, 2, generator=g) * 120 + 1), 1)
    b = torch.rand(777, 2, generator=g) * 500
    b = torch.cat((b, b + torch.rand(777, 2, generator=g) * 120 + 1), 1)
    assert torch.equal(box_iou_device(a.to(DEV), b.to(DEV)).cpu(), box_iou(a, b))
    assert box_iou_device(torch.zeros(0, 4, device=DEV), b.to(DEV)).shape == (0, 777)


def labels_for(pred, seed, nc):
    """Synthetic labels: some of the strongest predictions' boxes (jittered) with their best class, plus random boxes."""
    g = np.random.default_rng(seed)
    out = []
    for b in range(pred.shape[0]):
        p = pred[b].T.numpy()                        # (A, 4+nc)
        top = np.argsort(-p[:, 4:].max(1))[:6]
        rows = []
        for i in top[g.uniform(size=6) < 0.7]:
            cx, cy, w, h = p[i, :4] + g.normal(0, 2.0, 4)
            rows.append((b, int(p[i, 4:].argmax()), cx, cy, abs(w), abs(h)))
        rows.append((b, int(g.integers(0, nc)), *g.uniform(100, 500, 2), *g.uniform(30, 90, 2)))
        out += rows
    return np.array(out, np.float32)


def test_validator_postprocess_and_tp_match_oracle_pipeline():
    from sy11.engine.validator import DetectionValidator
    gold = load("nms_inputs.npz")
    pred = torch.from_numpy(gold["pred"])            # (2, 4+6, 600) decoded head output of the reference NMS fixture
    nc, imgsz = pred.shape[1] - 4, 640
    lab = labels_for(pred, 3, nc)
    batch = {"img": torch.zeros(pred.shape[0], 3, imgsz, imgsz), "batch_idx": torch.from_numpy(lab[:, 0]),
             "cls": torch.from_numpy(lab[:, 1:2]), "bboxes": torch.from_numpy(lab[:, 2:6] / imgsz)}
    v = DetectionValidator(device=DEV, conf=0.05, iou=0.7, names={i: str(i) for i in range(nc)})
    v.init_metrics(None)
    b = v.preprocess(dict(batch))
    dets = v.postprocess(pred.to(DEV))
    v.update_metrics([d.clone() for d in dets], b)
    stats = v.get_stats()
    # oracle pipeline on the CPU
    odets, _ = nms_ref.non_max_suppression(pred.clone(), conf_thres=0.05, iou_thres=0.7, multi_label=True, max_det=300)
    tps, confs, pcls, tcls = [], [], [], []
    iouv = torch.linspace(0.5, 0.95, 10).tolist()
    for si, od in enumerate(odets):
        assert torch.equal(dets[si].cpu(), od), "multi-label NMS output differs from the oracle"
        m = lab[:, 0] == si
        xywh = torch.from_numpy(lab[m, 2:6] / imgsz)
        gt = torch.cat((xywh[:, :2] - xywh[:, 2:] / 2, xywh[:, :2] + xywh[:, 2:] / 2), 1) * imgsz
        gcls = lab[m, 1]
        tp = MR.match_predictions(od[:, 5].numpy(), gcls, MR.box_iou(gt.numpy(), od[:, :4].numpy()), iouv)
        assert np.array_equal(v.stats["tp"][si].cpu().numpy(), tp), f"image {si}: TP matrix differs"
        tps.append(tp); confs.append(od[:, 4].numpy()); pcls.append(od[:, 5].numpy()); tcls.append(gcls)
    res = MR.ap_per_class(np.concatenate(tps), np.concatenate(confs), np.concatenate(pcls), np.concatenate(tcls))
    mp, mr, m50, m, fit = MR.summary(res)
    np.testing.assert_allclose([stats[k] for k in v.metrics.keys], [mp, mr, m50, m], rtol=1e-9)
    assert abs(stats["fitness"] - fit) < 1e-12 and v.seen == pred.shape[0]


def test_validator_call_runs_model_end_to_end():
    from sy11.engine.validator import DetectionValidator
    from sy11.nn.tasks import DetectionModel
    m = DetectionModel("yolo11n.yaml", nc=4, verbose=False)
    m.args = SimpleNamespace(box=7.5, cls=0.5, dfl=1.5)
    torch.manual_seed(0)
    batches = []
    for s in range(2):
        g = torch.Generator().manual_seed(s)
        batches.append({"img": torch.rand(3, 3, 128, 128, generator=g), "batch_idx": torch.tensor([0., 0., 2.]),
                        "cls": torch.tensor([[1.], [3.], [0.]]), "bboxes": torch.tensor([[0.4, 0.4, 0.3, 0.3], [0.6, 0.6, 0.2, 0.4], [0.5, 0.5, 0.5, 0.5]])})
    v = DetectionValidator(m, device=DEV, conf=0.001)
    stats = v(batches=batches)
    assert v.seen == 6 and v.nt_per_class.sum() == 6
    assert set(stats) == {"metrics/precision(B)", "metrics/recall(B)", "metrics/mAP50(B)", "metrics/mAP50-95(B)", "fitness"}
    assert all(np.isfinite(float(x)) and 0.0 <= float(x) <= 1.0 for x in stats.values())
    assert m.training                                   # the validator restores the mode it found
