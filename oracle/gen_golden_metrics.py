"""ORACLE — TEST INFRASTRUCTURE ONLY.  Golden vectors for the validation metrics.

Drives the REFERENCE's box_iou / BaseValidator.match_predictions / ap_per_class / DetMetrics with seeded synthetic
detections and labels and stores inputs + outputs.  Run:  python -m oracle.gen_golden_metrics -> tests/golden/metrics.npz
"""
from __future__ import annotations

import sys
from types import SimpleNamespace

import numpy as np
import torch

from oracle.gen_golden import OUT, ROOT, import_reference


def synth(seed, n_img=12, nc=5, imgsz=160):
    """Per image: a few ground-truth boxes and detections = jittered copies of some labels + random false positives."""
    g = np.random.default_rng(seed)
    images = []
    for _ in range(n_img):
        nl = int(g.integers(0, 5))
        cxy = g.uniform(30, imgsz - 30, (nl, 2))
        wh = g.uniform(16, 60, (nl, 2))
        gt = np.concatenate((cxy - wh / 2, cxy + wh / 2), 1).astype(np.float32)
        gcls = g.integers(0, nc, nl).astype(np.float32)
        dets = []
        for b, c in zip(gt, gcls):
            for _ in range(int(g.integers(0, 4))):
                jit = g.normal(0, 4.0, 4).astype(np.float32)
                cls = c if g.uniform() < 0.8 else float(g.integers(0, nc))
                dets.append(np.concatenate((b + jit, [g.uniform(0.05, 1.0), cls])))
        for _ in range(int(g.integers(0, 4))):
            c0 = g.uniform(20, imgsz - 20, 2)
            s = g.uniform(10, 50, 2)
            dets.append(np.concatenate((c0 - s / 2, c0 + s / 2, [g.uniform(0.01, 0.6), float(g.integers(0, nc))])))
        det = np.array(dets, np.float32).reshape(-1, 6)
        images.append((gt, gcls, det))
    return images


def main():
    import_reference()
    sys.path.insert(0, str(ROOT))
    from ultralytics.engine.validator import BaseValidator
    from ultralytics.utils.metrics import DetMetrics, ap_per_class, box_iou, compute_ap
    store = {}
    iouv = torch.linspace(0.5, 0.95, 10)
    fake = SimpleNamespace(iouv=iouv)
    for tag, seed, nc in (("a", 11, 5), ("b", 23, 3)):
        images = synth(seed, nc=nc)
        tps, confs, pcls, tcls = [], [], [], []
        store[f"{tag}.n_img"] = np.asarray(len(images))
        store[f"{tag}.nc"] = np.asarray(nc)
        for k, (gt, gcls, det) in enumerate(images):
            store[f"{tag}.{k}.gt"], store[f"{tag}.{k}.gcls"], store[f"{tag}.{k}.det"] = gt, gcls, det
            tp = np.zeros((len(det), 10), bool)
            if len(det) and len(gt):
                iou = box_iou(torch.from_numpy(gt), torch.from_numpy(det[:, :4]))
                store[f"{tag}.{k}.iou"] = iou.numpy()
                tp = BaseValidator.match_predictions(fake, torch.from_numpy(det[:, 5]), torch.from_numpy(gcls), iou).numpy()
            store[f"{tag}.{k}.tp"] = tp
            if len(det) or len(gt):
                tps.append(tp); confs.append(det[:, 4]); pcls.append(det[:, 5]); tcls.append(gcls)
        tp, conf, pc, tc = np.concatenate(tps), np.concatenate(confs), np.concatenate(pcls), np.concatenate(tcls)
        res = ap_per_class(tp, conf, pc, tc)
        for name, v in zip(("tpn", "fpn", "p", "r", "f1", "ap", "classes"), res[:7]):
            store[f"{tag}.apc.{name}"] = np.asarray(v)
        dm = DetMetrics(names={i: str(i) for i in range(nc)})
        dm.process(tp, conf, pc, tc)
        store[f"{tag}.mean_results"] = np.asarray(dm.mean_results(), np.float64)
        store[f"{tag}.fitness"] = np.asarray(dm.fitness, np.float64)
        store[f"{tag}.maps"] = np.asarray(dm.maps, np.float64)
    g = np.random.default_rng(5)
    rec = np.sort(g.uniform(0, 1, 40))
    pre = np.clip(1 - rec + g.normal(0, 0.1, 40), 0, 1)
    store["ap.recall"], store["ap.precision"] = rec, pre
    store["ap.value"] = np.asarray(compute_ap(rec, pre)[0])
    np.savez_compressed(OUT / "metrics.npz", **store)
    print("metrics.npz", (OUT / "metrics.npz").stat().st_size)


if __name__ == "__main__":
    main()
