"""GPU: the reference's only end-to-end engine test (tests/test_engine.py:28-64 `test_detect`: DetectionTrainer one epoch ->
DetectionValidator -> DetectionPredictor -> resume), on this build's surface and on files written by the test: dataset
+ fused augmentation -> trainer (AMP, graphs, schedule) -> checkpoint -> reload -> validator over the rect val loader ->
predictor on raw HWC images -> resume training from the checkpoint's optimizer state."""
import io
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _dataset(root, n, imgsz, seed):
    g = np.random.default_rng(seed)
    (root / "images").mkdir(parents=True)
    (root / "labels").mkdir()
    for i in range(n):
        h, w = [(imgsz, imgsz), (imgsz, imgsz * 3 // 4), (imgsz // 2, imgsz)][i % 3]
        img = np.full((h, w, 3), 30, np.uint8)
        rows = []
        for _ in range(int(g.integers(1, 4))):                       # bright rectangles = class 0, dark = class 1
            cx, cy = g.uniform(0.3, 0.7, 2)
            bw, bh = g.uniform(0.15, 0.35, 2)
            c = int(g.integers(0, 2))
            x1, x2 = int((cx - bw / 2) * w), int((cx + bw / 2) * w)
            y1, y2 = int((cy - bh / 2) * h), int((cy + bh / 2) * h)
            img[y1:y2, x1:x2] = 220 if c == 0 else 110
            rows.append((c, cx, cy, bw, bh))
        np.save(root / "images" / f"s{i:03d}.npy", img)
        (root / "labels" / f"s{i:03d}.txt").write_text("\n".join(" ".join(f"{v:.6f}" for v in r) for r in rows))
    return str(root / "images")


def test_train_val_predict_resume_flow(tmp_path):
    from sy11.data.dataset import YOLODataset, build_dataloader, read_image
    from sy11.engine.checkpoint import attempt_load_one_weight, load_checkpoint, save_checkpoint
    from sy11.engine.predictor import DetectionPredictor
    from sy11.engine.trainer import DetectionTrainer
    from sy11.engine.validator import DetectionValidator
    from sy11.nn.tasks import DetectionModel
    S, B = 128, 8
    train_dir = _dataset(tmp_path / "train", 24, S, 0)
    val_dir = _dataset(tmp_path / "val", 9, S, 1)
    data = {"names": {0: "bright", 1: "dark"}}
    torch.manual_seed(0); random.seed(0); np.random.seed(0)
    model = DetectionModel("yolo11n.yaml", nc=2, verbose=False)
    tr = DetectionTrainer(model, batch_size=B, device=DEV, overrides={"amp": True, "nbs": B, "imgsz": S, "warmup_epochs": 1.0}, graphs=True)
    ds = YOLODataset(train_dir, imgsz=S, augment=True, batch_size=B, data=data)
    dl = build_dataloader(ds, B, workers=2, out=tr.batch_buffer(S), dtype=torch.float32)
    epochs = 3
    tr.set_schedule(len(dl), epochs)
    losses = []
    for epoch in range(epochs):
        if epoch == epochs - 1:
            ds.close_mosaic(ds.hyp)                                   # trainer.py:341-343: the last epochs train without mosaic
        for batch in dl:
            losses.append(float(tr.train_step(batch)[0]))
        tr.end_epoch()
    assert len(losses) == epochs * len(dl) and all(np.isfinite(losses))
    # ---- checkpoint -> fresh model
    buf = io.BytesIO()
    save_checkpoint(buf, trainer=tr, epoch=epochs - 1)
    ckpt_bytes = buf.getvalue()
    loaded, ckpt = attempt_load_one_weight(io.BytesIO(ckpt_bytes), device=DEV)
    assert ckpt["epoch"] == epochs - 1 and ckpt["updates"] == tr.ema.updates and ckpt["optimizer"] is not None
    # ---- validation over the rect loader (uint8 batches, per-batch shapes, ratio_pad bookkeeping)
    vds = YOLODataset(val_dir, imgsz=S, augment=False, rect=True, batch_size=4, pad=0.5, stride=32, data=data)
    vdl = build_dataloader(vds, 4, workers=0, shuffle=False)
    stats = DetectionValidator(loaded, device=DEV)(loaded, list(vdl))
    assert stats and all(np.isfinite(float(v)) for v in stats.values())
    # ---- prediction on raw HWC BGR images of different shapes (LetterBox on the device)
    raw = [read_image(f) for f in vds.im_files[:3]]
    pred = DetectionPredictor(loaded, device=DEV, conf=0.001, imgsz=S)
    for im in raw:
        res = pred([im])
        assert len(res) == 1 and res[0].orig_shape == im.shape[:2]
        if len(res[0]):
            bx = res[0].boxes.xyxy
            assert float(bx.min()) >= -1 and float(bx[:, [0, 2]].max()) <= im.shape[1] + 1 and float(bx[:, [1, 3]].max()) <= im.shape[0] + 1
    # ---- resume: a new trainer takes weights and optimizer state from the checkpoint and keeps training
    ck = load_checkpoint(io.BytesIO(ckpt_bytes))
    model2 = DetectionModel("yolo11n.yaml", nc=2, verbose=False)
    model2.load_state_dict(ck["ema"].float().state_dict())
    tr2 = DetectionTrainer(model2, batch_size=B, device=DEV, overrides={"amp": True, "nbs": B, "imgsz": S}, graphs=True)
    more = [float(tr2.train_step(batch)[0]) for batch in dl]
    assert all(np.isfinite(more)) and np.mean(more) < 3 * np.mean(losses[-len(dl):]) + 1e-6      # continues from a trained state, not from scratch
