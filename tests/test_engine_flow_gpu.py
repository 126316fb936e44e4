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


def _tiny_batch(seed):
    g = torch.Generator().manual_seed(seed)
    return {"img": torch.rand(4, 3, 64, 64, generator=g).to(DEV), "batch_idx": torch.tensor([0., 1., 2., 3.]).to(DEV),
            "cls": torch.tensor([[1.], [0.], [1.], [0.]]).to(DEV), "bboxes": (0.3 + 0.3 * torch.rand(4, 4, generator=g)).to(DEV)}


@pytest.mark.parametrize("opt", ["SGD", "AdamW"])
def test_resume_from_flat_and_from_reference_layout_optimizer_state(opt):
    """resume_training (engine/trainer.py:728-756): optimizer state + EMA + epoch.  (a) a checkpoint of this build (flat
    groups, fp16-converted state as the reference stores it) resumes into the same buffers; (b) a per-parameter state_dict
    in the reference's layout lands in the same flat momentum buffers as training in the flat layout produced."""
    from oracle import yolo11_ref as R
    from sy11.engine.checkpoint import load_checkpoint, save_checkpoint
    from sy11.engine.trainer import DetectionTrainer
    from sy11.nn.tasks import DetectionModel
    sd = R.seeded_state_dict(R.empty_state_dict(R.resolve_graph("n", nc=2)), seed=4)

    def make(flat):
        m = DetectionModel("yolo11n.yaml", nc=2, verbose=False)
        m.load_state_dict(sd)
        return DetectionTrainer(m, batch_size=4, device=DEV, overrides={"amp": False, "nbs": 4, "optimizer": opt, "lr0": 1e-3}, graphs=False, flat=flat)

    a, b = make(True), make(False)                       # flat groups vs the reference's per-parameter optimizer
    for i in range(3):
        a.train_step(_tiny_batch(i))
        b.train_step(_tiny_batch(i))
    key = "momentum_buffer" if opt == "SGD" else "exp_avg"
    flat_mom = [a.optimizer.state[g["params"][0]][key].clone() for g in a.optimizer.param_groups]
    # (b) reference layout -> flat
    c = make(True)
    c.resume_training({"epoch": 4, "optimizer": b.optimizer.state_dict(), "ema": None})
    assert c.epoch == 5
    for g, want in zip(c.optimizer.param_groups, flat_mom):
        got = c.optimizer.state[g["params"][0]][key]
        # two independent 3-step trainings (f32 atomics reorder; AdamW amplifies it): same layout <=> the vectors line up
        cos = float(torch.dot(got, want) / (got.norm() * want.norm() + 1e-30))
        assert got.shape == want.shape and cos > (0.999 if opt == "SGD" else 0.97), cos
    # (a) this build's checkpoint (state stored in fp16) -> a fresh flat trainer
    buf = io.BytesIO()
    save_checkpoint(buf, trainer=a, epoch=2)
    ck = load_checkpoint(io.BytesIO(buf.getvalue()))
    d = make(True)
    assert d.resume_training(ck) == 3 and d.ema.updates == a.ema.updates
    for g, want in zip(d.optimizer.param_groups, flat_mom):
        got = d.optimizer.state[g["params"][0]][key]
        assert torch.allclose(got, want.half().float(), rtol=1e-3, atol=1e-6)
    for (k1, p), (_, q) in zip(a.ema.ema.state_dict().items(), d.ema.ema.state_dict().items()):
        if p.dtype.is_floating_point:
            assert torch.allclose(p.half().float(), q, rtol=1e-3, atol=1e-4), k1
    loss = float(d.train_step(_tiny_batch(9))[0])          # and it keeps training
    assert np.isfinite(loss)


def test_fit_loop_validates_saves_and_closes_mosaic(tmp_path):
    """DetectionTrainer.fit = the order of `_do_train` (trainer.py:318-474): iterations, scheduler step, EMA validation,
    fitness, last.pt / best.pt, mosaic closed for the last epochs; the saved best.pt loads and predicts."""
    from sy11.data.dataset import YOLODataset, build_dataloader
    from sy11.engine.checkpoint import attempt_load_one_weight
    from sy11.engine.trainer import DetectionTrainer
    from sy11.nn.tasks import DetectionModel
    S, B = 96, 6
    data = {"names": {0: "bright", 1: "dark"}}
    train_dir = _dataset(tmp_path / "train", 12, S, 3)
    val_dir = _dataset(tmp_path / "val", 6, S, 4)
    torch.manual_seed(1); random.seed(1); np.random.seed(1)
    tr = DetectionTrainer(DetectionModel("yolo11n.yaml", nc=2, verbose=False), batch_size=B, device=DEV,
                          overrides={"amp": True, "nbs": B, "imgsz": S, "warmup_epochs": 0.5}, graphs=True)
    ds = YOLODataset(train_dir, imgsz=S, augment=True, batch_size=B, data=data)
    dl = build_dataloader(ds, B, workers=0, out=tr.batch_buffer(S), dtype=torch.float32)
    vds = YOLODataset(val_dir, imgsz=S, augment=False, rect=True, batch_size=3, pad=0.5, stride=32, data=data)
    vdl = build_dataloader(vds, 3, workers=0, shuffle=False)
    hist = tr.fit(dl, epochs=3, val_batches=lambda: list(vdl), save_dir=tmp_path / "run", close_mosaic=1)
    assert [h["epoch"] for h in hist] == [0, 1, 2] and all(h["fitness"] is not None and len(h["train_loss"]) == 3 for h in hist)
    assert ds.hyp.mosaic == 0.0                                           # closed before the last epoch
    assert (tmp_path / "run" / "last.pt").exists() and (tmp_path / "run" / "best.pt").exists()
    lr_now = tr.optimizer.param_groups[1]["lr"]
    assert 0 < lr_now < tr.optimizer.param_groups[1]["initial_lr"]         # the linear schedule moved
    model, ck = attempt_load_one_weight(str(tmp_path / "run" / "best.pt"), device=DEV)
    assert ck["epoch"] in (0, 1, 2) and ck["train_metrics"] is not None
    with torch.no_grad():
        y, _ = model(torch.rand(1, 3, S, S, device=DEV))
    n_anchors = (S // 8) ** 2 + (S // 16) ** 2 + (S // 32) ** 2
    assert tuple(y.shape) == (1, 4 + 2, n_anchors) and torch.isfinite(y).all()


def test_yolo_front_door_train_val_predict(tmp_path):
    """`YOLO(cfg).train(data=yaml) / .val() / .predict()` — the reference's entry points (engine/model.py) on a data YAML."""
    from sy11 import YOLO
    S = 96
    _dataset(tmp_path / "ds" / "train", 12, S, 5)
    _dataset(tmp_path / "ds" / "val", 6, S, 6)
    (tmp_path / "ds" / "data.yaml").write_text("path: .\ntrain: train/images\nval: val/images\nnames:\n  0: bright\n  1: dark\n")
    y = YOLO("yolo11n.yaml", device=DEV)
    hist = y.train(data=str(tmp_path / "ds" / "data.yaml"), epochs=2, batch=6, imgsz=S, workers=0, save_dir=tmp_path / "run", close_mosaic=1,
                   warmup_epochs=0.5, fliplr=0.0)
    assert len(hist) == 2 and y.model.names == {0: "bright", 1: "dark"} and y.ckpt is not None          # continues with best.pt
    m = y.val(data=str(tmp_path / "ds" / "data.yaml"), batch=3, imgsz=S)
    assert "metrics/mAP50(B)" in m or "fitness" in m
    img = np.load(sorted((tmp_path / "ds" / "val" / "images").glob("*.npy"))[0])
    res = y.predict(img, conf=0.001, imgsz=S)
    assert len(res) == 1 and res[0].orig_shape == img.shape[:2]
    res2 = y([str(p) for p in sorted((tmp_path / "ds" / "val" / "images").glob("*.npy"))[:2]], conf=0.001, imgsz=S)
    assert len(res2) == 2 and res2[0].path.endswith(".npy")
    with pytest.raises(SyntaxError):
        y.train(data={"train": "x"}, epochs=1)
    with pytest.raises(SyntaxError):
        y.train(data=str(tmp_path / "ds" / "data.yaml"), epochs=1, not_an_argument=1)
