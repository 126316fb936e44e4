"""GPU: the fused augmentation render (sy11_image_mosaic_warp) and the dataset / collate path through the C-ABI.
Pixels, boxes and classes of every sample equal the goldens recorded from the REFERENCE's v8_transforms + Format run
(tests/golden/augment.npz: reference control flow, restated cv2 pixels — see oracle/image_ref.py), bit for bit."""
import random

import numpy as np
import pytest
import torch

from oracle import image_ref as IR
from tests._augment_util import CONFIGS, IMGSZ, run_pipeline
from tests._golden import load

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("name", list(CONFIGS))
def test_fused_render_matches_reference_pipeline(name):
    gold = load("augment.npz")
    n = 0
    for k, out in run_pipeline(gold, name, DEV):
        if k is None:
            assert np.array_equal(out, gold[f"{name}.rng_after"])
            continue
        di = out["img"]
        got = di.render(chw=True, reverse_c=di.final_reverse_c)
        assert got.is_cuda and got.dtype == torch.uint8
        assert np.array_equal(got.cpu().numpy(), gold[f"{name}.{k}.img"]), f"{name} sample {k}"
        assert np.array_equal(out["bboxes"].numpy(), gold[f"{name}.{k}.bboxes"]) and np.array_equal(out["cls"].numpy(), gold[f"{name}.{k}.cls"])
        # the float form folds preprocess_batch's /255 into the same launch
        slot = torch.empty((3, IMGSZ, IMGSZ), device=DEV)
        di.render(dst=slot, chw=True, reverse_c=di.final_reverse_c)
        assert torch.equal(slot.cpu(), torch.from_numpy(gold[f"{name}.{k}.img"]).float() / 255)
        n += 1
    assert n == 6


@pytest.mark.parametrize("seed", range(4))
def test_each_step_alone_vs_oracle(seed):
    """numpy in -> numpy out, one transform at a time (the drop-in form), against the oracle's sequential functions."""
    from sy11.data.augment import DeviceImage, RandomFlip, RandomHSV
    g = np.random.default_rng(seed)
    img = g.integers(0, 256, (int(g.integers(9, 90)), int(g.integers(9, 90)), 3), dtype=np.uint8)
    M = np.array([[g.uniform(0.5, 1.5), g.uniform(-0.3, 0.3), g.uniform(-9, 9)], [g.uniform(-0.3, 0.3), g.uniform(0.5, 1.5), g.uniform(-9, 9)]], np.float32)
    dsize = (int(g.integers(8, 100)), int(g.integers(8, 100)))
    got = DeviceImage.wrap(img).warp(M, dsize).unwrap()
    assert isinstance(got, np.ndarray) and np.array_equal(got, IR.cv2_warp_affine_u8(img, M, dsize))
    np.random.seed(seed)
    r = np.random.uniform(-1, 1, 3) * [0.5, 0.9, 0.9] + 1
    np.random.seed(seed)
    out = RandomHSV(0.5, 0.9, 0.9)({"img": img.copy()})["img"]
    assert np.array_equal(out, IR.random_hsv(img, r))
    from sy11.utils.instance import Instances
    lab = {"img": img.copy(), "instances": Instances(np.array([[0.5, 0.5, 0.2, 0.2]], np.float32))}
    out = RandomFlip(p=1.0, direction="vertical")(lab)
    out = RandomFlip(p=1.0, direction="horizontal")(out)
    assert np.array_equal(out["img"], img[::-1, ::-1])
    # a second warp after HSV + flip cannot fold into one pass: the recipe flattens and still equals the sequence
    di = DeviceImage.wrap(torch.from_numpy(img).to(DEV)).hsv(IR.hsv_luts(r)).flip(lr=True).warp(M, dsize)
    want = IR.cv2_warp_affine_u8(np.ascontiguousarray(IR.random_hsv(img, r)[:, ::-1]), M, dsize)
    assert np.array_equal(di.render().cpu().numpy(), want)


def test_full_size_sample_vs_oracle_and_properties():
    """BASELINE size (imgsz 640, 1280 x 1280 canvas): one mosaic sample through the product transforms vs the oracle's
    sequential numpy chain, plus size-independent properties of the render (identity map, flip involution, tile coverage)."""
    from types import SimpleNamespace
    from sy11.data.augment import DeviceImage, Format, v8_transforms
    from sy11.utils.instance import Instances
    S = 640
    g = np.random.default_rng(11)
    shapes = [(640, 480), (640, 640), (400, 640), (640, 512)]
    imgs = [g.integers(0, 256, (*sh, 3), dtype=np.uint8) for sh in shapes]

    class DS:
        data, use_keypoints, buffer = {}, False, [0, 1, 2, 3]

        def __len__(self):
            return 4

        def get_image_and_label(self, i):
            h, w = imgs[i].shape[:2]
            return {"im_file": str(i), "ori_shape": (h, w), "resized_shape": (h, w), "img": DeviceImage.wrap(torch.from_numpy(imgs[i]).to(DEV)),
                    "cls": np.zeros((1, 1), np.float32), "ratio_pad": (1.0, 1.0),
                    "instances": Instances(np.array([[0.5, 0.5, 0.3, 0.3]], np.float32))}

    hyp = SimpleNamespace(hsv_h=0.015, hsv_s=0.7, hsv_v=0.4, degrees=7.0, translate=0.1, scale=0.5, shear=1.0, perspective=0.0, flipud=0.5,
                          fliplr=0.5, bgr=0.0, mosaic=1.0, mixup=0.0, copy_paste=0.0)
    ds = DS()
    tf = v8_transforms(ds, S, hyp)
    tf.append(Format(defer=True))
    random.seed(5); np.random.seed(5)
    out = tf(ds.get_image_and_label(0))
    di = out["img"]
    got = di.render(chw=True, reverse_c=di.final_reverse_c).cpu().numpy()
    # oracle: paste -> warpAffine -> HSV -> flips -> CHW / RGB, from the recorded recipe
    canvas = np.full((2 * S, 2 * S, 3), 114, np.uint8)
    for t, x1, y1, x2, y2, pw, ph in di.tiles:
        canvas[y1:y2, x1:x2] = t.cpu().numpy()[y1 - ph:y2 - ph, x1 - pw:x2 - pw]
    sx, sy, fx, fy = IR.warp_coords(di.minv, S, S)
    src = canvas.astype(np.int64)

    def tap(xx, yy):
        inside = (xx >= 0) & (xx < 2 * S) & (yy >= 0) & (yy < 2 * S)
        return np.where(inside[..., None], src[np.clip(yy, 0, 2 * S - 1), np.clip(xx, 0, 2 * S - 1)], 114)

    w = [((32 - fy) * (32 - fx) * 32)[..., None], ((32 - fy) * fx * 32)[..., None], (fy * (32 - fx) * 32)[..., None], (fy * fx * 32)[..., None]]
    img = ((tap(sx, sy) * w[0] + tap(sx + 1, sy) * w[1] + tap(sx, sy + 1) * w[2] + tap(sx + 1, sy + 1) * w[3] + 16384) >> 15).astype(np.uint8)
    hsv = IR.cv2_bgr2hsv_u8(img)
    img = IR.cv2_hsv2bgr_u8(np.stack((di.lut[0][hsv[..., 0]], di.lut[1][hsv[..., 1]], di.lut[2][hsv[..., 2]]), -1))
    img = img[::-1] if di.flip_ud else img
    img = img[:, ::-1] if di.flip_lr else img
    want = img.transpose(2, 0, 1)
    want = want[::-1] if di.final_reverse_c else want
    assert got.shape == (3, S, S) and np.array_equal(got, want)
    # properties: identity map = the canvas itself; two flips = nothing; every canvas pixel is a tile pixel or 114
    plain = DeviceImage(di.tiles, di.canvas_hw)
    base = plain.render().cpu().numpy()
    assert np.array_equal(base, canvas)
    ident = DeviceImage(di.tiles, di.canvas_hw).warp(np.array([[1, 0, 0], [0, 1, 0]], np.float32), (2 * S, 2 * S)).render().cpu().numpy()
    assert np.array_equal(ident, canvas)
    twice = DeviceImage(di.tiles, di.canvas_hw).flip(ud=True, lr=True).render()
    back = DeviceImage.wrap(twice).flip(ud=True, lr=True).render().cpu().numpy()
    assert np.array_equal(back, canvas) and np.array_equal(twice.cpu().numpy(), canvas[::-1, ::-1])


def test_full_hsv_cube_vs_oracle():
    """All 2^24 BGR values through BGR->HSV->LUT->BGR in one launch each way of the LUT: integer + float paths bit-exact."""
    from sy11 import ops as K
    v = np.arange(256, dtype=np.uint8)
    for gains in ((1.0, 1.0, 1.0), (1.013, 0.41, 1.37)):
        lut = np.stack(IR.hsv_luts(np.array(gains)))
        for b0 in (0, 77, 200, 255):                                   # 4 slabs of the cube: 4 x 65536 colours
            img = np.stack(np.meshgrid(np.array([b0], np.uint8), v, v, indexing="ij"), -1).reshape(256, 256, 3)
            src = torch.from_numpy(np.ascontiguousarray(img)).to(DEV)
            dst = torch.empty((256, 256, 3), dtype=torch.uint8, device=DEV)
            K.image_mosaic_warp([(src, 0, 0, 256, 256, 0, 0)], (256, 256), dst, hsv_lut=lut, chw=False)
            assert np.array_equal(dst.cpu().numpy(), IR.random_hsv(img, np.array(gains)))


def _write_dataset(root, n=10, seed=0, imgsz=64):
    from PIL import Image
    g = np.random.default_rng(seed)
    (root / "images").mkdir(parents=True)
    (root / "labels").mkdir()
    for i in range(n):
        h, w = [(imgsz, imgsz), (imgsz, 40), (48, imgsz), (100, 128), (32, 20)][i % 5]
        img = g.integers(0, 256, (h, w, 3), dtype=np.uint8)
        if i % 2:
            np.save(root / "images" / f"s{i:02d}.npy", img)
        else:
            Image.fromarray(img[..., ::-1]).save(root / "images" / f"s{i:02d}.png")      # PNG stores RGB; the dataset returns BGR
        nb = int(g.integers(0, 4))
        rows = np.concatenate((g.integers(0, 2, (nb, 1)), g.uniform(0.3, 0.7, (nb, 2)), g.uniform(0.1, 0.3, (nb, 2))), 1)
        if nb and i % 3 == 0:
            rows = np.concatenate((rows, rows[:1]))                                        # a duplicate row: must be dropped
        (root / "labels" / f"s{i:02d}.txt").write_text("\n".join(" ".join(f"{v:.6f}" for v in r) for r in rows))
    return root


def test_dataset_train_batches_render_into_static_input(tmp_path):
    from sy11.data.dataset import YOLODataset, build_dataloader
    root = _write_dataset(tmp_path / "d")
    ds = YOLODataset(str(root / "images"), imgsz=64, augment=True, batch_size=4, data={"names": {0: "lte", 1: "nr"}})
    assert len(ds) == 10 and ds.labels[0]["bboxes"].shape[1] == 4
    im, hw0, hw = ds.load_image(3)                                                         # 100 x 128 -> long side 64
    raw = np.load(root / "images" / "s03.npy")
    assert hw0 == (100, 128) and hw == (50, 64) and np.array_equal(im.cpu().numpy(), IR.cv2_resize_linear_u8(raw, (64, 50)))
    static = torch.zeros((4, 3, 64, 64), device=DEV)                                       # stands for a graph's static input
    random.seed(3); np.random.seed(3)
    dl = build_dataloader(ds, 4, workers=2, shuffle=True, out=static, dtype=torch.float32)
    assert len(dl) == 3
    seen = 0
    for batch in dl:
        assert batch["img"].data_ptr() == static.data_ptr() and tuple(batch["img"].shape) in ((4, 3, 64, 64), (2, 3, 64, 64))
        assert 0.0 <= float(batch["img"].min()) and float(batch["img"].max()) <= 1.0
        nl = batch["cls"].shape[0]
        assert batch["bboxes"].shape == (nl, 4) and batch["batch_idx"].shape == (nl,)
        if nl:
            assert float(batch["bboxes"].min()) >= 0 and float(batch["bboxes"].max()) <= 1 and int(batch["batch_idx"].max()) < batch["img"].shape[0]
        seen += 1
        if batch["img"].shape[0] < 4:
            break
    assert seen >= 2


def test_train_from_files_renders_into_the_graph_input(tmp_path):
    """files -> dataset -> fused augmentation -> trainer: after graph capture the loader writes the model's static input."""
    from oracle import yolo11_ref as R
    from sy11.data.dataset import YOLODataset, build_dataloader
    from sy11.engine import graph_static_input
    from sy11.engine.trainer import DetectionTrainer
    from sy11.nn.tasks import DetectionModel
    root = _write_dataset(tmp_path / "d", n=8, imgsz=128)
    m = DetectionModel("yolo11n.yaml", nc=2, verbose=False)
    m.load_state_dict(R.seeded_state_dict(R.empty_state_dict(R.resolve_graph("n", nc=2)), seed=1))
    tr = DetectionTrainer(m, batch_size=4, device=DEV, overrides={"amp": False, "nbs": 4, "imgsz": 128}, graphs=True)
    ds = YOLODataset(str(root / "images"), imgsz=128, augment=True, batch_size=4, data={"nc": 2})
    random.seed(0); np.random.seed(0)
    dl = build_dataloader(ds, 4, workers=2, out=tr.batch_buffer(128), dtype=torch.float32)
    losses, hits = [], 0
    for epoch in range(4):
        for batch in dl:
            static = graph_static_input(tr.model, (4, 3, 128, 128))
            hits += int(static is not None and batch["img"].data_ptr() == static.data_ptr())
            batch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
            losses.append(float(tr.train_step(batch)[0]))
    assert len(losses) == 8 and all(np.isfinite(losses)) and hits >= 4, (losses, hits)


def test_dataset_val_rect_letterbox_matches_oracle(tmp_path):
    from sy11.data.dataset import YOLODataset
    root = _write_dataset(tmp_path / "d", n=5)
    ds = YOLODataset(str(root / "images"), imgsz=64, augment=False, rect=True, batch_size=2, pad=0.5, stride=32)
    assert ds.batch_shapes.shape == (3, 2) and (ds.batch_shapes % 32 == 0).all()
    for i in range(len(ds)):
        s = ds[i]
        di = s["img"]
        got = di.render(chw=True, reverse_c=di.final_reverse_c).cpu().numpy()
        raw = ds.im_files[i]
        from sy11.data.dataset import read_image
        im0 = read_image(raw)
        h0, w0 = im0.shape[:2]
        r = 64 / max(h0, w0)
        im1 = IR.cv2_resize_linear_u8(im0, (min(int(np.ceil(w0 * r)), 64), min(int(np.ceil(h0 * r)), 64))) if r != 1 else im0
        want, _, _ = IR.letterbox(im1, tuple(int(v) for v in ds.batch_shapes[ds.batch[i]]), scaleup=False)
        assert np.array_equal(got, want.transpose(2, 0, 1)[::-1])
    b = ds.collate_fn([ds[0], ds[1]])
    assert b["img"].dtype == torch.uint8 and b["img"].shape[0] == 2 and b["img"].is_cuda


def test_validator_over_rect_val_loader(tmp_path):
    """val mode end to end: rect batches from files (uint8, per-batch shapes) -> DetectionValidator -> metric dict; labels
    are mapped back to native image space through ori_shape / ratio_pad exactly as val.py:108-128 does."""
    from oracle import yolo11_ref as R
    from sy11.data.dataset import YOLODataset, build_dataloader
    from sy11.engine.validator import DetectionValidator
    from sy11.nn.tasks import DetectionModel
    root = _write_dataset(tmp_path / "d", n=10, imgsz=128)
    ds = YOLODataset(str(root / "images"), imgsz=128, augment=False, rect=True, batch_size=4, pad=0.5, stride=32, data={"nc": 2})
    dl = build_dataloader(ds, 4, workers=0, shuffle=False)
    batches = list(dl)
    assert len(batches) == 3 and all(b["img"].dtype == torch.uint8 and b["img"].is_cuda for b in batches)
    assert len({tuple(b["img"].shape[2:]) for b in batches}) >= 2                          # rect: shapes differ per batch
    b0 = batches[0]
    assert len(b0["ori_shape"]) == b0["img"].shape[0] and len(b0["ratio_pad"][0]) == 2    # ((rh, rw), (padw, padh))
    m = DetectionModel("yolo11n.yaml", nc=2, verbose=False)
    m.load_state_dict(R.seeded_state_dict(R.empty_state_dict(R.resolve_graph("n", nc=2)), seed=1))
    v = DetectionValidator(m, device=DEV)
    stats = v(m, batches)
    assert isinstance(stats, dict) and stats and all(np.isfinite(float(x)) for x in stats.values())


def test_mosaic_warp_rejects_bad_arguments():
    from sy11 import _lib, ops as K
    src = torch.zeros((8, 8, 3), dtype=torch.uint8, device=DEV)
    dst = torch.empty((3, 16, 16), dtype=torch.uint8, device=DEV)
    with pytest.raises(_lib.Sy11Error):
        K.image_mosaic_warp([(src, 0, 0, 9, 8, 0, 0)], (16, 16), dst)                       # region reads outside its source
    with pytest.raises(_lib.Sy11Error):
        K.image_mosaic_warp([(src, 10, 10, 18, 18, 10, 10)], (16, 16), dst)                 # region leaves the canvas
    with pytest.raises(_lib.Sy11Error):
        K.image_mosaic_warp([(src, 0, 0, 8, 8, 0, 0)], (8, 8), dst)                         # no warp: output must be the canvas
    with pytest.raises(_lib.Sy11Error):
        K.image_mosaic_warp([(src, 0, 0, 8, 8, 0, 0)] * 5, (16, 16), dst)
