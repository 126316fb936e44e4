"""CPU: the training augmentation chain, host side.  The product's Mosaic / RandomPerspective / RandomHSV / RandomFlip /
Format run lazily (no pixels are rendered, so no GPU is needed) on the generator's in-memory dataset with the same
seeds; boxes, classes and the consumed RNG streams must equal what the REFERENCE's own classes produced
(tests/golden/augment.npz), and the recorded recipe (tiles, inverted map, LUTs, flips) must be what the oracle would use."""
import numpy as np
import pytest
import torch

from oracle import image_ref as IR
from tests._augment_util import IMGSZ, run_pipeline
from tests._golden import load


@pytest.mark.parametrize("name", ["default", "rich"])          # the mosaic configs: no LetterBox launch on this path
def test_labels_and_rng_streams_match_reference(name):
    from sy11.data.augment import DeviceImage
    gold = load("augment.npz")
    n = 0
    for k, out in run_pipeline(gold, name, "cpu"):
        if k is None:
            assert np.array_equal(out, gold[f"{name}.rng_after"])              # same number of draws from both streams
            continue
        assert np.array_equal(out["bboxes"].numpy(), gold[f"{name}.{k}.bboxes"])
        assert np.array_equal(out["cls"].numpy(), gold[f"{name}.{k}.cls"])
        di = out["img"]
        assert isinstance(di, DeviceImage) and di.shape == (IMGSZ, IMGSZ, 3) and di.canvas_hw == (2 * IMGSZ, 2 * IMGSZ)
        assert 1 <= len(di.tiles) <= 4 and di.minv is not None and di.lut is not None and di.lut.shape == (3, 256)
        assert out["batch_idx"].shape[0] == out["cls"].shape[0]
        n += 1
    assert n == 6


def test_recipe_reproduces_golden_pixels_through_the_oracle():
    """Execute the recorded recipe with the oracle's numpy functions: canvas paste -> warp -> HSV -> flips -> CHW/RGB."""
    gold = load("augment.npz")
    for k, out in run_pipeline(gold, "rich", "cpu"):
        if k is None:
            break
        di = out["img"]
        canvas = np.full((*di.canvas_hw, 3), 114, np.uint8)
        for t, x1, y1, x2, y2, pw, ph in di.tiles:
            canvas[y1:y2, x1:x2] = t.numpy()[y1 - ph:y2 - ph, x1 - pw:x2 - pw]
        sx, sy, fx, fy = IR.warp_coords(di.minv, di.out_hw[1], di.out_hw[0])
        src = canvas.astype(np.int64)

        def tap(xx, yy):
            inside = (xx >= 0) & (xx < canvas.shape[1]) & (yy >= 0) & (yy < canvas.shape[0])
            return np.where(inside[..., None], src[np.clip(yy, 0, canvas.shape[0] - 1), np.clip(xx, 0, canvas.shape[1] - 1)], 114)

        w = [((32 - fy) * (32 - fx) * 32)[..., None], ((32 - fy) * fx * 32)[..., None], (fy * (32 - fx) * 32)[..., None], (fy * fx * 32)[..., None]]
        img = ((tap(sx, sy) * w[0] + tap(sx + 1, sy) * w[1] + tap(sx, sy + 1) * w[2] + tap(sx + 1, sy + 1) * w[3] + 16384) >> 15).astype(np.uint8)
        hsv = IR.cv2_bgr2hsv_u8(img)
        img = IR.cv2_hsv2bgr_u8(np.stack((di.lut[0][hsv[..., 0]], di.lut[1][hsv[..., 1]], di.lut[2][hsv[..., 2]]), -1))
        if di.flip_ud:
            img = img[::-1]
        if di.flip_lr:
            img = img[:, ::-1]
        chw = img.transpose(2, 0, 1)
        chw = chw[::-1] if di.final_reverse_c else chw
        assert np.array_equal(chw, gold[f"rich.{k}.img"])


def test_device_image_recipe_rules():
    from sy11.data.augment import DeviceImage, invert_affine
    t = torch.zeros((8, 6, 3), dtype=torch.uint8)
    di = DeviceImage.wrap(t)
    assert di.shape == (8, 6, 3) and di.plain_tensor() is t and not di.pending
    di.flip(lr=True).flip(lr=True)
    assert not di.pending                                              # two flips cancel
    di.warp(np.array([[1, 0, 2], [0, 1, 3]], np.float32), (10, 12))
    assert di.shape == (12, 10, 3) and di.pending and di.plain_tensor() is None
    assert np.allclose(di.minv, [1, 0, -2, 0, 1, -3]) and invert_affine(np.eye(3)[:2]) == [1, 0, 0, 0, 1, 0]
    with pytest.raises(ValueError):
        DeviceImage.wrap(np.zeros((4, 4), np.uint8))


def test_hsv_and_warp_known_answers():
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 0, 0], [128, 128, 128]]], np.uint8)
    assert IR.cv2_bgr2hsv_u8(px)[0].tolist() == [[120, 255, 255], [60, 255, 255], [0, 255, 255], [0, 0, 255], [0, 0, 0], [0, 0, 128]]
    assert np.array_equal(IR.cv2_hsv2bgr_u8(IR.cv2_bgr2hsv_u8(px)), px)                    # primaries and greys are exact
    g = np.random.default_rng(0)
    img = g.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    assert np.array_equal(IR.cv2_warp_affine_u8(img, np.array([[1, 0, 0], [0, 1, 0]], np.float32), (30, 20)), img)
    sh = IR.cv2_warp_affine_u8(img, np.array([[1, 0, 3], [0, 1, -2]], np.float32), (30, 20))
    assert np.array_equal(sh[:18, 3:], img[2:, :27]) and (sh[18:] == 114).all() and (sh[:, :3] == 114).all()
    half = IR.cv2_warp_affine_u8(img, np.array([[1, 0, 0.5], [0, 1, 0]], np.float32), (30, 20))       # half-pixel shift = 2-tap mean
    a = img.astype(int)
    assert np.array_equal(half[:, 1:], ((a[:, :-1] + a[:, 1:]) * 16384 + 16384 >> 15).astype(np.uint8))
    lh, ls, lv = IR.hsv_luts(np.array([1.0, 1.0, 1.0]))
    assert lh[179] == 179 and lh[200] == 20 and ls[255] == 255 and np.array_equal(lv, np.arange(256))


def _write_small(root, n=7, imgsz=32):
    g = np.random.default_rng(1)
    (root / "images").mkdir(parents=True)
    (root / "labels").mkdir()
    for i in range(n):
        h, w = [(imgsz, imgsz), (imgsz, 20), (24, imgsz)][i % 3]
        np.save(root / "images" / f"s{i}.npy", g.integers(0, 256, (h, w, 3), dtype=np.uint8))
        rows = [[i % 2, 0.5, 0.5, 0.2, 0.2], [i % 2, 0.5, 0.5, 0.2, 0.2], [1, 0.3, 0.6, 0.1, 0.2]][: 1 + i % 3]
        (root / "labels" / f"s{i}.txt").write_text("\n".join(" ".join(str(v) for v in r) for r in rows))
    return root


def test_dataset_files_labels_rect_and_sharding(tmp_path):
    from sy11.data.dataset import InfiniteDataLoader, YOLODataset, img2label_paths, read_label
    root = _write_small(tmp_path / "d")
    assert img2label_paths(["/a/images/b/images/x.png"]) == ["/a/images/b/labels/x.txt"]
    ds = YOLODataset(str(root / "images"), imgsz=32, augment=True, batch_size=2, device="cpu", data={"nc": 2})
    assert len(ds) == 7 and [len(l["cls"]) for l in ds.labels] == [1, 1, 2, 1, 1, 2, 1]                # duplicate rows dropped
    assert ds.max_buffer_length == 7 and ds.labels[2]["bboxes"].dtype == np.float32
    lab = ds.get_image_and_label(1)                                                                    # long side == imgsz: no launch
    assert lab["resized_shape"] == (32, 20) and lab["ori_shape"] == (32, 20) and ds.buffer == [1] and len(lab["instances"]) == 1
    with pytest.raises(AssertionError):
        (root / "labels" / "bad.txt").write_text("0 1.5 0.5 0.1 0.1")
        read_label(str(root / "labels" / "bad.txt"))
    with pytest.raises(AssertionError):
        read_label(str(root / "labels" / "s2.txt"), nc=1)
    only1 = YOLODataset(str(root / "images"), imgsz=32, augment=False, batch_size=2, device="cpu", classes=[1], single_cls=True)
    assert sum(len(l["cls"]) for l in only1.labels) == 5 and all((l["cls"] == 0).all() for l in only1.labels)
    rect = YOLODataset(str(root / "images"), imgsz=32, augment=False, rect=True, batch_size=2, device="cpu", stride=8, pad=0.5)
    ar = [np.load(f).shape[0] / np.load(f).shape[1] for f in rect.im_files]
    # sorted aspect ratios [.75 .75 | 1 1 | 1 1.6 | 1.6] -> shapes [.75,1] [1,1] [1,1] [1,.625] -> ceil(s*32/8 + .5)*8
    assert ar == sorted(ar) and rect.batch_shapes.tolist() == [[32, 40], [40, 40], [40, 40], [40, 24]]
    # sharding: two ranks see disjoint halves of one permutation, padded by wrap-around, same batch count
    a = InfiniteDataLoader(ds, 2, shuffle=True, rank=0, world_size=2, prefetch=0)
    b = InfiniteDataLoader(ds, 2, shuffle=True, rank=1, world_size=2, prefetch=0)
    ia, ib = a._epoch_indices(), b._epoch_indices()
    assert len(ia) == len(ib) == 4 and len(a) == len(b) == 2 and set(ia) | set(ib) == set(range(7))
    g = torch.Generator(); g.manual_seed(0)
    perm = torch.randperm(7, generator=g).tolist()
    assert ia == (perm + perm[:1])[0::2] and ib == (perm + perm[:1])[1::2]                             # DistributedSampler's rule
    a.set_epoch(1)
    assert a._epoch_indices() != ia
    single = InfiniteDataLoader(ds, 4, shuffle=False, prefetch=0)
    assert single._epoch_indices() == list(range(7)) and len(single) == 2
