"""CPU: the loader's worker processes (sy11.data.dataset.WorkerLoader, data/recipe.py).  A worker never touches the GPU: what it
sends back per sample is a recipe over LazyImage nodes plus the labels.  Checked here without a GPU: the recipes of one worker equal
what the same dataset produces in-process under the worker's seed (same draws, same boxes, same tile geometry), a sample pickles to
a few KB, batches arrive in order from several workers, and close_mosaic reaches the workers."""
import pickle
import random

import numpy as np
import pytest
import torch


def make_dataset(tmp_path, n=12, size=64, device="cpu"):
    from sy11.data.dataset import YOLODataset
    g = np.random.default_rng(0)
    (tmp_path / "images").mkdir()
    (tmp_path / "labels").mkdir()
    for i in range(n):
        np.save(tmp_path / "images" / f"s{i:03d}.npy", g.integers(0, 256, (size - 8 * (i % 3), size, 3), dtype=np.uint8))
        rows = np.concatenate((g.integers(0, 2, (3, 1)), g.uniform(0.3, 0.7, (3, 2)), g.uniform(0.1, 0.4, (3, 2))), 1)
        (tmp_path / "labels" / f"s{i:03d}.txt").write_text("\n".join(" ".join(f"{v:.6f}" for v in r) for r in rows))
    return YOLODataset(str(tmp_path / "images"), imgsz=size, augment=True, batch_size=4, data={"nc": 2}, device=device)


def recipe_signature(sample):
    img = sample["img"]
    tiles = [(t[0].op if t[0].op[0] == "file" else (t[0].op[0], t[0].op[1].op, *t[0].op[2:]), t[0].shape, *t[1:]) for t in img.tiles]
    return (tiles, img.canvas_hw, img.out_hw, None if img.minv is None else tuple(img.minv), None if img.lut is None else img.lut.tobytes(),
            img.flip_ud, img.flip_lr, img.final_reverse_c, sample["bboxes"].numpy().tobytes(), sample["cls"].numpy().tobytes())


def test_one_worker_reproduces_the_in_process_recipes(tmp_path):
    from sy11.data.augment import DeviceImage
    from sy11.data.dataset import WorkerLoader
    from sy11.data.recipe import LazyImage
    ds = make_dataset(tmp_path)
    dl = WorkerLoader(ds, 4, procs=1, shuffle=False, seed=3)
    try:
        it = dl._recipes()
        got = [next(it) for _ in range(4)]
    finally:
        dl.close()
    # the same thing in this process: a pickled copy of the dataset in recipe mode under the worker's seed
    twin = pickle.loads(pickle.dumps(ds))
    twin.recipe_mode = True
    seed = 1000003 * (3 + 1)
    random.seed(seed); np.random.seed(seed % 2**32); torch.manual_seed(seed)
    want = [[twin[i] for i in range(k, k + 4)] for k in (0, 4, 8)] + [[twin[i] for i in range(0, 4)]]
    for a, b in zip(got, want):
        assert [recipe_signature(x) for x in a] == [recipe_signature(x) for x in b]
    s = got[0][0]
    assert isinstance(s["img"], DeviceImage) and all(isinstance(t[0], LazyImage) for t in s["img"].tiles) and len(s["img"].tiles) >= 1
    assert len(pickle.dumps(s)) < 8000                           # a recipe, not an image (the rendered sample would be 12 KB even at 64 x 64)
    with pytest.raises(RuntimeError):
        s["img"].render(chw=True)                                # pixels only exist after resolve() in the training process


def test_batches_arrive_in_order_from_several_workers_and_close_mosaic_reaches_them(tmp_path):
    from sy11.data.dataset import WorkerLoader, build_dataloader
    ds = make_dataset(tmp_path)
    dl = build_dataloader(ds, 4, workers=3, shuffle=False, procs=3)
    assert isinstance(dl, WorkerLoader) and dl.procs == 3
    try:
        it = dl._recipes()
        files = [[s["im_file"] for s in next(it)] for _ in range(6)]
        order = [ds.im_files[i] for i in range(12)]
        assert [f for b in files[:3] for f in b] == order and [f for b in files[3:] for f in b] == order      # two epochs, batch order kept
        assert all(len(s["img"].tiles) > 1 for s in next(it))                                               # mosaic on: several tiles
        ds.close_mosaic(ds.hyp)                                                                             # ... off everywhere, stale batches dropped
        it = dl._recipes()
        batch = next(it)
        assert all(len(s["img"].tiles) == 1 for s in batch)
    finally:
        dl.close()
    assert all(not p.is_alive() for p in dl.workers)
    # rect / validation datasets and workers <= 1 stay in-process
    from sy11.data.dataset import InfiniteDataLoader
    assert type(build_dataloader(ds, 4, workers=1)) is InfiniteDataLoader


def test_unguarded_main_script_is_not_rerun_in_the_workers(tmp_path):
    """a spawned worker re-imports the parent's main script; a script without an `if __name__ == '__main__'` guard
    (fine with the reference's fork workers) then ran its top level again inside every worker, the worker died in its bootstrap and
    the training process waited forever.  The workers are started with `__main__` hidden from the spawn bootstrap: the very same
    unguarded script must finish, and its top level must have run exactly once."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    make_dataset(tmp_path)
    marker = tmp_path / "runs.txt"
    script = tmp_path / "train_unguarded.py"
    script.write_text(
        "import sys\n"
        f"sys.path.insert(0, {str(root / 'spectrogram-yolov11_amd')!r})\n"
        f"open({str(marker)!r}, 'a').write('x')\n"                                      # top level: once per (re-)execution
        "from sy11.data.dataset import WorkerLoader, YOLODataset\n"
        f"ds = YOLODataset({str(tmp_path / 'images')!r}, imgsz=64, augment=True, batch_size=4, data={{'nc': 2}}, device='cpu')\n"
        "dl = WorkerLoader(ds, 4, procs=2, shuffle=False, seed=1)\n"
        "it = dl._recipes()\n"
        "n = sum(len(next(it)) for _ in range(3))\n"
        "dl.close()\n"
        "print('samples', n)\n")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "samples 12" in r.stdout
    assert marker.read_text() == "x"


def test_a_dead_worker_raises_instead_of_hanging(tmp_path):
    from sy11.data.dataset import WorkerLoader
    ds = make_dataset(tmp_path)
    dl = WorkerLoader(ds, 4, procs=2, shuffle=False, seed=1)
    try:
        it = dl._recipes()
        next(it)
        for p in dl.workers:
            p.kill()                                             # exact PIDs this loader started
        for p in dl.workers:
            p.join(timeout=10)
        with pytest.raises(RuntimeError, match="died"):
            for _ in range(16):                                  # whatever was already queued is delivered; then the loss is noticed
                next(it)
    finally:
        dl.close()


def test_materializer_cache_is_bounded_by_bytes():
    """the source cache was bounded by entry count only (raw decoded images of any resolution in HBM).  It now also
    carries a byte budget: the oldest entries go first, an image larger than the whole budget is not cached at all."""
    from sy11.data.recipe import Materializer, file_image
    one = 100 * 100 * 3
    m = Materializer(lambda i: torch.zeros(100, 100, 3, dtype=torch.uint8), "cpu", capacity=1000, max_bytes=10 * one)
    for i in range(50):
        m(file_image(i, (100, 100)))
        assert m.bytes <= m.max_bytes and m.bytes == sum(v.numel() for v in m.cache.values())
    assert 1 <= len(m.cache) <= 10 and file_image(49, (100, 100)).key() in m.cache          # the newest survives, the oldest went
    big = Materializer(lambda i: torch.zeros(400, 400, 3, dtype=torch.uint8), "cpu", max_bytes=one)
    big(file_image(0, (400, 400)))
    assert len(big.cache) == 0 and big.bytes == 0
