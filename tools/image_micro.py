"""Throughput of the image-side kernels at the BASELINE shape (640 x 640, batch 64) and of the whole data path
(dataset -> fused augmentation -> batch tensor), with the CPU oracle timed beside it.
  python tools/image_micro.py [--loader]"""
import sys
import tempfile
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
import numpy as np
import torch
from sy11 import ops as K


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    B, S = 64, 640
    g = np.random.default_rng(0)
    x8 = torch.randint(0, 256, (B, 3, S, S), dtype=torch.uint8, device="cuda")
    y = torch.empty((B, 3, S, S), device="cuda")
    ms = timeit(lambda: K.image_u8_to_float(x8, out=y))
    by = x8.numel() + y.numel() * 4
    print(f"u8_to_float      {B}x3x{S}x{S}: {ms * 1e3:8.1f} us  {by / ms / 1e6:7.0f} GB/s  ({by / 1e6:.0f} MB)")
    y2 = torch.empty((B, 3, 960, 960), device="cuda")
    ms = timeit(lambda: K.image_resize_bilinear(x8, (960, 960), out=y2))
    by = x8.numel() + y2.numel() * 4
    print(f"resize_bilinear  640->960 u8->f32: {ms * 1e3:8.1f} us  {by / ms / 1e6:7.0f} GB/s  ({by / 1e6:.0f} MB)")
    from sy11.data.augment import LetterBox
    raw = [torch.from_numpy(g.integers(0, 256, (1080, 810, 3), dtype=np.uint8)).cuda() for _ in range(B)]
    batch = torch.empty((B, 3, 640, 480), device="cuda")
    lb = LetterBox((640, 640), auto=True)
    ms = timeit(lambda: [lb.into(r, batch[i]) for i, r in enumerate(raw)], reps=5)
    by = sum(r.numel() for r in raw) + batch.numel() * 4
    print(f"letterbox        {B} x 1080x810->640x480 f32 CHW: {ms * 1e3:8.1f} us/batch  {by / ms / 1e6:7.0f} GB/s  {B / ms * 1e3:8.0f} img/s (incl. host launch loop)")
    one = timeit(lambda: K.image_letterbox(raw[0], batch[0], (640, 480), 0, 0, 114, True, True), reps=50)
    print(f"  single launch {one * 1e3:6.1f} us  {(raw[0].numel() + batch[0].numel() * 4) / one / 1e6:6.0f} GB/s")
    from sy11.data.augment import invert_affine
    tiles = [torch.from_numpy(g.integers(0, 256, (S, S, 3), dtype=np.uint8)).cuda() for _ in range(4)]
    xc = yc = 600
    T = [(tiles[0], xc - S if xc > S else 0, yc - S if yc > S else 0, xc, yc, xc - S, yc - S), (tiles[1], xc, 0, min(xc + S, 2 * S), yc, xc, yc - S),
         (tiles[2], 0, yc, xc, min(2 * S, yc + S), xc - S, yc), (tiles[3], xc, yc, min(xc + S, 2 * S), min(2 * S, yc + S), xc, yc)]
    T = [(t, max(x1, 0), max(y1, 0), x2, y2, pw, ph) for t, x1, y1, x2, y2, pw, ph in T]
    M = np.array([[0.9, 0.05, -250.0], [-0.05, 0.9, -260.0]], np.float32)
    lut = np.stack([np.arange(256) % 180, np.clip(np.arange(256) * 1.2, 0, 255), np.clip(np.arange(256) * 0.9, 0, 255)]).astype(np.uint8)
    dst = torch.empty((3, S, S), device="cuda")
    one = timeit(lambda: K.image_mosaic_warp(T, (2 * S, 2 * S), dst, minv=invert_affine(M), hsv_lut=lut, flip_lr=True, reverse_c=True), reps=50)
    by = S * S * 3 + dst.numel() * 4                                   # one source pixel per output pixel (scale ~1) + the float write
    print(f"mosaic_warp      4 tiles -> 3x{S}x{S} f32 (warp+hsv+flip): {one * 1e3:6.1f} us  {by / one / 1e6:6.0f} GB/s  {1e3 / one:8.0f} img/s (incl. host launch)")
    if "--loader" in sys.argv:
        from sy11.data.dataset import YOLODataset, build_dataloader
        root = Path(tempfile.mkdtemp()) / "d"
        (root / "images").mkdir(parents=True); (root / "labels").mkdir()
        n = 512
        for i in range(n):
            np.save(root / "images" / f"s{i:04d}.npy", g.integers(0, 256, (S, S, 3), dtype=np.uint8))
            rows = np.concatenate((g.integers(0, 2, (4, 1)), g.uniform(0.3, 0.7, (4, 2)), g.uniform(0.05, 0.3, (4, 2))), 1)
            (root / "labels" / f"s{i:04d}.txt").write_text("\n".join(" ".join(f"{v:.6f}" for v in r) for r in rows))
        ds = YOLODataset(str(root / "images"), imgsz=S, augment=True, batch_size=B, data={"nc": 2})
        static = torch.empty((B, 3, S, S), device="cuda")
        dl = build_dataloader(ds, B, workers=int(__import__("os").environ.get("SY11_LOADER_THREADS", "8")), out=static, dtype=torch.float32)
        it = iter(dl)
        next(it)
        torch.cuda.synchronize(); t0 = time.perf_counter(); nb = 0
        for batch in it:
            nb += 1
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"loader           mosaic train pipeline, {nb} batches of {B}: {nb * B / dt:8.0f} img/s ({dt / nb * 1e3:.1f} ms/batch, files on tmpfs/page cache)")
        # where the host time goes: recipe building (python) vs the render launches
        samples = []
        t0 = time.perf_counter()
        for i in range(256):
            samples.append(ds[i])
        t1 = time.perf_counter()
        for k in range(0, 256, B):
            ds.collate_fn(samples[k:k + B], out=static, dtype=torch.float32)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"  host split     recipe {1e6 * (t1 - t0) / 256:6.0f} us/sample   collate+render {1e6 * (t2 - t1) / 256:6.0f} us/sample")
        import cProfile, pstats, io
        pr = cProfile.Profile(); pr.enable()
        for i in range(128):
            ds[i]
        pr.disable()
        buf = io.StringIO(); pstats.Stats(pr, stream=buf).sort_stats("tottime").print_stats(14)
        print("\n".join(l[:150] for l in buf.getvalue().splitlines() if l.strip())[:3000])
        if "--train" in sys.argv:
            # end to end: files -> loader -> fused augmentation into the graph's static input -> yolo11s train step (f16)
            from sy11.engine.trainer import DetectionTrainer
            from sy11.nn.tasks import DetectionModel
            model = DetectionModel("yolo11s.yaml", nc=2, verbose=False)
            tr = DetectionTrainer(model, batch_size=B, device="cuda", overrides={"amp": True}, graphs=True)
            dl2 = build_dataloader(ds, B, workers=8, out=tr.batch_buffer(S), dtype=torch.float32)
            it2 = iter(dl2)
            for _ in range(5):                                   # eager warm-up + tuner + graph capture
                tr.train_step(next(it2))
            torch.cuda.synchronize(); t0 = time.perf_counter(); nb = 0
            for epoch in range(3):
                for batch in dl2:
                    tr.train_step(batch); nb += 1
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print(f"train from files yolo11s f16, {nb} steps of {B}: {nb * B / dt:8.0f} img/s ({dt / nb * 1e3:.1f} ms/step; synthetic-data step 23.7-24.6 ms)")
        # CPU oracle on a few samples: the same chain in numpy
        from oracle import image_ref as IR
        t0 = time.perf_counter(); k = 4
        for i in range(k):
            canvas = np.full((2 * S, 2 * S, 3), 114, np.uint8)
            ims = [t.cpu().numpy() for t in tiles]
            canvas[:S, :S], canvas[:S, S:], canvas[S:, :S], canvas[S:, S:] = ims
            w = IR.cv2_warp_affine_u8(canvas, M, (S, S))
            h = IR.random_hsv(w, np.array([1.01, 1.2, 0.9]))
            out = np.ascontiguousarray(h[:, ::-1].transpose(2, 0, 1)[::-1]).astype(np.float32) / 255
        dt = (time.perf_counter() - t0) / k
        print(f"cpu oracle       same chain in numpy (1 core): {1 / dt:8.1f} img/s ({dt * 1e3:.0f} ms/img)")


if __name__ == "__main__":
    main()
