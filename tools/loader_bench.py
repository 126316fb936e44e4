"""Loader throughput with and without worker processes, alone and feeding a yolo11s f16 training step (files on tmpfs).
   python tools/loader_bench.py [procs ...]     (default: 0 4 8)"""
import sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
import numpy as np
import torch


def main():
    from sy11.data.dataset import YOLODataset, build_dataloader
    from sy11.engine.trainer import DetectionTrainer
    from sy11.nn.tasks import DetectionModel
    S, B, n = 640, 64, 1024
    g = np.random.default_rng(0)
    root = Path(tempfile.mkdtemp()) / "d"
    (root / "images").mkdir(parents=True); (root / "labels").mkdir()
    for i in range(n):
        np.save(root / "images" / f"s{i:04d}.npy", g.integers(0, 256, (S, S, 3), dtype=np.uint8))
        rows = np.concatenate((g.integers(0, 2, (4, 1)), g.uniform(0.3, 0.7, (4, 2)), g.uniform(0.05, 0.3, (4, 2))), 1)
        (root / "labels" / f"s{i:04d}.txt").write_text("\n".join(" ".join(f"{v:.6f}" for v in r) for r in rows))
    procs = [int(a) for a in sys.argv[1:] if not a.startswith('-')] or [0, 4, 8]
    model = DetectionModel("yolo11s.yaml", nc=2, verbose=False)
    tr = DetectionTrainer(model, batch_size=B, device="cuda", overrides={"amp": True}, graphs=True)
    for p in procs:
        ds = YOLODataset(str(root / "images"), imgsz=S, augment=True, batch_size=B, data={"nc": 2})
        static = torch.empty((B, 3, S, S), device="cuda")
        dl = build_dataloader(ds, B, workers=8, out=static, dtype=torch.float32, procs=p)
        it = iter(dl)
        for _ in range(3):
            next(dl._it)
        torch.cuda.synchronize(); t0 = time.perf_counter(); nb = 0
        import cProfile, pstats, io
        pr = cProfile.Profile()
        if "--profile" in sys.argv: pr.enable()
        for _ in range(2):
            for batch in dl:
                nb += 1
        if "--profile" in sys.argv:
            pr.disable()
            buf = io.StringIO(); pstats.Stats(pr, stream=buf).sort_stats("tottime").print_stats(18)
            print("\n".join(l[:160] for l in buf.getvalue().splitlines() if l.strip())[:4000])
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        alone = nb * B / dt
        dl.out = tr.batch_buffer(S)
        for _ in range(5):
            tr.train_step(next(dl._it))
        torch.cuda.synchronize(); t0 = time.perf_counter(); nb = 0
        for _ in range(3):
            for batch in dl:
                tr.train_step(batch); nb += 1
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"procs {p}: loader alone {alone:7.0f} img/s; training from files {nb * B / dt:7.0f} img/s ({dt / nb * 1e3:.1f} ms/step)", flush=True)
        if hasattr(dl, "close"):
            dl.close()


if __name__ == "__main__":
    main()
