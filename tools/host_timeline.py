"""Host-side cost of one training step (is the launch queue kept full?): per-phase host time WITHOUT synchronising,
then the synchronised step time.  python tools/host_timeline.py"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
sys.path.insert(0, str(ROOT))
import torch
import bench
from sy11.data.spectrogram import SpectrogramProducer
from sy11.engine.trainer import DetectionTrainer
from sy11.nn.tasks import DetectionModel


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    producer = SpectrogramProducer(dev, n_frames=640, n_mel=640)
    t = DetectionTrainer(DetectionModel("yolo11s.yaml", nc=80, verbose=False), batch_size=64, device=dev, overrides={"amp": True},
                         producer=producer, graphs=True)
    batch = {"iq": bench.synthetic_iq(64, producer.n_samples, 1, dev), **bench.synthetic_labels(64, 100, dev)}
    for _ in range(6):
        t.train_step(dict(batch))
    torch.cuda.synchronize()
    N = 20
    marks = []
    t0 = time.perf_counter()
    for _ in range(N):
        h0 = time.perf_counter()
        b = t.preprocess_batch(dict(batch))
        h1 = time.perf_counter()
        loss, items = t.model(b)
        h2 = time.perf_counter()
        t.scaler.scale(loss).backward()
        h3 = time.perf_counter()
        t.optimizer_step()
        h4 = time.perf_counter()
        marks.append((h1 - h0, h2 - h1, h3 - h2, h4 - h3))
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    tg = time.perf_counter() - t0
    m = [sum(x[i] for x in marks[5:]) / (N - 5) * 1e3 for i in range(4)]
    print(f"host ms/step: preprocess {m[0]:.2f}  forward+loss {m[1]:.2f}  backward {m[2]:.2f}  optimizer {m[3]:.2f}  | host loop {th / N * 1e3:.2f} ms/step, synced {tg / N * 1e3:.2f} ms/step")
    print("per-step host (ms):", [" ".join(f"{v * 1e3:.1f}" for v in x) for x in marks[-4:]])


if __name__ == "__main__":
    main()
