"""Wall/GPU time of the phases of one training step (events on the current stream)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
import torch
import bench
from sy11.data.spectrogram import SpectrogramProducer
from sy11.engine.trainer import DetectionTrainer
from sy11.nn.tasks import DetectionModel

dev = torch.device("cuda:0")
model = DetectionModel("yolo11s.yaml", nc=80, verbose=False)
prod = SpectrogramProducer(dev)
tr = DetectionTrainer(model, batch_size=64, device=dev, overrides={"amp": True}, producer=prod, graphs="--no-graphs" not in sys.argv)
labels = bench.synthetic_labels(64, 100, dev)
iq = bench.synthetic_iq(64, prod.n_samples, 1, dev)
for _ in range(4):
    tr.train_step({"iq": iq, **labels})
torch.cuda.synchronize()
names = ["stft", "forward", "loss", "backward", "optimizer"]
acc = {n: [0.0, 0.0] for n in names}
N = 5
for _ in range(N):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    ts = []
    tr.model.train()
    ev[0].record(); ts.append(time.perf_counter())
    b = tr.preprocess_batch({"iq": iq, **labels})
    ev[1].record(); ts.append(time.perf_counter())
    maps = tr.model(b["img"])
    ev[2].record(); ts.append(time.perf_counter())
    loss, items = tr.model.loss(b, maps)
    ev[3].record(); ts.append(time.perf_counter())
    tr.scaler.scale(loss).backward()
    ev[4].record(); ts.append(time.perf_counter())
    tr.optimizer_step()
    ev[5].record(); ts.append(time.perf_counter())
    torch.cuda.synchronize()
    for i, n in enumerate(names):
        acc[n][0] += ev[i].elapsed_time(ev[i + 1])
        acc[n][1] += (ts[i + 1] - ts[i]) * 1e3
for n in names:
    print(f"{n:10s} gpu {acc[n][0] / N:7.2f} ms   host {acc[n][1] / N:7.2f} ms")
