"""300 training steps on the bench workload: memory must not grow, step time must not drift, the loss must go down."""
import sys, time
from pathlib import Path; ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'spectrogram-yolov11_amd'))
import torch, bench
from sy11.data.spectrogram import SpectrogramProducer
from sy11.engine.trainer import DetectionTrainer
from sy11.nn.tasks import DetectionModel
dev = torch.device("cuda", 0)
producer = SpectrogramProducer(dev, n_frames=640, n_mel=640)
t = DetectionTrainer(DetectionModel("yolo11s.yaml", nc=80, verbose=False), batch_size=64, device=dev, overrides={"amp": True}, producer=producer, graphs=True)
batch = {"iq": bench.synthetic_iq(64, producer.n_samples, 1, dev), **bench.synthetic_labels(64, 100, dev)}
marks = {}
t0 = time.perf_counter()
for i in range(300):
    loss, _ = t.train_step(dict(batch))
    if i in (20, 150, 299):
        torch.cuda.synchronize()
        marks[i] = (torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30, float(loss), time.perf_counter() - t0)
for k, v in marks.items():
    print(f"step {k}: allocated {v[0]:.2f} GiB reserved {v[1]:.2f} GiB loss {v[2]:.1f} t {v[3]:.1f}s")
a, b = marks[150], marks[299]
print("ms/step between 150 and 299:", (b[3] - a[3]) / 149 * 1e3, " alloc growth GiB:", b[0] - a[0], " reserved growth:", b[1] - a[1], " scale", t.scaler.get_scale())
