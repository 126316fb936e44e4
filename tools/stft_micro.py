"""Time the STFT/log-mel producer at the bench shape (64 x 164608 complex samples -> 640 frames x 640 mel)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
import torch
from sy11.data.spectrogram import SpectrogramProducer

dev = torch.device("cuda", 0)
p = SpectrogramProducer(dev, n_frames=640, n_mel=640)
iq = torch.randn(64, p.n_samples, dtype=torch.complex64, device=dev)
for _ in range(3):
    img = p(iq)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(20):
    img = p(iq)
e1.record()
torch.cuda.synchronize()
print(f"producer: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per 64-image batch")
db, mm = p.logmel_db(iq)
torch.cuda.synchronize()
for what, fn in (("stft_logmel alone", lambda: p.logmel_db(iq)),):
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    by = 64 * p.n_samples * 8 + 64 * 640 * 640 * 4                      # every IQ sample once + every dB value once
    print(f"{what}: {us:.1f} us = {by / us / 1e3:.0f} GB/s algorithmic ({by / 1e6:.0f} MB)")
