"""Per-C-ABI-entry time of one fused eval forward (predictor path) of yolo11s at batch 64, 640 x 640, f16."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
import torch
from sy11 import _lib
from sy11.nn.tasks import DetectionModel

m = DetectionModel("yolo11s.yaml", nc=80, verbose=False).cuda().eval()
m.fuse()
m._sy11_dtype = torch.float16
img = torch.rand(64, 3, 640, 640, device="cuda")
with torch.no_grad():
    for _ in range(3):
        m(img)
    torch.cuda.synchronize()
    _lib.PROFILE = []
    m(img)
    torch.cuda.synchronize()
prof, _lib.PROFILE = _lib.PROFILE, None
fam = {}
for name, e0, e1, meta in prof:
    f = fam.setdefault(name, [0.0, 0])
    f[0] += e0.elapsed_time(e1); f[1] += 1
tot = sum(v[0] for v in fam.values())
for k, (ms, n) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
    print(f"{ms:7.3f} ms  {n:4d} launches  {k}")
print(f"{tot:7.3f} ms total in C-ABI launches")
