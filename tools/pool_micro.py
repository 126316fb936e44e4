"""Time the SPPF max-pools (5 x 5, stride 1) of yolo11s at batch 64: 20 x 20 x 256, forward (with argmax) and backward,
inside a replayed graph.   python tools/pool_micro.py      (SY11_MAXPOOL_SEP=0: the 25-way forward)"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
sys.path.insert(0, str(ROOT / "tools"))
import torch
from sy11 import ops
from bn_sweep import timed

x = torch.randn(64, 20, 20, 256, device="cuda", dtype=torch.float16)
y = torch.empty_like(x)
idx = torch.empty(64, 20, 20, 256, dtype=torch.uint8, device="cuda")
dy = torch.randn_like(x)
dx = torch.empty_like(x)
for name, fn in (("fwd", lambda: ops.maxpool5_fwd(x, y, idx)), ("bwd", lambda: ops.maxpool5_bwd(dy, idx, dx))):
    ms = timed(fn, 10, False)
    print(f"maxpool5 {name} 64x20x20x256  {ms * 1e3:6.1f} us")
