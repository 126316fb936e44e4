"""Time the SPPF max-pool kernels at the bench shape (64 x 20 x 20 x 256, f16)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
import torch
from sy11 import ops
x = torch.randn(64, 20, 20, 256, device="cuda", dtype=torch.float16)
y = torch.empty_like(x); dx = torch.empty_like(x)
idx = torch.empty(64, 20, 20, 256, dtype=torch.uint8, device="cuda")
for name, f in (("fwd", lambda: ops.maxpool5_fwd(x, y, idx)), ("bwd", lambda: ops.maxpool5_bwd(y, idx, dx))):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50):
        f()
    e1.record(); torch.cuda.synchronize()
    print(f"maxpool5 {name}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
