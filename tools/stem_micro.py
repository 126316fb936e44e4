"""Time the stem kernels (3 -> 32, 3x3 s2) at the bench shape."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
import torch
from sy11 import ops
B, H = 64, 640
x = torch.rand(B, 3, H, H, device="cuda")
w = (torch.randn(32, 3, 3, 3, device="cuda") * 0.2).half()
y = torch.empty(B, 320, 320, 32, device="cuda", dtype=torch.float16)
dy = torch.randn_like(y)
dw = torch.zeros(32, 3, 3, 3, device="cuda")
st = torch.zeros(2, 32, 32, device="cuda")
for name, f in (("fwd", lambda: ops.stem_conv_fwd(x, w, y, 2, 1, stats=(st[0], st[1]))), ("wgrad", lambda: ops.stem_conv_wgrad(x, dy, dw, 2, 1))):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20):
        f()
    e1.record(); torch.cuda.synchronize()
    print(f"stem {name}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
