"""Time the depthwise 3x3 kernels (forward with BN statistics, input gradient, filter gradient) on the Detect / C2PSA shapes of
yolo11s at 640x640, batch 64, f16, inside a replayed hipGraph of 10 launches.   python tools/dw_micro.py"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
sys.path.insert(0, str(ROOT / "tools"))
import torch
from sy11 import ops
from bn_sweep import timed

B = 64
for (H, C, cnt) in ((80, 128, 2), (40, 256, 2), (20, 512, 2), (20, 256, 1)):
    x = torch.randn(B, H, H, C, device="cuda", dtype=torch.float16)
    dy = torch.randn_like(x)
    y = torch.empty_like(x)
    dx = torch.empty_like(x)
    w = (torch.randn(C, 3, 3, 1, device="cuda") * 0.3).half()
    st = torch.zeros(2, 32, C, device="cuda")
    dw = torch.zeros(C, 3, 3, 1, device="cuda")
    nb = x.numel() * 2
    f = timed(lambda: ops.conv2d_fwd(x, w, y, 3, 1, 1, 1, C, stats=(st[0], st[1])), 10, False)
    g = timed(lambda: ops.conv2d_dgrad(dy, w, dx, (B, H, H, C), 3, 1, 1, 1, C), 10, False)
    h = timed(lambda: ops.conv2d_wgrad(x, dy, dw, 3, 1, 1, 1, C), 10, False)
    print(f"{H}x{H}x{C} x{cnt}: fwd {f * 1e3:6.1f} us {2 * nb / f / 1e9:.2f} TB/s | dgrad {g * 1e3:6.1f} us {2 * nb / g / 1e9:.2f} TB/s | "
          f"wgrad {h * 1e3:6.1f} us {2 * nb / h / 1e9:.2f} TB/s", flush=True)
