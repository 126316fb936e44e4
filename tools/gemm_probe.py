"""How fast does the vendor GEMM (torch.mm -> hipBLASLt / rocBLAS) run the 1x1-conv backward GEMMs of yolo11s?
Probe only (decides whether a library candidate in the tuner would be worth adding); not part of the product path."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
sys.path.insert(0, str(ROOT / "tools"))
import torch
from sy11 import ops
from conv_sweep import LAYERS

tot = {"wgrad_mine": 0.0, "wgrad_lib": 0.0, "dgrad_mine": 0.0, "dgrad_lib": 0.0}
for (H, W, C, N, k, s, g, cnt) in LAYERS:
    if k != 1 or g != 1:
        continue
    M = 64 * H * W
    x = torch.randn(M, C, device="cuda", dtype=torch.float16)
    dy = torch.randn(M, N, device="cuda", dtype=torch.float16)
    w = torch.randn(N, C, device="cuda", dtype=torch.float16)
    dw = torch.zeros(N, 1, 1, C, device="cuda")
    dx = torch.empty(64, H, W, C, device="cuda", dtype=torch.float16)
    wt = ops.weight_transpose(w.view(N, 1, 1, C))
    fns = {"wgrad_mine": lambda: ops.conv2d_wgrad(x.view(64, H, W, C), dy.view(64, H, W, N), dw, 1, 1, 0),
           "wgrad_lib": lambda: torch.mm(dy.t(), x),
           "dgrad_mine": lambda: ops.conv2d_dgrad(dy.view(64, H, W, N), wt, dx, (64, H, W, N), 1, 1, 0),
           "dgrad_lib": lambda: torch.mm(dy, w)}
    row = f"{H:3d}x{W:<3d} {C:4d}->{N:<4d} x{cnt}"
    for name, f in fns.items():
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10):
            f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        tot[name] += ms * cnt
        row += f"  {name} {ms * 1e3:6.1f}"
    print(row)
print("TOTAL ms/step:", {k: round(v, 3) for k, v in tot.items()})
