"""Time the 3x3 stride-2 input gradient on the five down-sampling layers of yolo11s (640x640, batch 64, f16): the four igemm launches
(one per output-pixel parity) against the fused-parity halo_dgrad_s2_kernel, plain and accumulating.   python tools/dgrad_s2_micro.py"""
import sys, math
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
import torch
from sy11 import _lib, ops as o
DEV = "cuda"
cases = [(64, 32, 320, 320, 64), (64, 128, 160, 160, 128), (64, 256, 80, 80, 256), (64, 256, 40, 40, 512), (64, 128, 80, 80, 128), (64, 256, 40, 40, 256)]
for B, Cn, H, W, N in cases:
    OH, OW = o.conv_out_hw(H, W, 3, 2, 1)
    dy = torch.randn(B, OH, OW, N, device=DEV, dtype=torch.float16)
    wk = (torch.randn(N, 3, 3, Cn, device=DEV) / math.sqrt(Cn * 9)).half()
    wt = o.weight_transpose(wk)
    dx = torch.zeros(B, H, W, Cn, dtype=torch.float16, device=DEV)
    for acc in (False, True):
        res = []
        for flag in (0, 2):
            _lib.set_option("dgrad_s2_halo", flag)
            dx.zero_()
            for _ in range(3):
                o.conv2d_dgrad(dy, wt, dx, (B, OH, OW, N), 3, 2, 1, accumulate=acc)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                o.conv2d_dgrad(dy, wt, dx, (B, OH, OW, N), 3, 2, 1, accumulate=acc)
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 20 * 1000)
        print(f"dx {Cn}ch {H}x{W} <- dy {N}ch accumulate={acc}: igemm x4 {res[0]:.0f} us, fused {res[1]:.0f} us", flush=True)
_lib.set_option("dgrad_s2_halo", 1)
