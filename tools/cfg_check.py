"""Correctness of a forced igemm tile configuration (SY11_IGEMM_CFG=0..8) on layer-sized problems: conv fwd (+BN
statistics) and dgrad against torch's own convolution.   SY11_IGEMM_CFG=3 python tools/cfg_check.py"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
import torch
import torch.nn.functional as F
from sy11 import ops

torch.manual_seed(0)
worst = 0.0
for (B, H, W, C, N, k, s) in [(16, 160, 160, 32, 128, 3, 1), (64, 80, 80, 64, 128, 1, 1), (64, 40, 40, 128, 256, 3, 1), (64, 81, 79, 64, 128, 3, 2), (10, 100, 100, 64, 256, 3, 1)]:
    p = k // 2
    OH, OW = ops.conv_out_hw(H, W, k, s, p)
    x = torch.randn(B, H, W, C, device="cuda", dtype=torch.float16)
    w = (torch.randn(N, k, k, C, device="cuda") / (C * k * k) ** 0.5).half()
    y = torch.empty(B, OH, OW, N, device="cuda", dtype=torch.float16)
    st = torch.zeros(2, 32, N, device="cuda")
    ops.conv2d_fwd(x, w, y, k, s, p, stats=(st[0], st[1]))
    ref = F.conv2d(x.permute(0, 3, 1, 2).float(), w.permute(0, 3, 1, 2).float(), stride=s, padding=p).permute(0, 2, 3, 1)
    e1 = (y.float() - ref).abs().max().item() / ref.abs().max().item()
    e2 = (st[0].sum(0) - ref.sum((0, 1, 2))).abs().max().item() / ref.abs().sum((0, 1, 2)).max().item()
    e3 = (st[1].sum(0) - (ref * ref).sum((0, 1, 2))).abs().max().item() / (ref * ref).sum((0, 1, 2)).max().item()
    dy = torch.randn(B, OH, OW, N, device="cuda", dtype=torch.float16)
    dx = torch.empty_like(x)
    ops.conv2d_dgrad(dy, ops.weight_transpose(w), dx, (B, OH, OW, N), k, s, p)
    xr = x.permute(0, 3, 1, 2).float().requires_grad_(True)
    F.conv2d(xr, w.permute(0, 3, 1, 2).float(), stride=s, padding=p).backward(dy.permute(0, 3, 1, 2).float())
    gref = xr.grad.permute(0, 2, 3, 1)
    e4 = (dx.float() - gref).abs().max().item() / gref.abs().max().item()
    print(f"cfg={os.environ.get('SY11_IGEMM_CFG', 'tuned')} B{B} {H}x{W} {C}->{N} k{k}s{s}: fwd {e1:.2e} sum {e2:.2e} sumsq {e3:.2e} dgrad {e4:.2e}")
    worst = max(worst, e1, e2, e3, e4)
assert worst < 4e-3, worst
print("ok", worst)
