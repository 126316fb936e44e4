"""Criterion kernels at the benchmarked size (64 images, 8400 anchors, 80 classes): time per launch of the terms pass and of the
gradient pass.    python tools/loss_micro.py [reps=50]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
import torch
from sy11 import ops as K

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B, nc = 64, 80
g = torch.Generator().manual_seed(0)
maps = [torch.randn(B, s, s, 64 + nc, generator=g).cuda() for s in (80, 40, 20)]
n = 8
gt = torch.zeros(B, n, 5)
gt[..., 0] = torch.randint(0, nc, (B, n), generator=g).float()
xy = 80 + 480 * torch.rand(B, n, 2, generator=g)
wh = 40 + 200 * torch.rand(B, n, 2, generator=g)
gt[..., 1:3], gt[..., 3:5] = xy - wh / 2, xy + wh / 2
w = K.det_loss_assign(maps, (8.0, 16.0, 32.0), nc, gt.cuda())
up = torch.ones(1, device="cuda")


def timed(fn):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f"terms {timed(lambda: K.det_loss_terms(w)):.1f} us   gradient {timed(lambda: K.det_loss_backward(w, up, (7.5, 0.5, 1.5))):.1f} us   assign stage {timed(lambda: K.det_loss_assign(maps, (8.0, 16.0, 32.0), nc, gt.cuda())):.1f} us")
