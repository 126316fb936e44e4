"""What do the vendor libraries (MIOpen via torch.nn.functional.conv2d, channels_last f16, benchmark mode) need for the conv
layers of yolo11s (640x640, batch 64)?  Forward and backward (input + filter gradients) per layer, multiplicity-weighted.
Probe only — context for DESIGN.md; not part of the product path."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))
import torch
import torch.nn.functional as F
from conv_sweep import LAYERS

torch.backends.cudnn.benchmark = True
tot_f = tot_b = 0.0
for (H, W, C, N, k, s, g, cnt) in LAYERS:
    x = torch.randn(64, C, H, W, device="cuda", dtype=torch.float16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(N, C // g, k, k, device="cuda", dtype=torch.float16) * 0.05).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = F.conv2d(x, w, None, s, k // 2, 1, g)
    gy = torch.randn_like(y)

    def fwd():
        return F.conv2d(x, w, None, s, k // 2, 1, g)

    def bwd():
        return torch.autograd.grad(y, (x, w), gy, retain_graph=True)

    res = []
    for f in (fwd, bwd):
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(5):
            f()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 5)
    tot_f += res[0] * cnt
    tot_b += res[1] * cnt
    print(f"{H:3d}x{W:<3d} {C:4d}->{N:<4d} k{k} s{s} g{g:<3d} x{cnt}  fwd {res[0] * 1e3:8.1f} us  bwd(dgrad+wgrad) {res[1] * 1e3:8.1f} us", flush=True)
print(f"TOTAL ms/step: fwd {tot_f:.3f}  bwd {tot_b:.3f}")
