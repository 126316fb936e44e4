"""What does plain streaming reach on this GPU at BatchNorm tensor sizes?  torch copy / add / sum vs the BN passes."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
import torch
from sy11 import ops


def t(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for (B, H, W, C) in [(64, 160, 160, 64), (64, 160, 160, 128), (64, 80, 80, 128), (64, 80, 80, 256), (64, 40, 40, 256), (64, 20, 20, 512)]:
    y = torch.randn(B, H, W, C, device="cuda", dtype=torch.float16)
    z = torch.empty_like(y)
    dz = torch.randn_like(y)
    nb = y.numel() * 2
    f = lambda *s: torch.rand(*s, device="cuda") + 0.5
    mean, rstd, scale, shift, gamma = f(C), f(C), f(C), f(C), f(C)
    sg, sgx = torch.zeros(32, C, device="cuda"), torch.zeros(32, C, device="cuda")
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    r = {}
    r["copy"] = 2 * nb / t(lambda: z.copy_(y))
    r["sum"] = nb / t(lambda: torch.sum(y, dtype=torch.float32))
    r["colsum"] = nb / t(lambda: y.view(-1, C).sum(0, dtype=torch.float32))
    r["mul2"] = 3 * nb / t(lambda: torch.mul(y, dz, out=z))
    r["silu"] = 2 * nb / t(lambda: torch.ops.aten.silu.out(y, out=z))
    r["silu_bwd"] = 3 * nb / t(lambda: torch.ops.aten.silu_backward.grad_input(dz, y, grad_input=z))
    r["bn_fwd"] = 2 * nb / t(lambda: ops.bn_act_fwd(y, scale, shift, z, silu=True))
    r["bn_red"] = 2 * nb / t(lambda: ops.bn_act_bwd_reduce(y, dz, mean, rstd, scale, shift, True, sg, sgx))
    r["bn_app"] = 3 * nb / t(lambda: ops.bn_act_bwd_apply(y, dz, mean, rstd, scale, shift, gamma, True, sg, sgx, z, dg, db))
    print(f"{B}x{H}x{W}x{C} ({nb / 1e6:.0f} MB): " + "  ".join(f"{k} {v / 1e9:.2f}" for k, v in r.items()) + "  TB/s")
