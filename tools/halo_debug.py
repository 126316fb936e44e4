"""Compare the halo-tiled 3x3 kernel (igemm_cfg 15 / 16) with the generic igemm (cfg 0) on one problem; print where they differ."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
import torch
from sy11 import _lib, ops
B, C, H, W, N, s = [int(v) for v in sys.argv[1:7]]
mode = sys.argv[7] if len(sys.argv) > 7 else "fwd"
cfg = int(sys.argv[8]) if len(sys.argv) > 8 else 15
torch.manual_seed(0)
k, p = 3, 1
OH, OW = ops.conv_out_hw(H, W, k, s, p)
x = torch.randn(B, H, W, C, device="cuda").half()
w = (torch.randn(N, k, k, C, device="cuda") / (C * 9) ** 0.5).half()
dy = torch.randn(B, OH, OW, N, device="cuda").half()
wt = ops.weight_transpose(w)
_lib.set_option("tune", 0)
outs = {}
for c in (0, cfg):
    if mode == "dgrad2":                    # fused-parity stride-2 input gradient (conv3x3.hip) vs the four igemm launches
        _lib.set_option("igemm_cfg", -1)
        _lib.set_option("dgrad_s2_halo", 0 if c == 0 else 2)
    else:
        _lib.set_option("igemm_cfg", c)
    if mode == "fwd":
        y = torch.zeros(B, OH, OW, N, device="cuda", dtype=torch.float16)
        ops.conv2d_fwd(x, w, y, k, s, p)
    else:
        y = torch.zeros(B, H, W, C, device="cuda", dtype=torch.float16)
        ops.conv2d_dgrad(dy, wt, y, (B, OH, OW, N), k, s, p)
    torch.cuda.synchronize()
    outs[c] = y.float().cpu()
d = (outs[0] - outs[cfg]).abs()
print(f"{mode} B{B} {H}x{W}x{C}->{N} s{s} cfg {cfg}: max diff {d.max():.4f} (scale {outs[0].abs().max():.3f}); equal to igemm: {bool((d < 2e-2).all())}")
if d.max() > 2e-2:
    bad = (d > 2e-2)
    print(" bad fraction", bad.float().mean().item())
    print(" by image", bad.float().mean((1, 2, 3)).tolist())
    print(" by row  ", [round(v, 2) for v in bad.float().mean((0, 2, 3)).tolist()])
    print(" by col  ", [round(v, 2) for v in bad.float().mean((0, 1, 3)).tolist()])
    ch = bad.float().mean((0, 1, 2))
    print(" by channel (groups of 8)", [round(ch[i:i + 8].mean().item(), 2) for i in range(0, ch.numel(), 8)])
