"""Time the stem convolution (3 -> N, 3x3, stride 2, f32 NCHW image in, 16-bit NHWC out) forward and filter gradient at the
benchmarked size inside a replayed hipGraph.   python tools/stem_micro.py [N] [B] [size]     (SY11_STEM_TILE=0: the gather kernels)"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
sys.path.insert(0, str(ROOT / "tools"))
import torch
from sy11 import ops
from bn_sweep import timed


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 640
    dt = torch.float16
    x = torch.rand(B, 3, S, S, device="cuda")
    w = (torch.randn(N, 3, 3, 3, device="cuda") * 0.2).to(dt)
    OH, OW = ops.conv_out_hw(S, S, 3, 2, 1)
    y = torch.empty(B, OH, OW, N, device="cuda", dtype=dt)
    dy = torch.randn_like(y)
    st = torch.zeros(2, 32, N, device="cuda")
    dw = torch.zeros(N, 3, 3, 3, device="cuda")
    byt = x.numel() * 4 + y.numel() * 2
    for name, fn in (("fwd (+stats)", lambda: ops.stem_conv_fwd(x, w, y, 2, 1, stats=(st[0], st[1]))),
                     ("wgrad", lambda: ops.stem_conv_wgrad(x, dy, dw, 2, 1))):
        ms = timed(fn, 10, False)
        print(f"stem {B}x3x{S}x{S} -> {N}  {name:13s} {ms * 1e3:7.1f} us   {byt / ms / 1e9:5.2f} TB/s of image + map bytes")


if __name__ == "__main__":
    main()
