"""Time chosen igemm tile configurations (forward with BN statistics, replayed graph) on the large layers of yolo11s; the
SY11_IGEMM_DEBUG ablations apply (1: no LDS-DMA issue, 2: DMA only, 3: no epilogue, 4: epilogue only).
python tools/igemm_probe.py [cfg ...] [--eager]      (--eager: plain launches, for per-dispatch PMC counters under rocprofv3)"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
sys.path.insert(0, str(ROOT / "tools"))
import torch
from sy11 import ops, _lib
from bn_sweep import timed

B, dt = 64, torch.float16
SHAPES = [(80, 80, 256, 256, 3, 2), (160, 160, 128, 128, 3, 2), (40, 40, 768, 256, 1, 1), (40, 40, 384, 256, 1, 1), (20, 20, 1024, 512, 1, 1),
          (80, 80, 512, 128, 1, 1), (80, 80, 192, 256, 1, 1), (40, 40, 256, 512, 3, 2)]


def main():
    eager = "--eager" in sys.argv                     # plain launches (per-dispatch PMC counters under rocprofv3)
    cfgs = [int(v) for v in sys.argv[1:] if not v.startswith("-")] or [0, 3, 12]
    _lib.set_option("tune", 0)
    for (H, W, C, N, k, s) in SHAPES:
        p = k // 2
        OH, OW = ops.conv_out_hw(H, W, k, s, p)
        x = torch.randn(B, H, W, C, device="cuda", dtype=dt)
        w = (torch.randn(N, k, k, C, device="cuda") / (C * k * k) ** 0.5).to(dt)
        y = torch.empty(B, OH, OW, N, device="cuda", dtype=dt)
        st = torch.zeros(2, 32, N, device="cuda")
        row = f"{H}x{W} {C}->{N} k{k}s{s}"
        for cfg in cfgs:
            _lib.set_option("igemm_cfg", cfg)
            ms = timed(lambda: ops.conv2d_fwd(x, w, y, k, s, p, stats=(st[0], st[1])), 3 if eager else 10, eager)
            gf = 2.0 * B * OH * OW * N * C * k * k / 1e9
            row += f"  cfg{cfg} {ms * 1e3:6.1f}us {gf / ms:5.0f}TF"
        print(row, flush=True)


if __name__ == "__main__":
    main()
