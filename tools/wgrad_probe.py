"""Time every filter-gradient configuration on the 3x3 stride-1 layers of yolo11s (replayed graph).   python tools/wgrad_probe.py"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
sys.path.insert(0, str(ROOT / "tools"))
import torch
from sy11 import ops, _lib
from bn_sweep import timed

B, dt = 64, torch.float16
SHAPES = [(160, 160, 16, 32, 1), (160, 160, 32, 16, 1), (80, 80, 32, 64, 2), (80, 80, 64, 32, 2), (80, 80, 64, 64, 1), (80, 80, 128, 64, 1), (40, 40, 64, 64, 5), (40, 40, 64, 128, 2), (40, 40, 128, 64, 2), (40, 40, 256, 64, 1),
          (20, 20, 64, 64, 1), (20, 20, 128, 128, 8), (20, 20, 512, 64, 1)]


def main():
    _lib.set_option("tune", 0)
    tot_old = tot_new = 0.0
    for (H, W, C, N, cnt) in SHAPES:
        x = torch.randn(B, H, W, C, device="cuda", dtype=dt)
        dy = torch.randn(B, H, W, N, device="cuda", dtype=dt)
        dw = torch.zeros(N, 3, 3, C, device="cuda")
        gf = 2.0 * B * H * W * N * C * 9 / 1e9
        res = {}
        for cfg in range(20):
            _lib.set_option("wgrad_cfg", cfg)
            res[cfg] = timed(lambda: ops.conv2d_wgrad(x, dy, dw, 3, 1, 1), 10, False)
        _lib.set_option("wgrad_cfg", -1)
        bo = min([c for c in range(12)], key=lambda c: res[c])
        bn = min(range(12, 16), key=lambda c: res[c])
        tot_old += res[bo] * cnt
        tot_new += min(res[bo], res[bn]) * cnt
        print(f"{H}x{W} {C}->{N} x{cnt}: best GEMM cfg {bo} {res[bo] * 1e3:6.1f} us {gf / res[bo]:5.0f} TF   patch cfg {bn} {res[bn] * 1e3:6.1f} us {gf / res[bn]:5.0f} TF   "
              + " ".join(f"{res[c] * 1e3:.0f}" for c in range(12, 16)), flush=True)
    print(f"TOTAL per step: best GEMM {tot_old:.3f} ms, with the patch kernel {tot_new:.3f} ms")


if __name__ == "__main__":
    main()
