"""Time the GEMM filter-gradient configurations (0-11) against the wide tile (16-19) on the yolo11s layers that take it.
python tools/wgrad_wide_probe.py"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
sys.path.insert(0, str(ROOT / "tools"))
import torch
from sy11 import ops, _lib
from bn_sweep import timed

B, dt = 64, torch.float16
# (H, W of the input, C, N, k, s, count per step)
SHAPES = [(160, 160, 64, 128, 3, 2, 1), (80, 80, 128, 256, 3, 2, 1), (80, 80, 256, 256, 3, 2, 1), (40, 40, 256, 512, 3, 2, 1), (40, 40, 128, 128, 3, 2, 1),
          (80, 80, 128, 256, 1, 1, 1), (80, 80, 256, 128, 1, 1, 2), (80, 80, 384, 128, 1, 1, 1), (80, 80, 512, 128, 1, 1, 1), (40, 40, 256, 256, 1, 1, 3), (40, 40, 384, 256, 1, 1, 4), (40, 40, 768, 256, 1, 1, 1),
          (20, 20, 512, 512, 1, 1, 2), (20, 20, 512, 256, 1, 1, 2), (20, 20, 1024, 512, 1, 1, 1), (20, 20, 768, 512, 1, 1, 1), (20, 20, 256, 256, 3, 1, 2), (40, 40, 128, 128, 3, 1, 2)]


def main():
    _lib.set_option("tune", 0)
    tot_old = tot_new = 0.0
    for (H, W, C, N, k, s, cnt) in SHAPES:
        OH, OW = ops.conv_out_hw(H, W, k, s, k // 2)
        x = torch.randn(B, H, W, C, device="cuda", dtype=dt)
        dy = torch.randn(B, OH, OW, N, device="cuda", dtype=dt)
        dw = torch.zeros(N, k, k, C, device="cuda")
        gf = 2.0 * B * OH * OW * N * C * k * k / 1e9
        res = {}
        for cfg in list(range(12)) + list(range(16, 20)):
            _lib.set_option("wgrad_cfg", cfg)
            res[cfg] = timed(lambda: ops.conv2d_wgrad(x, dy, dw, k, s, k // 2), 10, False)
        _lib.set_option("wgrad_cfg", -1)
        bo = min(range(12), key=lambda c: res[c])
        bn = min(range(16, 20), key=lambda c: res[c])
        tot_old += res[bo] * cnt
        tot_new += min(res[bo], res[bn]) * cnt
        print(f"{H}x{W} {C}->{N} k{k} s{s} x{cnt}: best GEMM cfg {bo} {res[bo] * 1e3:6.1f} us {gf / res[bo]:5.0f} TF   wide cfg {bn} {res[bn] * 1e3:6.1f} us {gf / res[bn]:5.0f} TF   "
              + " ".join(f"{res[c] * 1e3:.0f}" for c in range(16, 20)), flush=True)
    print(f"TOTAL per step: best GEMM {tot_old:.3f} ms, with the wide tile {tot_new:.3f} ms")


if __name__ == "__main__":
    main()
