"""Time the few-channel 3x3 layers of yolo11s, forward (with BN statistics) and input gradient, per igemm configuration
(replayed graph).   python tools/smallc_probe.py"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
sys.path.insert(0, str(ROOT / "tools"))
import torch
from sy11 import ops, _lib
from bn_sweep import timed

B, dt = 64, torch.float16
SHAPES = [(160, 160, 16, 32, 1), (160, 160, 32, 16, 1), (80, 80, 32, 64, 2), (80, 80, 64, 32, 2)]


def main():
    _lib.set_option("tune", 0)
    tot = {"old": 0.0, "new": 0.0}
    for (H, W, C, N, cnt) in SHAPES:
        x = torch.randn(B, H, W, C, device="cuda", dtype=dt)
        w = (torch.randn(N, 3, 3, C, device="cuda") / (C * 9) ** 0.5).to(dt)
        y = torch.empty(B, H, W, N, device="cuda", dtype=dt)
        dy = torch.randn(B, H, W, N, device="cuda", dtype=dt)
        dx = torch.empty_like(x)
        wt = ops.weight_transpose(w)
        st = torch.zeros(2, 32, N, device="cuda")
        byt = (x.numel() + y.numel()) * 2
        row = f"{H}x{W} {C}->{N} x{cnt}"
        for name, fn in (("fwd", lambda: ops.conv2d_fwd(x, w, y, 3, 1, 1, stats=(st[0], st[1]))),
                         ("dgrad", lambda: ops.conv2d_dgrad(dy, wt, dx, (B, H, W, N), 3, 1, 1))):
            res = {}
            for cfg in (-1, 19):
                _lib.set_option("igemm_cfg", cfg)
                res[cfg] = timed(fn, 10, False)
            _lib.set_option("igemm_cfg", -1)
            tot["old"] += res[-1] * cnt
            tot["new"] += min(res[-1], res[19]) * cnt
            row += f"  {name}: heuristic {res[-1] * 1e3:6.1f} us  cfg19 {res[19] * 1e3:6.1f} us ({byt / res[19] / 1e9:4.2f} TB/s)"
        print(row, flush=True)
    print(f"TOTAL per step: heuristic {tot['old']:.3f} ms, with cfg 19 {tot['new']:.3f} ms")


if __name__ == "__main__":
    main()
