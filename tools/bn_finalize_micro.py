"""Time sy11_bn_finalize in a replayed graph (what a ~5 us kernel costs there).   python tools/bn_finalize_micro.py"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
sys.path.insert(0, str(ROOT / "tools"))
import torch
from sy11 import ops
from bn_sweep import timed

for C, slots in ((64, 32), (128, 32), (256, 32), (512, 32), (64, 1)):
    ssum, ssq = torch.rand(slots, C, device="cuda"), torch.rand(slots, C, device="cuda") + 1
    gamma, beta = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    v = torch.empty(4, C, device="cuda")
    ms = timed(lambda: ops.bn_finalize(409600, ssum, ssq, gamma, beta, 1e-3, 0.03, rm, rv, v[0], v[1], v[2], v[3]), 20, False)
    print(f"C {C:4d} slots {slots:2d}: {ms * 1e3:5.2f} us per launch")
