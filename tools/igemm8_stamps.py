"""Where a phase of the 8-wave pipeline (igemm cfg 20) spends its cycles: SY11_IGEMM_DEBUG=9 build stamps (s_memtime) summed over
all phases of workgroup 0, waves 0 (group 0) and 4 (group 1).   SY11_IGEMM_DEBUG=9 python tools/igemm8_stamps.py"""
import ctypes as C, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
os.environ["SY11_IGEMM_DEBUG"] = "9"
import torch
from sy11 import ops, _lib
B, dt = 64, torch.float16
SHAPES = [(80, 80, 256, 256, 3, 2), (160, 160, 128, 128, 3, 2), (40, 40, 768, 256, 1, 1), (80, 80, 512, 128, 1, 1)]
_lib.set_option("tune", 0)
_lib.set_option("igemm_cfg", int(sys.argv[1]) if len(sys.argv) > 1 else 20)
names = ["read issue", "piece issue", "waits", "barrier 1", "mfma issue", "barrier 2"]
for (H, W, Cc, N, k, s) in SHAPES:
    p = k // 2
    OH, OW = ops.conv_out_hw(H, W, k, s, p)
    x = torch.randn(B, H, W, Cc, device="cuda", dtype=dt)
    w = (torch.randn(N, k, k, Cc, device="cuda") / (Cc * k * k) ** 0.5).to(dt)
    y = torch.empty(B, OH, OW, N, device="cuda", dtype=dt)
    st = torch.zeros(2, 32, N, device="cuda")
    for _ in range(3):
        ops.conv2d_fwd(x, w, y, k, s, p, stats=(st[0], st[1]))
    torch.cuda.synchronize()
    buf = (C.c_uint64 * 16)()
    _lib.check(_lib.load().sy11_debug_stamps(buf), "sy11_debug_stamps")
    v = list(buf)
    print(f"{H}x{W} {Cc}->{N} k{k}s{s}")
    for g in range(2):
        r = v[g * 8:(g + 1) * 8]
        n = max(r[6], 1)
        print(f"  group {g}: {r[6]} phases, kernel {r[7]} cycles, loop {sum(r[:6])} = {sum(r[:6]) / n:.0f} per phase: " +
              ", ".join(f"{nm} {r[i] / n:.0f}" for i, nm in enumerate(names)))
