"""Micro-benchmark of one conv problem through the C-ABI (used for rocprofv3 --pmc runs)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
import torch
from sy11 import ops

def main():
    B, H, W, C, N, k, s = [int(v) for v in sys.argv[1:8]]
    mode = sys.argv[8] if len(sys.argv) > 8 else "fwd"
    reps = int(sys.argv[9]) if len(sys.argv) > 9 else 20
    dt = torch.float16
    p = k // 2
    OH, OW = ops.conv_out_hw(H, W, k, s, p)
    x = torch.randn(B, H, W, C, device="cuda", dtype=dt)
    w = (torch.randn(N, k, k, C, device="cuda") / (C * k * k) ** 0.5).to(dt)
    y = torch.empty(B, OH, OW, N, device="cuda", dtype=dt)
    dy = torch.randn(B, OH, OW, N, device="cuda", dtype=dt)
    dx = torch.empty_like(x)
    dw = torch.zeros(N, k, k, C, device="cuda")
    wt = ops.weight_transpose(w)
    st = torch.zeros(2, 128 if B * OH * OW >= 819200 else 32, N, device="cuda")
    def run():
        if mode == "fwd":
            ops.conv2d_fwd(x, w, y, k, s, p, stats=(st[0], st[1]))
        elif mode == "fwd_nostats":
            ops.conv2d_fwd(x, w, y, k, s, p)
        elif mode == "dgrad":
            ops.conv2d_dgrad(dy, wt, dx, (B, OH, OW, N), k, s, p)
        else:
            ops.conv2d_wgrad(x, dy, dw, k, s, p)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    gf = 2.0 * B * OH * OW * N * C * k * k / 1e9
    print(f"{mode} B{B} {H}x{W}x{C}->{OH}x{OW}x{N} k{k}s{s}: {ms:.4f} ms  {gf / ms:.1f} TFLOP/s")

if __name__ == "__main__":
    main()
