"""Time every distinct convolution of yolo11s (640x640, batch 64, f16) through the C-ABI, per direction, and print the
multiplicity-weighted totals.  Used to tune tile / split heuristics:  python tools/conv_sweep.py [fwd,fused,dgrad,wgrad] [-v] [--eager]
Timed inside a replayed hipGraph of 10 calls, as the trainer runs them (--eager: plain calls, host call rate included).

Layer table = (IH, IW, C, N, k, s, groups, count) of the model graph (count = how many layers share the shape)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
sys.path.insert(0, str(ROOT / "tools"))
import torch
from sy11 import ops

LAYERS = [(320, 320, 32, 64, 3, 2, 1, 1), (160, 160, 64, 64, 1, 1, 1, 1), (160, 160, 96, 128, 1, 1, 1, 1), (160, 160, 16, 32, 3, 1, 1, 1),
          (160, 160, 32, 16, 3, 1, 1, 1), (160, 160, 128, 128, 3, 2, 1, 1), (80, 80, 64, 64, 1, 1, 1, 1), (80, 80, 128, 128, 1, 1, 1, 3),
          (80, 80, 128, 80, 1, 1, 1, 1), (80, 80, 192, 256, 1, 1, 1, 1), (80, 80, 192, 128, 1, 1, 1, 1), (80, 80, 512, 128, 1, 1, 1, 1),
          (80, 80, 32, 64, 3, 1, 1, 2), (80, 80, 64, 32, 3, 1, 1, 2), (80, 80, 64, 64, 3, 1, 1, 1), (80, 80, 128, 128, 3, 2, 1, 1),
          (80, 80, 128, 64, 3, 1, 1, 1), (80, 80, 128, 128, 3, 1, 128, 2), (80, 80, 256, 256, 3, 2, 1, 1), (40, 40, 64, 64, 1, 1, 1, 1),
          (40, 40, 128, 64, 1, 1, 1, 2), (40, 40, 128, 128, 1, 1, 1, 2), (40, 40, 128, 80, 1, 1, 1, 1), (40, 40, 256, 256, 1, 1, 1, 1),
          (40, 40, 256, 128, 1, 1, 1, 1), (40, 40, 384, 256, 1, 1, 1, 4), (40, 40, 768, 256, 1, 1, 1, 1), (40, 40, 64, 64, 3, 1, 1, 5),
          (40, 40, 64, 128, 3, 1, 1, 2), (40, 40, 128, 64, 3, 1, 1, 2), (40, 40, 128, 128, 3, 1, 128, 1), (40, 40, 256, 512, 3, 2, 1, 1),
          (40, 40, 256, 256, 3, 2, 1, 1), (40, 40, 256, 64, 3, 1, 1, 1), (40, 40, 256, 256, 3, 1, 256, 1), (20, 20, 64, 64, 1, 1, 1, 1),
          (20, 20, 128, 128, 1, 1, 1, 1), (20, 20, 128, 80, 1, 1, 1, 1), (20, 20, 256, 128, 1, 1, 1, 4), (20, 20, 256, 256, 1, 1, 1, 3),
          (20, 20, 256, 512, 1, 1, 1, 2), (20, 20, 512, 512, 1, 1, 1, 3), (20, 20, 512, 256, 1, 1, 1, 2), (20, 20, 512, 128, 1, 1, 1, 1),
          (20, 20, 768, 512, 1, 1, 1, 3), (20, 20, 1024, 512, 1, 1, 1, 1), (20, 20, 64, 64, 3, 1, 1, 1), (20, 20, 128, 128, 3, 1, 1, 8),
          (20, 20, 128, 128, 3, 1, 128, 1), (20, 20, 256, 256, 3, 1, 256, 1), (20, 20, 512, 64, 3, 1, 1, 1), (20, 20, 512, 512, 3, 1, 512, 1)]


def main():
    modes = sys.argv[1].split(",") if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else ["fwd", "dgrad", "wgrad"]
    from bn_sweep import timed                     # 10 calls inside a replayed hipGraph (a small kernel's eager time is the host's call rate)
    verbose, eager = "-v" in sys.argv, "--eager" in sys.argv
    B, dt, reps = 64, torch.float16, 10
    tot = {m: 0.0 for m in modes}
    floors = {}
    lines = []
    for (H, W, C, N, k, s, g, cnt) in LAYERS:
        p = k // 2
        OH, OW = ops.conv_out_hw(H, W, k, s, p)
        x = torch.randn(B, H, W, C, device="cuda", dtype=dt)
        w = (torch.randn(N, k, k, C // g, device="cuda") / (C // g * k * k) ** 0.5).to(dt)
        y = torch.empty(B, OH, OW, N, device="cuda", dtype=dt)
        dy = torch.randn(B, OH, OW, N, device="cuda", dtype=dt)
        dx = torch.empty_like(x)
        dw = torch.zeros(N, k, k, C // g, device="cuda")
        wt = ops.weight_transpose(w) if g == 1 else w
        st = torch.zeros(2, 32, N, device="cuda")
        bias = torch.zeros(N, device="cuda")
        fns = {"fwd": lambda: ops.conv2d_fwd(x, w, y, k, s, p, groups=g, stats=(st[0], st[1])),
               "fused": lambda: ops.conv2d_fwd(x, w, y, k, s, p, groups=g, bias=bias, silu=True),      # the inference conv (BN folded)
               "dgrad": lambda: ops.conv2d_dgrad(dy, wt, dx, (B, OH, OW, N), k, s, p, groups=g),
               "wgrad": lambda: ops.conv2d_wgrad(x, dy, dw, k, s, p, groups=g)}
        row = f"{H:3d}x{W:<3d} {C:4d}->{N:<4d} k{k} s{s} g{g:<3d} x{cnt}"
        for m in modes:
            ms = timed(fns[m], reps, eager)
            tot[m] += ms * cnt
            gf = 2.0 * B * OH * OW * N * (C // g) * k * k / 1e9
            by = (B * H * W * C + B * OH * OW * N + N * (C // g) * k * k) * 2.0
            floor = max(by / 5.5e12, gf * 1e9 / 1.6e15) * 1e3          # ms: streaming rate / sustained MFMA rate measured on this part
            floors[m] = floors.get(m, 0.0) + floor * cnt
            row += f"  {m} {ms * 1e3:7.1f} us {gf / ms:6.0f} TF/s x{ms / floor:4.1f}"
        lines.append(row)
    if verbose:
        print("\n".join(lines))
    print("TOTAL ms/step: " + "  ".join(f"{m} {v:.3f} (floor {floors[m]:.3f})" for m, v in tot.items()))


if __name__ == "__main__":
    main()
