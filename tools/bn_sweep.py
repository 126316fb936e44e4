"""Time the BatchNorm+SiLU passes (fwd apply, bwd reduce, bwd apply) on every BN'd conv output shape of yolo11s
(640x640, batch 64, f16) and print achieved HBM bandwidth.   python tools/bn_sweep.py [-v] [--row-map 0|1] [--eager]
Launches are timed inside a replayed hipGraph of 10 calls (what the trainer runs; a small kernel's eager time is the host's call
rate, not the kernel); --eager times plain calls.  --row-map: the row walk of the kernels (elementwise.hip RowWalk), default both."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
sys.path.insert(0, str(ROOT / "tools"))
import torch
from sy11 import _lib, ops
from conv_sweep import LAYERS


def timed(fn, reps, eager):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if eager:
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
    else:
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.cuda.graph(g, stream=side):
            for _ in range(reps):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        g.replay()
        torch.cuda.synchronize()
        e0.record()
        g.replay()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    maps = [int(v) for v in sys.argv[sys.argv.index("--row-map") + 1].split(",")] if "--row-map" in sys.argv else [0, 1]
    for rm in maps:
        _lib.set_option("row_map", rm)
        print(f"row_map = {rm} ({'chunked' if rm else 'strided (r01)'})")
        sweep("--eager" in sys.argv)


def sweep(eager):
    verbose = "-v" in sys.argv
    B, dt, reps = 64, torch.float16, 10
    tot = {"fwd": 0.0, "reduce": 0.0, "apply": 0.0}
    byt = {"fwd": 0.0, "reduce": 0.0, "apply": 0.0}
    shapes = {}
    for (H, W, C, N, k, s, g, cnt) in LAYERS + [(640, 640, 3, 32, 3, 2, 1, 1)]:
        OH, OW = ops.conv_out_hw(H, W, k, s, k // 2)
        shapes[(OH, OW, N)] = shapes.get((OH, OW, N), 0) + cnt
    for (OH, OW, N), cnt in sorted(shapes.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[0][2]):
        y = torch.randn(B, OH, OW, N, device="cuda", dtype=dt)
        dz = torch.randn_like(y)
        z = torch.empty_like(y)
        dy = torch.empty_like(y)
        f = lambda *sh: torch.rand(*sh, device="cuda") + 0.5
        mean, rstd, scale, shift, gamma = f(N), f(N), f(N), f(N), f(N)
        import os
        ns = int(os.environ.get("BN_SLOTS", "8"))
        sg, sgx = torch.zeros(ns, N, device="cuda"), torch.zeros(ns, N, device="cuda")      # 8 slots, as nn/modules/conv.py
        dgam, dbet = torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")
        act = os.environ.get("BN_SILU", "1") != "0"            # 0: the passes without the activation (how much of them is SiLU arithmetic)
        fns = {"fwd": (lambda: ops.bn_act_fwd(y, scale, shift, z, act), 2),
               "reduce": (lambda: ops.bn_act_bwd_reduce(y, dz, mean, rstd, scale, shift, act, sg, sgx), 2),
               "apply": (lambda: ops.bn_act_bwd_apply(y, dz, mean, rstd, scale, shift, gamma, act, sg, sgx, dy, dgam, dbet), 3)}
        row = f"{OH:3d}x{OW:<3d}x{N:<4d} x{cnt:<2d}"
        for name, (fn, passes) in fns.items():
            ms = timed(fn, reps, eager)
            nb = passes * y.numel() * 2
            tot[name] += ms * cnt
            byt[name] += nb * cnt
            row += f"  {name} {ms * 1e3:7.1f} us {nb / ms / 1e9:5.2f} TB/s"
        if verbose:
            print(row)
    print("TOTAL ms/step: " + "  ".join(f"{m} {v:.3f} ({byt[m] / v / 1e9:.2f} TB/s)" for m, v in tot.items()))


if __name__ == "__main__":
    main()
