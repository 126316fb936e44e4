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
CFG = int(sys.argv[1]) if len(sys.argv) > 1 else 20
_lib.set_option("igemm_cfg", CFG)
names = ["early piece issue (group 1)", "mfma + reads", "late piece issue (group 0) + waits", "barrier"]
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
    if CFG == 25:
        pb = (C.c_uint64 * 2048)()
        _lib.check(_lib.load().sy11_debug_stamps_persistent(pb), "sy11_debug_stamps_persistent")
        rec = [list(pb)[8 * i:8 * i + 8] for i in range(256)]
        rec = [r for r in rec if r[4]]
        med = lambda k: sorted(r[k] / r[4] for r in rec)[len(rec) // 2]
        print(f"{H}x{W} {Cc}->{N} k{k}s{s}: persistent, {len(rec)} workgroups, {sum(r[4] for r in rec)} tiles; median cycles per tile: first-stage wait {med(0):.0f}, "
              f"main loop {med(1):.0f}, next-tile setup + issue {med(2):.0f}, epilogue {med(3):.0f}; workgroup span {sorted(r[5] for r in rec)[len(rec) // 2] / 100:.1f} us")
        continue
    buf = (C.c_uint64 * (16 + 8 * 2048))()
    buf[0] = 1
    _lib.check(_lib.load().sy11_debug_stamps(buf), "sy11_debug_stamps")
    v = list(buf)
    print(f"{H}x{W} {Cc}->{N} k{k}s{s}")
    nwg = min(-(-B * OH * OW // 256) * -(-N // (256 if CFG == 24 else 128)), 2048)
    rec = [v[16 + 8 * i:16 + 8 * i + 8] for i in range(nwg)]
    t0 = min(r[0] for r in rec)
    span = (max(r[1] for r in rec) - t0) / 100.0
    dur = sorted((r[1] - r[0]) / 100.0 for r in rec)
    ghz = sorted(sum(r[3:7]) / max(r[1] - r[0], 1) * 0.1 for r in rec)
    starts = sorted((r[0] - t0) / 100.0 for r in rec)
    print(f"  {nwg} workgroups over {span:.1f} us; per workgroup {dur[0]:.1f} / {dur[len(dur) // 2]:.1f} / {dur[-1]:.1f} us (min / median / max); "
          f"clock {ghz[0]:.2f} / {ghz[len(ghz) // 2]:.2f} / {ghz[-1]:.2f} GHz; start times (us) at workgroup 0, 255, 256, 511, 512, last: "
          + ", ".join(f"{starts[i]:.1f}" for i in (0, min(255, nwg - 1), min(256, nwg - 1), min(511, nwg - 1), min(512, nwg - 1), nwg - 1)))
    per_xcc = {}
    for r in rec:
        per_xcc.setdefault(r[2] & 15, []).append((r[1] - r[0]) / 100.0)
    med = lambda k: sorted(r[k] for r in rec)[len(rec) // 2]
    print(f"  median cycles per workgroup: address setup {med(3)}, prologue (first stages in flight, stage 0 in registers) {med(4)}, main loop {med(5)}, epilogue {med(6)}")
    print("  per XCC: " + ", ".join(f"{x}: {len(d)} wgs, mean {sum(d) / len(d):.1f} us" for x, d in sorted(per_xcc.items())))
    for g in range(2):
        r = v[g * 8:(g + 1) * 8]
        n = max(r[6], 1)
        print(f"  group {g}: {r[6]} steps, kernel {r[7]} cycles, loop {sum(r[:6])} = {sum(r[:6]) / n:.0f} per step: " +
              ", ".join(f"{nm} {r[i] / n:.0f}" for i, nm in enumerate(names)))
