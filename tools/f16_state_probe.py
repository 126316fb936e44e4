"""How far is the 16-bit gradient from the 32-bit one, on the device and in the oracle, for model states that come out of
atomic-mode (non-reproducible) device training?  For each of R states: whole-gradient distances
  dev16-o16 (what tests/_f16_parity.py bounds), dev16-o32 and o16-o32 (each side's own 16-bit error), dev32-o32 (sanity).
If dev16-o16 ~ sqrt(2) x the 16-bit errors, the two sides carry independent rounding noise of the same size and the state is
simply ill-conditioned; a dev16-o32 well above o16-o32 would be a device bias.
    python tools/f16_state_probe.py [states=6]"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
os.environ["SY11_DETERMINISTIC"] = "0"
from types import SimpleNamespace
import torch
from oracle import loss_ref, yolo11_ref as R
from sy11 import _lib
from sy11.nn.tasks import DetectionModel
from tests._f16_parity import device_pretrained_state as pretrained_state, pinned_device_step, oracle_assignment, device_targets, GAINS, DEV

states = int(sys.argv[1]) if len(sys.argv) > 1 else 6
cfg, nc, nb, sz = "yolo11n.yaml", 80, 16, 256
layers = R.resolve_graph("n", nc=nc)
g = torch.Generator().manual_seed(3)
img = torch.rand(nb, 3, sz, sz, generator=g)
n = 2 * nb
batch = {"img": img, "batch_idx": torch.arange(nb).repeat_interleave(2).float(), "cls": torch.randint(0, nc, (n, 1), generator=g).float(),
         "bboxes": torch.cat((0.3 + 0.4 * torch.rand(n, 2, generator=g), 0.15 + 0.4 * torch.rand(n, 2, generator=g)), 1)}


def dev(sd, dtype, pin=None):
    m = DetectionModel(cfg, ch=3, nc=nc, verbose=False)
    m.args = SimpleNamespace(box=GAINS[0], cls=GAINS[1], dfl=GAINS[2])
    m.load_state_dict(sd)
    m._sy11_dtype = dtype
    return pinned_device_step(m.to(DEV).train(), batch, nc, 64.0 if dtype == torch.float16 else 1.0, pin)


def orc(sd, emu, tg):
    o = {k: v.clone() for k, v in sd.items()}
    for k, v in o.items():
        if v.dtype.is_floating_point and "running" not in k and ".dfl." not in k:
            v.requires_grad_(True)
    if emu:
        with R.emulate_f16():
            maps = R.forward(o, layers, img, train=True)
    else:
        maps = R.forward(o, layers, img, train=True)
    loss, _ = loss_ref.detection_loss(maps, batch, nc=nc, pinned=tg)
    S = 64.0 if emu else 1.0
    (loss * S).backward()
    return {k: v.grad / S for k, v in o.items() if v.requires_grad and v.grad is not None}


def dist(a, b, keys):
    fa, fb = torch.cat([a[k].flatten() for k in keys]), torch.cat([b[k].flatten() for k in keys])
    return (fa - fb).norm().item() / fb.norm().item()


worst = None
for s in range(states):
    _lib.set_option("deterministic", 0)
    sd = pretrained_state(cfg, nc, nb, sz, 200)
    _lib.set_option("deterministic", 1)
    _, _, maps, _, _ = dev(sd, torch.float16)
    tg = oracle_assignment(maps, batch, nc)
    B, A = maps[0].shape[0], sum(m.shape[2] * m.shape[3] for m in maps)
    pin = device_targets(tg, B, A)
    d16, d32 = dev(sd, torch.float16, pin)[1], dev(sd, torch.float32, pin)[1]
    o16, o32 = orc(sd, True, tg), orc(sd, False, tg)
    keys = [k for k in d16 if k in o16]
    row = (dist(d16, o16, keys), dist(d16, o32, keys), dist(o16, o32, keys), dist(d32, o32, keys))
    stem = ["model.0.conv.weight", "model.1.conv.weight"]
    print(f"state {s}: dev16-o16 {row[0]:.3e}  dev16-o32 {row[1]:.3e}  o16-o32 {row[2]:.3e}  dev32-o32 {row[3]:.3e}   | first two convs only: "
          f"{dist(d16, o16, stem):.3e} {dist(d16, o32, stem):.3e} {dist(o16, o32, stem):.3e} {dist(d32, o32, stem):.3e}", flush=True)
    if worst is None or row[0] > worst[0]:
        worst = (row[0], sd)
torch.save(worst[1], ROOT / "gpurun_out" / "f16_worst_state.pt")
print("saved the worst state:", worst[0])
