"""Given a saved susceptible state (tools/f16_event_probe.py), rerun the f16 device step in atomic mode until an unusual run shows
up, recording for every BatchNorm backward (in call = backward order) the arriving gradient dz and the produced dy; print the
distance of the unusual and of a usual run from run 0 per call, to see where the difference enters.
    python tools/f16_event_locate.py state.pt [runs=40]"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
os.environ["SY11_DETERMINISTIC"] = "0"
from types import SimpleNamespace
import torch
from sy11 import _lib, ops
from sy11.nn.tasks import DetectionModel
from tests._f16_parity import pinned_device_step, GAINS, DEV

sd = torch.load(sys.argv[1])
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfg, nc, nb, sz = "yolo11n.yaml", 80, 16, 256
g = torch.Generator().manual_seed(3)
img = torch.rand(nb, 3, sz, sz, generator=g)
n = 2 * nb
batch = {"img": img, "batch_idx": torch.arange(nb).repeat_interleave(2).float(), "cls": torch.randint(0, nc, (n, 1), generator=g).float(),
         "bboxes": torch.cat((0.3 + 0.4 * torch.rand(n, 2, generator=g), 0.15 + 0.4 * torch.rand(n, 2, generator=g)), 1)}
_lib.set_option("deterministic", 0)
rec = []
orig = ops.bn_act_bwd_apply


def spy(y, dz, mean, rstd, scale, shift, gamma, silu, s0, s1, dy, *a, **k):
    r = orig(y, dz, mean, rstd, scale, shift, gamma, silu, s0, s1, dy, *a, **k)
    torch.cuda.synchronize()
    rec.append((tuple(y.shape), dz.detach().float().cpu().clone(), dy.detach().float().cpu().clone(), s0.detach().float().cpu().clone(), s1.detach().float().cpu().clone(),
                y.detach().float().cpu().clone()))
    return r


ops.bn_act_bwd_apply = spy
import sy11.nn.modules.conv as C
if hasattr(C, "ops"):
    C.ops.bn_act_bwd_apply = spy


def dev():
    m = DetectionModel(cfg, ch=3, nc=nc, verbose=False)
    m.args = SimpleNamespace(box=GAINS[0], cls=GAINS[1], dfl=GAINS[2])
    m.load_state_dict(sd)
    m._sy11_dtype = torch.float16
    rec.clear()
    _, grads, _, _, _ = pinned_device_step(m.to(DEV).train(), batch, nc, 64.0)
    return grads, list(rec)


def dist(a, b):
    keys = sorted(a)
    fa, fb = torch.cat([a[k].flatten() for k in keys]), torch.cat([b[k].flatten() for k in keys])
    return (fa - fb).norm().item() / fb.norm().item()


base = dev()
usual = None
for r in range(runs):
    cur = dev()
    d = dist(cur[0], base[0])
    print(f"run {r}: {d:.2e}", flush=True)
    if d < 1e-2 and usual is None:
        usual = cur
    if d > 1e-2 and usual is not None:
        print("unusual run; per BatchNorm backward in call order: shape | dz: unusual vs base (usual vs base) | dy: same | sums s0 s1: same | stored y equal")
        for i, (b, u, c) in enumerate(zip(base[1], usual[1], cur[1])):
            def rel(p, q):
                return (p - q).norm().item() / (q.norm().item() + 1e-30)
            print(f"  {i:3d} {str(b[0]):22s} dz {rel(c[1], b[1]):.2e} ({rel(u[1], b[1]):.2e})  dy {rel(c[2], b[2]):.2e} ({rel(u[2], b[2]):.2e})  "
                  f"s0 {rel(c[3].sum(0), b[3].sum(0)):.2e} ({rel(u[3].sum(0), b[3].sum(0)):.2e})  s1 {rel(c[4].sum(0), b[4].sum(0)):.2e} ({rel(u[4].sum(0), b[4].sum(0)):.2e})  "
                  f"y {rel(c[5], b[5]):.1e} ({rel(u[5], b[5]):.1e})  |dz| {b[1].norm().item():.2e} max {b[1].abs().max().item():.2e}")
        break
