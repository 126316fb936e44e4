"""Is the default (atomic) reduction mode only a different summation ORDER?  One fixed model state and batch; the gradient of
N steps in atomic mode against the ordered-mode gradient, f32 (where a different order moves the last bits only, so anything
above ~1e-5 would be a race) and f16 (where it also moves 16-bit roundings and max-pool winners: the instance's noise level).
    python tools/atomic_mode_probe.py [runs=24] [scale=n] [batch=16] [imgsz=256]"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
os.environ.setdefault("SY11_DETERMINISTIC", "1")
import torch
from sy11 import _lib
from tests._f16_parity import device_pretrained_state as pretrained_state, pinned_device_step, GAINS, DEV
from types import SimpleNamespace
from sy11.nn.tasks import DetectionModel

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 24
scale = sys.argv[2] if len(sys.argv) > 2 else "n"
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 16
sz = int(sys.argv[4]) if len(sys.argv) > 4 else 256
cfg, nc = f"yolo11{scale}.yaml", 80
_lib.set_option("deterministic", 1)
sd = pretrained_state(cfg, nc, nb, sz, 200)
g = torch.Generator().manual_seed(3)
n = 2 * nb
batch = {"img": torch.rand(nb, 3, sz, sz, generator=g), "batch_idx": torch.arange(nb).repeat_interleave(2).float(),
         "cls": torch.randint(0, nc, (n, 1), generator=g).float(),
         "bboxes": torch.cat((0.3 + 0.4 * torch.rand(n, 2, generator=g), 0.15 + 0.4 * torch.rand(n, 2, generator=g)), 1)}


def step(dtype, pin=None):
    m = DetectionModel(cfg, ch=3, nc=nc, verbose=False)
    m.args = SimpleNamespace(box=GAINS[0], cls=GAINS[1], dfl=GAINS[2])
    m.load_state_dict(sd)
    m._sy11_dtype = dtype
    return pinned_device_step(m.to(DEV).train(), batch, nc, 64.0 if dtype == torch.float16 else 1.0, pin)


for dtype in (torch.float32, torch.float16):
    _lib.set_option("deterministic", 1)
    l0, g0, _, a0, _ = step(dtype)
    from sy11 import ops as K
    w_pin = None
    keys = sorted(g0)
    f0 = torch.cat([g0[k].flatten() for k in keys])
    _lib.set_option("deterministic", 0)
    worst = []
    for r in range(runs):
        l, gr, _, a, _ = step(dtype)
        f = torch.cat([gr[k].flatten() for k in keys])
        d = (f - f0).norm().item() / f0.norm().item()
        k = max(keys, key=lambda k: (gr[k] - g0[k]).norm().item())
        worst.append((d, abs(l - l0) / abs(l0), int((a != a0).sum()), k, (gr[k] - g0[k]).norm().item() / g0[k].norm().item()))
    worst.sort(reverse=True)
    print(f"{dtype}: atomic vs ordered over {runs} runs: whole-gradient distance max {worst[0][0]:.3e}, median {worst[len(worst) // 2][0]:.3e}, min {worst[-1][0]:.3e}")
    for w in worst[:4]:
        print(f"   whole {w[0]:.3e}, loss rel {w[1]:.2e}, {w[2]} anchors assigned differently, largest tensor distance {w[3]} ({w[4]:.3e} of its norm)")
