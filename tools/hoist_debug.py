"""Does head hoisting (SY11_HEAD_HOIST=1) take effect in the trainer's graph path and in a plain eager step?  Prints the branch
sections of the tape and compares ordered-mode gradients hoisted vs not, per layer index."""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
os.environ["SY11_DETERMINISTIC"] = "1"
from types import SimpleNamespace
import torch
import sy11.engine as E
from sy11.nn.tasks import DetectionModel
from sy11.utils.torch_utils import set_deterministic
set_deterministic(True)
DEV = "cuda"
g = torch.Generator().manual_seed(1)
B, nc = 8, 80
n = 3 * B
batch = {"img": torch.rand(B, 3, 160, 160, generator=g).to(DEV), "batch_idx": torch.arange(B).repeat_interleave(3).float().to(DEV),
         "cls": torch.randint(0, nc, (n, 1), generator=g).float().to(DEV),
         "bboxes": torch.cat((0.25 + 0.5 * torch.rand(n, 2, generator=g), 0.1 + 0.4 * torch.rand(n, 2, generator=g)), 1).to(DEV)}
seen = []
orig_init = E.Ctx.__init__


def spy(self, *a, **k):
    orig_init(self, *a, **k)
    seen.append(self)


E.Ctx.__init__ = spy


def step(hoist, dtype):
    E._HEAD_HOIST = hoist
    torch.manual_seed(3)
    m = DetectionModel("yolo11n.yaml", nc=nc, verbose=False)
    m.args = SimpleNamespace(box=7.5, cls=0.5, dfl=1.5)
    m._sy11_dtype = dtype
    m = m.to(DEV).train()
    seen.clear()
    loss, _ = m(batch)
    ec = seen[-1]
    print(f"  hoist {hoist}: hoist_head {ec.hoist_head}, use_branches {ec.use_branches}, tape {len(ec.tape)} closures, branch sections {ec.tape_branches}")
    (loss * 64.0).backward()
    return {k: p.grad.float().clone() for k, p in m.named_parameters() if p.grad is not None}


for dtype in (torch.float32, torch.float16):
    print(dtype)
    a, b = step(False, dtype), step(True, dtype)
    c = step(True, dtype)
    per = {}
    for k in a:
        li = int(k.split(".")[1])
        d = per.setdefault(li, [0, 0, 0])
        d[0] += int((a[k] != b[k]).sum()); d[1] += a[k].numel(); d[2] += int((b[k] != c[k]).sum())
    print("  layer: elements differing hoisted vs not / total   (hoisted run twice: differing)")
    for li in sorted(per):
        print(f"   {li:2d}: {per[li][0]:8d} / {per[li][1]:8d}   ({per[li][2]})")

print("trainer, graph replay, f16:")
from oracle import yolo11_ref as R
from sy11.engine.trainer import DetectionTrainer


def trainer_run(hoist):
    E._HEAD_HOIST = hoist
    m = DetectionModel("yolo11n.yaml", nc=80, verbose=False)
    m.load_state_dict(R.seeded_state_dict(R.empty_state_dict(R.resolve_graph("n", nc=80)), seed=4))
    tr = DetectionTrainer(m, batch_size=8, device=DEV, overrides={"amp": True, "nbs": 8, "warmup_epochs": 0, "deterministic": True}, graphs=True)
    seen.clear()
    for i in range(5):
        tr.train_step(dict(batch))
    print(f"  hoist {hoist}: {len(seen)} contexts; hoist_head {[c.hoist_head for c in seen][:6]}; branch sections of the last: {seen[-1].tape_branches}")
    return tr.flat.flat.clone()


w0, w1 = trainer_run(False), trainer_run(True)
print("  weights differing hoisted vs not:", int((w0 != w1).sum()), "of", w0.numel())
