"""The criterion's gradient on IDENTICAL logits, device (HIP) vs oracle (f32 autograd), over model states trained in atomic mode.
Where a level differs by more than 1e-3 of its norm: which elements, and how close that anchor's predicted box edges are to its
target's (the IoU / enclosing-box terms of CIoU have a gradient jump where they coincide).
    python tools/criterion_kink_probe.py [states=12]"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
os.environ["SY11_DETERMINISTIC"] = "0"
from types import SimpleNamespace
import torch
from oracle import loss_ref
from sy11 import _lib
from sy11.nn.tasks import DetectionModel
from tests._f16_parity import device_pretrained_state as pretrained_state, pinned_device_step, oracle_assignment, device_targets, GAINS, DEV, STRIDES

states = int(sys.argv[1]) if len(sys.argv) > 1 else 12
cfg, nc, nb, sz = "yolo11n.yaml", 80, 16, 256
g = torch.Generator().manual_seed(3)
img = torch.rand(nb, 3, sz, sz, generator=g)
n = 2 * nb
batch = {"img": img, "batch_idx": torch.arange(nb).repeat_interleave(2).float(), "cls": torch.randint(0, nc, (n, 1), generator=g).float(),
         "bboxes": torch.cat((0.3 + 0.4 * torch.rand(n, 2, generator=g), 0.15 + 0.4 * torch.rand(n, 2, generator=g)), 1)}


def dev(sd, pin=None):
    m = DetectionModel(cfg, ch=3, nc=nc, verbose=False)
    m.args = SimpleNamespace(box=GAINS[0], cls=GAINS[1], dfl=GAINS[2])
    m.load_state_dict(sd)
    m._sy11_dtype = torch.float16
    return pinned_device_step(m.to(DEV).train(), batch, nc, 64.0, pin)


for s in range(states):
    _lib.set_option("deterministic", 0)
    sd = pretrained_state(cfg, nc, nb, sz, 200)
    _, _, maps, assign, _ = dev(sd)
    tg = oracle_assignment(maps, batch, nc)
    B, A = assign.shape
    pin = device_targets(tg, B, A)
    l1, _, maps1, _, dm1 = dev(sd, pin)
    leaves = [t.clone().requires_grad_(True) for t in maps1]
    la, _ = loss_ref.detection_loss(leaves, batch, nc=nc, pinned=tg)
    (la * 64.0).backward()
    rel = [(d - t.grad).norm().item() / t.grad.norm().item() for d, t in zip(dm1, leaves)]
    print(f"state {s}: loss rel {abs(l1 - la.item()) / abs(la.item()):.1e}; d loss / d maps per level " + " ".join(f"{r:.2e}" for r in rel), flush=True)
    for lv, (d, t) in enumerate(zip(dm1, leaves)):
        if rel[lv] > 1e-3:
            diff = (d - t.grad).abs()
            per_anchor = diff.sum(1)                                      # (B, H, W)
            flat = per_anchor.flatten()
            top = torch.topk(flat, 3).indices
            H, W = d.shape[2:]
            off = sum(m.shape[2] * m.shape[3] for m in maps1[:lv])
            for ix in top.tolist():
                b, y, x = ix // (H * W), (ix % (H * W)) // W, ix % W
                a = off + y * W + x
                box_part, cls_part = diff[b, :64, y, x].sum().item(), diff[b, 64:, y, x].sum().item()
                t_lab, t_box, t_sc, fg, gi = tg
                # the anchor's predicted box from the oracle's own decode
                logits = t.detach()[b, :64, y, x].view(4, 16)
                dist = (logits.softmax(-1) * torch.arange(16.)).sum(-1) * STRIDES[lv]
                cx, cy = (x + 0.5) * STRIDES[lv], (y + 0.5) * STRIDES[lv]
                pb = torch.tensor([cx - dist[0], cy - dist[1], cx + dist[2], cy + dist[3]])
                print(f"    level {lv} image {b} cell ({y},{x}): |diff| box part {box_part:.3e}, class part {cls_part:.3e}; foreground {bool(fg[b, a])}; "
                      f"predicted box {[round(v, 4) for v in pb.tolist()]} target {[round(v, 4) for v in t_box[b, a].tolist()]} edge gaps {[f'{v:.2e}' for v in (pb - t_box[b, a]).tolist()]}")
