"""ORACLE — TEST INFRASTRUCTURE ONLY.  Golden-vector generator (dev container only).

Imports the *reference* (/root/reference, read-only) with the four absent third-party packages stubbed
(SURVEY.md appendix A), drives it with closed-form deterministic weights/inputs (oracle.yolo11_ref.closed_form)
and writes small fixtures to tests/golden/.  Fixtures hold only inputs' recipes and the reference's OUTPUT
numbers (samples + moments) — never reference source text.  Run:  python -m oracle.gen_golden
"""
from __future__ import annotations

import importlib.metadata as md
import os
import sys
import types
from pathlib import Path
from unittest.mock import MagicMock

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
OUT = ROOT / "tests" / "golden"
MAX_FULL = 1024


def import_reference():
    os.environ["YOLO_OFFLINE"] = "true"
    os.environ["YOLO_CONFIG_DIR"] = "/tmp/yolo_cfg"
    os.makedirs("/tmp/yolo_cfg", exist_ok=True)
    sys.dont_write_bytecode = True

    class _Stub(types.ModuleType):
        def __getattr__(self, n):
            if n.startswith("__"):
                raise AttributeError(n)
            m = MagicMock(name=f"{self.__name__}.{n}")
            setattr(self, n, m)
            return m

    for n in ["cv2", "thop", "torchvision", "torchvision.ops", "torchvision.transforms", "torchvision.datasets",
              "timm", "timm.layers", "timm.layers.create_act", "timm.layers.helpers", "timm.layers.mlp",
              "timm.layers.norm", "timm.models", "timm.models.layers"]:
        s = _Stub(n)
        s.__path__ = []
        s.__spec__ = None
        sys.modules[n] = s
    sys.modules["cv2"].__version__ = "4.10.0"
    sys.modules["torchvision"].__version__ = "0.25.0"
    _v = md.version
    md.version = lambda name: "0.25.0" if name == "torchvision" else _v(name)
    sys.path.insert(0, "/root/reference")


def summarize(store: dict, name: str, t: torch.Tensor):
    """Full tensor if small, else a strided sample of <=MAX_FULL values; always float64 moments."""
    t = t.detach().to(torch.float32).contiguous()
    flat = t.flatten()
    n = flat.numel()
    stride = max(1, -(-n // MAX_FULL))
    store[name + ".shape"] = np.asarray(t.shape, dtype=np.int64)
    store[name + ".stride"] = np.asarray(stride, dtype=np.int64)
    store[name + ".s"] = flat[::stride].numpy().copy()
    d = flat.double()
    store[name + ".m"] = np.asarray([d.sum(), d.abs().sum(), (d * d).sum(), d.max() if n else 0, d.min() if n else 0],
                                    dtype=np.float64)


def main():
    import_reference()
    sys.path.insert(0, str(ROOT))
    from oracle.yolo11_ref import closed_form, closed_form_state_dict, empty_state_dict, resolve_graph, seeded_image, seeded_state_dict
    from ultralytics.nn.modules import block as B, conv as C, head as H
    from ultralytics.nn.tasks import DetectionModel, yaml_model_load
    from ultralytics.utils import IterableSimpleNamespace
    from ultralytics.utils import ops as uops
    from ultralytics.utils.torch_utils import initialize_weights

    torch.manual_seed(0)
    torch.set_num_threads(4)
    OUT.mkdir(parents=True, exist_ok=True)

    # ---------------------------------------------------------------- per-module fixtures
    cases = {
        "conv_k1": (lambda: C.Conv(32, 64, 1, 1), (2, 32, 8, 8)),
        "conv_k3": (lambda: C.Conv(32, 64, 3, 1), (2, 32, 8, 8)),
        "conv_k3s2": (lambda: C.Conv(32, 64, 3, 2), (2, 32, 10, 10)),
        "conv_k3s2_odd": (lambda: C.Conv(16, 32, 3, 2), (1, 16, 9, 7)),
        "conv_noact": (lambda: C.Conv(32, 48, 1, 1, act=False), (2, 32, 8, 8)),
        "conv_stem": (lambda: C.Conv(3, 16, 3, 2), (2, 3, 16, 16)),
        "dwconv": (lambda: C.DWConv(64, 64, 3), (2, 64, 8, 8)),
        "bottleneck": (lambda: B.Bottleneck(64, 64, True, 1, (3, 3), 0.5), (2, 64, 8, 8)),
        "c3k": (lambda: B.C3k(64, 64, 2, True, 1), (2, 64, 8, 8)),
        "c3k2_plain": (lambda: B.C3k2(64, 128, 1, False, 0.25), (2, 64, 8, 8)),
        "c3k2_c3k": (lambda: B.C3k2(64, 64, 1, True), (2, 64, 8, 8)),
        "sppf": (lambda: B.SPPF(64, 64, 5), (2, 64, 8, 8)),
        "attention": (lambda: B.Attention(128, num_heads=2, attn_ratio=0.5), (2, 128, 6, 5)),
        "psablock": (lambda: B.PSABlock(128, 0.5, 2), (2, 128, 6, 5)),
        "c2psa": (lambda: B.C2PSA(128, 128, 1), (2, 128, 6, 5)),
    }
    store = {}
    for name, (ctor, shape) in cases.items():
        m = ctor()
        initialize_weights(m)
        m.load_state_dict({k: closed_form(name + "." + k, tuple(v.shape)) if v.dtype.is_floating_point else v
                           for k, v in m.state_dict().items()})
        x = closed_form("in." + name, shape, "signed").requires_grad_(True)
        m.train()
        y = m(x)
        g = closed_form("g." + name, tuple(y.shape), "signed")
        (y * g).sum().backward()
        summarize(store, f"{name}.train.y", y)
        summarize(store, f"{name}.train.dx", x.grad)
        for k, p in m.named_parameters():
            summarize(store, f"{name}.train.grad.{k}", p.grad)
        for k, b in m.named_buffers():
            if b.dtype.is_floating_point:
                summarize(store, f"{name}.train.buf.{k}", b)
        m.eval()
        with torch.no_grad():
            summarize(store, f"{name}.eval.y", m(x.detach()))
    # DFL + Detect
    det = H.Detect(nc=5, ch=(32, 64, 128))
    det.stride = torch.tensor([8.0, 16.0, 32.0])
    initialize_weights(det)
    det.load_state_dict({k: closed_form("detect." + k, tuple(v.shape)) if (v.dtype.is_floating_point and not k.startswith("dfl.")) else v
                         for k, v in det.state_dict().items()})
    feats = [closed_form(f"in.detect.{i}", s, "signed").requires_grad_(True)
             for i, s in enumerate([(2, 32, 8, 8), (2, 64, 4, 4), (2, 128, 2, 2)])]
    det.train()
    maps = det([f for f in feats])
    tot = 0
    for i, mp in enumerate(maps):
        summarize(store, f"detect.train.map{i}", mp)
        tot = tot + (mp * closed_form(f"g.detect.{i}", tuple(mp.shape), "signed")).sum()
    tot.backward()
    for i, f in enumerate(feats):
        summarize(store, f"detect.train.dx{i}", f.grad)
    for k, p in det.named_parameters():
        if p.grad is not None:
            summarize(store, f"detect.train.grad.{k}", p.grad)
    det.eval()
    with torch.no_grad():
        y, _ = det([f.detach() for f in feats])
    summarize(store, "detect.eval.y", y)
    np.savez_compressed(OUT / "modules.npz", **store)

    # ---------------------------------------------------------------- tiny end-to-end model (scale t)
    store = {}
    d = yaml_model_load("yolo11n.yaml")
    d["scales"]["t"] = [0.5, 0.125, 1024]
    d["scale"] = "t"
    nc = 4
    model = DetectionModel(d, ch=3, nc=nc, verbose=False)
    model.args = IterableSimpleNamespace(box=7.5, cls=0.5, dfl=1.5)
    tiny_sd = lambda: seeded_state_dict(empty_state_dict(resolve_graph("t", nc=nc)), seed=0)
    assert set(tiny_sd().keys()) == set(model.state_dict().keys())
    model.load_state_dict(tiny_sd())
    Bsz, S = 2, 64
    img = seeded_image((Bsz, 3, S, S), seed=5)
    batch = {
        "img": img,
        "batch_idx": torch.tensor([0, 0, 1, 1, 1], dtype=torch.float32),
        "cls": torch.tensor([[1.0], [3.0], [0.0], [2.0], [1.0]]),
        "bboxes": torch.tensor([[0.30, 0.35, 0.40, 0.50], [0.70, 0.60, 0.35, 0.45], [0.50, 0.50, 0.80, 0.70],
                                [0.25, 0.70, 0.30, 0.40], [0.52, 0.48, 0.60, 0.55]]),
    }
    for k in ("batch_idx", "cls", "bboxes"):
        store["batch." + k] = batch[k].numpy()
    model.train()
    maps = model(img)
    for i, mp in enumerate(maps):
        summarize(store, f"train.map{i}", mp)
    model.zero_grad()
    # fresh BN buffers again so loss forward sees the same statistics state as a single train step would
    model.load_state_dict(tiny_sd())
    crit = model.init_criterion()
    model.criterion = crit
    # capture TAL outputs
    tal_out = {}
    orig = crit.assigner.forward

    def wrapped(*a, **k):
        r = orig(*a, **k)
        tal_out["r"] = r
        return r

    crit.assigner.forward = wrapped
    loss, items = model(batch)
    loss.backward()
    store["loss"] = np.asarray([loss.item()], dtype=np.float64)
    store["loss_items"] = items.double().numpy()
    tl, tb, ts, fg, gi = tal_out["r"]
    store["tal.fg"] = fg.numpy()
    store["tal.gt_idx"] = gi.numpy()
    store["tal.labels"] = tl.numpy()
    summarize(store, "tal.scores", ts)
    summarize(store, "tal.bboxes", tb)
    gn = {}
    for k, p in model.named_parameters():
        if p.grad is not None:
            gn[k] = [p.grad.double().norm().item(), p.grad.double().sum().item()]
    store["grad.names"] = np.asarray(list(gn.keys()))
    store["grad.norm_sum"] = np.asarray(list(gn.values()), dtype=np.float64)
    for k in ("model.0.conv.weight", "model.2.m.0.cv1.conv.weight", "model.10.m.0.attn.qkv.conv.weight",
              "model.23.cv2.0.2.bias", "model.23.cv3.2.2.weight", "model.8.m.0.m.1.cv2.bn.weight"):
        summarize(store, "grad." + k, dict(model.named_parameters())[k].grad)
    for k in ("model.0.bn.running_mean", "model.0.bn.running_var", "model.22.cv2.bn.running_var"):
        summarize(store, "buf." + k, model.state_dict()[k])
    # eval + fused eval
    model.load_state_dict(tiny_sd())
    model.eval()
    with torch.no_grad():
        y, maps = model(img)
    summarize(store, "eval.y", y)
    for i, mp in enumerate(maps):
        summarize(store, f"eval.map{i}", mp)
    model.fuse(verbose=False)
    with torch.no_grad():
        yf, _ = model(img)
    summarize(store, "eval_fused.y", yf)
    np.savez_compressed(OUT / "model_t.npz", **store)

    # ---------------------------------------------------------------- NMS wrapper inputs (core is third party)
    store = {}
    rec = []

    def fake_nms(boxes, scores, iou):
        rec.append((boxes.clone(), scores.clone(), float(iou)))
        return torch.arange(boxes.shape[0])

    sys.modules["torchvision"].ops.nms = fake_nms
    A, ncls = 600, 6
    pred = torch.zeros(2, 4 + ncls, A)
    pred[:, 0] = 40 + 560 * closed_form("nms.cx", (2, A), "input")
    pred[:, 1] = 40 + 560 * closed_form("nms.cy", (2, A), "input")
    pred[:, 2] = 20 + 120 * closed_form("nms.w", (2, A), "input")
    pred[:, 3] = 20 + 120 * closed_form("nms.h", (2, A), "input")
    pred[:, 4:] = closed_form("nms.cls", (2, ncls, A), "input") ** 3
    store["pred"] = pred.numpy()
    for tag, kw in (("best", dict(conf_thres=0.25, iou_thres=0.7, multi_label=False)),
                    ("multi", dict(conf_thres=0.05, iou_thres=0.7, multi_label=True))):
        rec.clear()
        uops.non_max_suppression(pred.clone(), max_det=300, **kw)
        for i, (b, s, iou) in enumerate(rec):
            store[f"{tag}.{i}.boxes"] = b.numpy()
            store[f"{tag}.{i}.scores"] = s.numpy()
        store[f"{tag}.n"] = np.asarray(len(rec))
    np.savez_compressed(OUT / "nms_inputs.npz", **store)
    for f in sorted(OUT.glob("*.npz")):
        print(f.name, f.stat().st_size)


if __name__ == "__main__":
    main()
