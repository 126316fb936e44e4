"""ORACLE — TEST INFRASTRUCTURE ONLY.  Golden vectors for the fusion model variant.

Same recipe as oracle/gen_golden.py (reference imported with stubbed third-party packages, deterministic weights and
inputs, only OUTPUT numbers stored): DDWConv (7x7 / 3x3, stride 2, dilation 2, 8 groups), Fusion('ESChannel') with 2 and
3 inputs, and one end-to-end train / eval step of yolo11s_fusion_sand3_new at 64x64.
Run:  python -m oracle.gen_golden_fusion      -> tests/golden/fusion.npz
"""
from __future__ import annotations

import sys

import numpy as np
import torch

from oracle.gen_golden import OUT, ROOT, import_reference, summarize


def main():
    import_reference()
    sys.path.insert(0, str(ROOT))
    from oracle.yolo11_ref import (GRAPH_FUSION, closed_form, empty_state_dict, resolve_graph, seeded_image, seeded_state_dict)
    from ultralytics.nn.modules import conv as C
    from ultralytics.nn.tasks import DetectionModel, yaml_model_load
    from ultralytics.utils import IterableSimpleNamespace
    from ultralytics.utils.torch_utils import initialize_weights

    torch.manual_seed(0)
    torch.set_num_threads(4)
    store = {}

    def params_closed_form(name, m):
        sd = {}
        for k, v in m.state_dict().items():
            if not v.dtype.is_floating_point:
                sd[k] = v
            elif k.endswith("alpha"):
                sd[k] = closed_form(name + "." + k, tuple(v.shape), "gamma")
            elif k.endswith("gamma") or k.endswith("beta"):
                sd[k] = 0.5 * closed_form(name + "." + k, tuple(v.shape), "signed")
            else:
                sd[k] = closed_form(name + "." + k, tuple(v.shape))
        m.load_state_dict(sd)

    cases = {
        "ddwconv_k7": (lambda: C.DDWConv(64, 128, 7, 2, 2), [(2, 64, 18, 18)]),
        "ddwconv_k3": (lambda: C.DDWConv(128, 64, 3, 2, 2), [(2, 128, 11, 9)]),
        "fusion2": (lambda: C.Fusion([128, 128], "ESChannel"), [(2, 128, 6, 5)] * 2),
        "fusion3": (lambda: C.Fusion([128, 128, 128], "ESChannel"), [(2, 128, 7, 4)] * 3),
    }
    for name, (ctor, shapes) in cases.items():
        m = ctor()
        initialize_weights(m)
        params_closed_form(name, m)
        xs = [closed_form(f"in.{name}.{i}", s, "signed").requires_grad_(True) for i, s in enumerate(shapes)]
        m.train()
        y = m(xs[0] if len(xs) == 1 else xs)
        g = closed_form("g." + name, tuple(y.shape), "signed")
        (y * g).sum().backward()
        summarize(store, f"{name}.train.y", y)
        for i, x in enumerate(xs):
            summarize(store, f"{name}.train.dx{i}", x.grad)
        for k, p in m.named_parameters():
            if p.grad is not None:
                summarize(store, f"{name}.train.grad.{k}", p.grad)
        for k, b in m.named_buffers():
            if b.dtype.is_floating_point:
                summarize(store, f"{name}.train.buf.{k}", b)
        m.eval()
        with torch.no_grad():
            summarize(store, f"{name}.eval.y", m(xs[0].detach() if len(xs) == 1 else [x.detach() for x in xs]))

    # ---------------------------------------------------------------- end-to-end: yolo11s_fusion_sand3_new, nc = 2
    nc = 2
    d = yaml_model_load("yolo11s_fusion_sand3_new.yaml")
    model = DetectionModel(d, ch=3, nc=nc, verbose=False)
    model.args = IterableSimpleNamespace(box=7.5, cls=0.5, dfl=1.5)
    fresh = lambda: seeded_state_dict(empty_state_dict(resolve_graph("s", nc=nc, graph=GRAPH_FUSION)), seed=3)   # noqa: E731
    assert set(fresh().keys()) == set(model.state_dict().keys()), set(fresh().keys()) ^ set(model.state_dict().keys())
    store["e2e.n_params"] = np.asarray(sum(p.numel() for p in model.parameters()))
    model.load_state_dict(fresh())
    img = seeded_image((2, 3, 64, 64), seed=7)
    batch = {
        "img": img,
        "batch_idx": torch.tensor([0, 0, 1], dtype=torch.float32),
        "cls": torch.tensor([[1.0], [0.0], [1.0]]),
        "bboxes": torch.tensor([[0.30, 0.35, 0.40, 0.50], [0.70, 0.60, 0.35, 0.45], [0.50, 0.50, 0.80, 0.70]]),
    }
    for k in ("batch_idx", "cls", "bboxes"):
        store["e2e.batch." + k] = batch[k].numpy()
    model.train()
    maps = model(img)
    for i, mp in enumerate(maps):
        summarize(store, f"e2e.train.map{i}", mp)
    model.zero_grad()
    model.load_state_dict(fresh())
    model.criterion = model.init_criterion()
    loss, items = model(batch)
    loss.backward()
    store["e2e.loss"] = np.asarray([loss.item()], dtype=np.float64)
    store["e2e.loss_items"] = items.double().numpy()
    gn = {k: [p.grad.double().norm().item(), p.grad.double().sum().item()] for k, p in model.named_parameters() if p.grad is not None}
    store["e2e.grad.names"] = np.asarray(list(gn.keys()))
    store["e2e.grad.norm_sum"] = np.asarray(list(gn.values()), dtype=np.float64)
    for k in ("model.11.conv1.conv.weight", "model.13.conv1.conv.weight", "model.17.sab.cv1.weight", "model.17.gsc3.alpha",
              "model.20.gsc3.gamma", "model.26.gsc2.beta", "model.0.conv.weight"):
        summarize(store, "e2e.grad." + k, dict(model.named_parameters())[k].grad)
    model.load_state_dict(fresh())
    model.eval()
    with torch.no_grad():
        y, maps = model(img)
    summarize(store, "e2e.eval.y", y)
    np.savez_compressed(OUT / "fusion.npz", **store)
    print("fusion.npz", (OUT / "fusion.npz").stat().st_size)


if __name__ == "__main__":
    main()
