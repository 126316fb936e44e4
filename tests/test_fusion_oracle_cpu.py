"""CPU: the oracle's fusion-variant restatement (DDWConv, GCT, WeightedSpatialAttention, Fusion('ESChannel'), graph of
yolo11_fusion_sand3_new) must reproduce the REFERENCE's numbers in tests/golden/fusion.npz (oracle/gen_golden_fusion.py)."""
import numpy as np
import pytest
import torch

from oracle import loss_ref, yolo11_ref as R
from tests._golden import check, load

CASES = {
    "ddwconv_k7": (lambda sd, xs, tr: R.ddwconv(sd, "", xs[0], 7, 2, 2, tr), [(2, 64, 18, 18)]),
    "ddwconv_k3": (lambda sd, xs, tr: R.ddwconv(sd, "", xs[0], 3, 2, 2, tr), [(2, 128, 11, 9)]),
    "fusion2": (lambda sd, xs, tr: R.fusion_eschannel(sd, "", xs), [(2, 128, 6, 5)] * 2),
    "fusion3": (lambda sd, xs, tr: R.fusion_eschannel(sd, "", xs), [(2, 128, 7, 4)] * 3),
}


def fusion_param(name, k, shape):
    """Closed-form parameter recipe of oracle/gen_golden_fusion.py:params_closed_form."""
    if k.endswith("alpha"):
        return R.closed_form(name + "." + k, shape, "gamma")
    if k.endswith("gamma") or k.endswith("beta"):
        return 0.5 * R.closed_form(name + "." + k, shape, "signed")
    return R.closed_form(name + "." + k, shape)


def module_state(gold, name):
    pre_g, pre_b = name + ".train.grad.", name + ".train.buf."
    keys = {k[len(pre_g):-len(".shape")] for k in gold if k.startswith(pre_g) and k.endswith(".shape")}
    bufs = {k[len(pre_b):-len(".shape")] for k in gold if k.startswith(pre_b) and k.endswith(".shape")}
    sd = {k: fusion_param(name, k, tuple(gold[f"{pre_g}{k}.shape"].tolist())).requires_grad_(True) for k in keys}
    for k in bufs:
        sd[k] = R.closed_form(name + "." + k, tuple(gold[f"{pre_b}{k}.shape"].tolist()))
    if name.startswith("fusion"):                        # the unused GCT of the other arity gets no gradient: not in the fixture
        other = "gsc3." if name == "fusion2" else "gsc2."
        assert not any(k.startswith(other) for k in keys)
    return sd, keys, bufs


@pytest.mark.parametrize("name", list(CASES))
def test_fusion_module_matches_reference(name):
    gold = load("fusion.npz")
    fn, shapes = CASES[name]
    sd, keys, bufs = module_state(gold, name)
    xs = [R.closed_form(f"in.{name}.{i}", s, "signed").requires_grad_(True) for i, s in enumerate(shapes)]
    y = fn(sd, xs, True)
    g = R.closed_form("g." + name, tuple(y.shape), "signed")
    (y * g).sum().backward()
    check(gold, f"{name}.train.y", y)
    for i, x in enumerate(xs):
        check(gold, f"{name}.train.dx{i}", x.grad, rtol=2e-4)
    for k in keys:
        check(gold, f"{name}.train.grad.{k}", sd[k].grad, rtol=5e-4, atol=2e-5)
    for k in bufs:
        check(gold, f"{name}.train.buf.{k}", sd[k])
    with torch.no_grad():
        check(gold, f"{name}.eval.y", fn(sd, [x.detach() for x in xs], False))


def fusion_model(nc=2):
    layers = R.resolve_graph("s", nc=nc, graph=R.GRAPH_FUSION)
    return layers, R.seeded_state_dict(R.empty_state_dict(layers), seed=3)


def test_fusion_model_inventory():
    gold = load("fusion.npz")
    layers, sd = fusion_model()
    n = sum(v.numel() for k, v in sd.items() if "running" not in k and "num_batches" not in k)
    assert n == int(gold["e2e.n_params"]) == 6824734          # SURVEY.md §8 row 15 [probe]


def test_fusion_model_train_and_eval_match_reference():
    gold = load("fusion.npz")
    layers, sd = fusion_model()
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    batch = {"img": R.seeded_image((2, 3, 64, 64), seed=7), "batch_idx": torch.from_numpy(gold["e2e.batch.batch_idx"]),
             "cls": torch.from_numpy(gold["e2e.batch.cls"]), "bboxes": torch.from_numpy(gold["e2e.batch.bboxes"])}
    maps = R.forward(sd, layers, batch["img"], train=True)
    for i, mp in enumerate(maps):
        check(gold, f"e2e.train.map{i}", mp, rtol=5e-4, atol=5e-5)
    loss, items = loss_ref.detection_loss(maps, batch, nc=2)
    assert abs(loss.item() - gold["e2e.loss"][0]) <= 2e-4 * abs(gold["e2e.loss"][0])
    np.testing.assert_allclose(items.double().numpy(), gold["e2e.loss_items"], rtol=2e-4)
    loss.backward()
    names = [str(n) for n in gold["e2e.grad.names"]]
    gmax = float(gold["e2e.grad.norm_sum"][:, 0].max())
    for n, (gn, gs) in zip(names, gold["e2e.grad.norm_sum"]):
        assert abs(sd[n].grad.double().norm().item() - gn) <= 5e-3 * gn + 1e-5 * gmax, n
    for k in ("model.11.conv1.conv.weight", "model.13.conv1.conv.weight", "model.17.sab.cv1.weight", "model.17.gsc3.alpha",
              "model.20.gsc3.gamma", "model.26.gsc2.beta", "model.0.conv.weight"):
        check(gold, "e2e.grad." + k, sd[k].grad, rtol=5e-3, atol=5e-4)
    layers, sd = fusion_model()
    with torch.no_grad():
        y, _ = R.forward(sd, layers, batch["img"], train=False)
    check(gold, "e2e.eval.y", y, rtol=5e-4, atol=5e-5)
