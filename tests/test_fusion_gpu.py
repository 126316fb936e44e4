"""GPU parity of the fusion model variant: grouped / dilated DDWConv, Fusion('ESChannel') with its
GCT gate and spatial attention, and the whole yolo11s_fusion_sand3_new graph, against the REFERENCE's numbers in
tests/golden/fusion.npz (fp32: 1e-3 class tolerances as for the base model) and against the oracle in fp16."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import loss_ref, yolo11_ref as R
from tests._golden import check, load
from tests.test_fusion_oracle_cpu import fusion_param

pytestmark = pytest.mark.gpu
DEV = "cuda"

CASES = {
    "ddwconv_k7": (lambda M: M.DDWConv(64, 128, 7, 2, 2), [(2, 64, 18, 18)]),
    "ddwconv_k3": (lambda M: M.DDWConv(128, 64, 3, 2, 2), [(2, 128, 11, 9)]),
    "fusion2": (lambda M: M.Fusion([128, 128], "ESChannel"), [(2, 128, 6, 5)] * 2),
    "fusion3": (lambda M: M.Fusion([128, 128, 128], "ESChannel"), [(2, 128, 7, 4)] * 3),
}


def build(name, dtype=torch.float32):
    from sy11.nn import modules as M
    from sy11.utils.torch_utils import initialize_weights
    m = CASES[name][0](M)
    initialize_weights(m)
    m.load_state_dict({k: fusion_param(name, k, tuple(v.shape)) if v.dtype.is_floating_point else v for k, v in m.state_dict().items()})
    m._sy11_dtype = dtype
    return m.to(DEV)


@pytest.mark.parametrize("name", list(CASES))
def test_fusion_module_matches_reference_fp32(name):
    gold = load("fusion.npz")
    m = build(name)
    xs = [R.closed_form(f"in.{name}.{i}", s, "signed").to(DEV).requires_grad_(True) for i, s in enumerate(CASES[name][1])]
    m.train()
    y = m(xs[0] if len(xs) == 1 else xs)
    g = R.closed_form("g." + name, tuple(y.shape), "signed").to(DEV)
    (y * g).sum().backward()
    check(gold, f"{name}.train.y", y, rtol=1e-3, atol=1e-4)
    for i, x in enumerate(xs):
        check(gold, f"{name}.train.dx{i}", x.grad, rtol=2e-3, atol=2e-4)
    for k, p in m.named_parameters():
        if f"{name}.train.grad.{k}.shape" in gold:
            check(gold, f"{name}.train.grad.{k}", p.grad, rtol=3e-3, atol=3e-4)
        else:                                            # the GCT of the other arity is never touched
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
    for k, b in m.named_buffers():
        if b.dtype.is_floating_point:
            check(gold, f"{name}.train.buf.{k}", b, rtol=1e-3, atol=1e-4)
    m.eval()
    with torch.no_grad():
        check(gold, f"{name}.eval.y", m(xs[0].detach() if len(xs) == 1 else [x.detach() for x in xs]), rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("name", ["ddwconv_k7", "fusion3"])
def test_fusion_module_fp16_vs_reference(name):
    """fp16 activations (the reference's AMP dtype), f32 statistics: 2e-2 of the output scale."""
    gold = load("fusion.npz")
    m = build(name, torch.float16)
    xs = [R.closed_form(f"in.{name}.{i}", s, "signed").to(DEV).requires_grad_(True) for i, s in enumerate(CASES[name][1])]
    m.train()
    y = m(xs[0] if len(xs) == 1 else xs)
    g = R.closed_form("g." + name, tuple(y.shape), "signed").to(DEV)
    (y.float() * g).sum().backward()
    check(gold, f"{name}.train.y", y.float(), rtol=2e-2, atol=2e-2)
    for i, x in enumerate(xs):
        check(gold, f"{name}.train.dx{i}", x.grad.float(), rtol=3e-2, atol=3e-2)


def fusion_model(dtype=torch.float32, nc=2):
    from sy11.nn.tasks import DetectionModel
    m = DetectionModel("yolo11s_fusion_sand3_new.yaml", ch=3, nc=nc, verbose=False)
    m.args = SimpleNamespace(box=7.5, cls=0.5, dfl=1.5)
    m.load_state_dict(fusion_sd(nc))
    m._sy11_dtype = dtype
    return m.to(DEV)


def fusion_sd(nc=2):
    return R.seeded_state_dict(R.empty_state_dict(R.resolve_graph("s", nc=nc, graph=R.GRAPH_FUSION)), seed=3)


def test_fusion_model_state_dict_layout():
    m = fusion_model()
    assert set(m.state_dict().keys()) == set(fusion_sd().keys())
    assert sum(p.numel() for p in m.parameters()) == 6824734


def test_fusion_model_train_step_matches_reference_fp32():
    gold = load("fusion.npz")
    m = fusion_model()
    batch = {"img": R.seeded_image((2, 3, 64, 64), seed=7).to(DEV), "batch_idx": torch.from_numpy(gold["e2e.batch.batch_idx"]).to(DEV),
             "cls": torch.from_numpy(gold["e2e.batch.cls"]).to(DEV), "bboxes": torch.from_numpy(gold["e2e.batch.bboxes"]).to(DEV)}
    m.train()
    maps = m(batch["img"])
    for i, mp in enumerate(maps):       # P5 map: train-mode BN over 2 x 2 x 2 = 8 samples amplifies the run-to-run f32 summation order
        check(gold, f"e2e.train.map{i}", mp, rtol=1e-3, atol=1e-4 if i < 2 else 1e-3)      # of the statistic atomics (tiles are pinned)
    m.load_state_dict(fusion_sd())
    loss, items = m(batch)
    loss.backward()
    assert abs(loss.item() - gold["e2e.loss"][0]) <= 1e-3 * abs(gold["e2e.loss"][0]), (loss.item(), gold["e2e.loss"][0])
    np.testing.assert_allclose(items.double().cpu().numpy(), gold["e2e.loss_items"], rtol=1e-3)
    params = dict(m.named_parameters())
    gmax = float(gold["e2e.grad.norm_sum"][:, 0].max())
    for n, (gn, gs) in zip([str(n) for n in gold["e2e.grad.names"]], gold["e2e.grad.norm_sum"]):
        g = params[n].grad.double()
        assert abs(g.norm().item() - gn) <= 1e-2 * gn + 1e-5 * gmax, (n, g.norm().item(), gn)
    for k in ("model.11.conv1.conv.weight", "model.13.conv1.conv.weight", "model.17.sab.cv1.weight", "model.17.gsc3.alpha",
              "model.20.gsc3.gamma", "model.26.gsc2.beta", "model.0.conv.weight"):
        check(gold, "e2e.grad." + k, params[k].grad, rtol=5e-3, atol=1e-3)
    m.load_state_dict(fusion_sd())
    m.eval()
    with torch.no_grad():
        y, _ = m(batch["img"])
    check(gold, "e2e.eval.y", y, rtol=1e-3, atol=1e-4)


def test_fusion_model_f16_path_matches_f16_emulating_oracle():
    """configs[4]'s dtype on configs[4]'s model: yolo11s_fusion_sand3_new (nc = 2) in f16 against the oracle under emulate_f16
    — the same split and bars as the yolo11n test (tests/_f16_parity.py): assignment bit-exact, criterion gradient on identical
    logits 1e-3, loss 2e-3, whole gradient 1e-2, every tensor 5 % + floor (fixed bars, CPU-trained fixed state)."""
    from tests._f16_parity import run_f16_parity
    r = run_f16_parity("yolo11s_fusion_sand3_new.yaml", R.resolve_graph("s", nc=2, graph=R.GRAPH_FUSION), nc=2, nb=8, sz=192)
    print("f16 parity fusion variant:", r)


def test_fusion_model_trainer_graph_replay_matches_eager():
    """The trainer surface (flat state, hipGraph replay, SGD + EMA) drives the fusion variant too: 4 steps with graph
    replay (after 2 eager warm-up steps) give the same losses and weights as 4 fully eager steps."""
    from sy11.engine.trainer import DetectionTrainer

    def mk(graphs):
        from sy11.nn.tasks import DetectionModel
        m = DetectionModel("yolo11s_fusion_sand3_new.yaml", ch=3, nc=2, verbose=False)
        m.load_state_dict(fusion_sd())
        return DetectionTrainer(m, batch_size=4, device=DEV, overrides={"amp": False, "nbs": 4}, graphs=graphs)

    def batch(seed):
        g = torch.Generator().manual_seed(seed)
        return {"img": torch.rand(4, 3, 64, 64, generator=g).to(DEV), "batch_idx": torch.tensor([0., 1., 3.]).to(DEV),
                "cls": torch.tensor([[1.], [0.], [1.]]).to(DEV),
                "bboxes": torch.tensor([[0.4, 0.4, 0.5, 0.4], [0.6, 0.65, 0.3, 0.5], [0.5, 0.5, 0.7, 0.6]]).to(DEV)}
    te, tg = mk(False), mk(True)
    le, lg = [], []
    for i in range(4):
        b = batch(i)
        le.append(te.train_step(dict(b))[0].item())
        lg.append(tg.train_step(dict(b))[0].item())
    assert len(tg.model.__dict__["_sy11_graph_cfg"]["entries"]) == 1
    for a, b_ in zip(le, lg):
        assert abs(a - b_) <= 5e-3 * abs(a), (le, lg)
    for (k, p), (_, q) in zip(te.model.state_dict().items(), tg.model.state_dict().items()):
        if p.dtype.is_floating_point:
            assert torch.allclose(p, q, rtol=1e-2, atol=1e-3), k
