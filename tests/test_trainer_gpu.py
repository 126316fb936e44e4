"""GPU: the trainer step (AMP off/on, optimizer, EMA) and hipGraph replay vs eager execution."""
from types import SimpleNamespace

import pytest
import torch

from oracle import yolo11_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda"


def make_trainer(graphs, amp=False, seed=1):
    from sy11.engine.trainer import DetectionTrainer
    from sy11.nn.tasks import DetectionModel
    m = DetectionModel("yolo11n.yaml", nc=80, verbose=False)
    m.load_state_dict(R.seeded_state_dict(R.empty_state_dict(R.resolve_graph("n", nc=80)), seed=seed))
    return DetectionTrainer(m, batch_size=4, device=DEV, overrides={"amp": amp, "nbs": 4}, graphs=graphs)


def batch(seed=0, n=4):
    g = torch.Generator().manual_seed(seed)
    return {"img": torch.rand(n, 3, 128, 128, generator=g).to(DEV),
            "batch_idx": torch.tensor([0., 1., 2., 3.]).to(DEV), "cls": torch.tensor([[1.], [5.], [9.], [30.]]).to(DEV),
            "bboxes": (0.3 + 0.3 * torch.rand(4, 4, generator=g)).to(DEV)}


def test_graph_replay_matches_eager_fp32():
    """5 optimizer steps with hipGraph replay (after 2 eager warm-up steps) vs 5 fully eager steps: same losses, same weights."""
    te, tg = make_trainer(False), make_trainer(True)
    le, lg = [], []
    for i in range(5):
        b = batch(i)
        le.append(te.train_step(dict(b))[0].item())
        lg.append(tg.train_step(dict(b))[0].item())
    assert "_sy11_graph_cfg" in tg.model.__dict__ and len(tg.model.__dict__["_sy11_graph_cfg"]["entries"]) == 1
    for a, b_ in zip(le, lg):
        assert abs(a - b_) <= 2e-3 * abs(a), (le, lg)
    for (k, p), (_, q) in zip(te.model.state_dict().items(), tg.model.state_dict().items()):
        if p.dtype.is_floating_point:
            assert torch.allclose(p, q, rtol=5e-3, atol=5e-4), k
    assert te.ema.updates == 5 and tg.ema.updates == 5


def test_train_step_decreases_loss_amp():
    """fp16 AMP + GradScaler + SGD on a fixed batch.  The seeded random model has a loss of ~3e4, so the scaler first
    halves its scale (65536 -> ...) on overflowing steps — exactly GradScaler's contract — then the loss goes down."""
    t = make_trainer(True, amp=True)
    b = batch(3)
    losses = [t.train_step(dict(b))[0].item() for _ in range(24)]
    assert all(l == l for l in losses)
    assert 0 < t.scaler.get_scale() < 65536
    assert min(losses[-6:]) < 0.9 * losses[0], losses


def test_accumulation_window_and_grad_views():
    from sy11.engine.trainer import DetectionTrainer
    from sy11.nn.tasks import DetectionModel
    m = DetectionModel("yolo11n.yaml", nc=80, verbose=False)
    t = DetectionTrainer(m, batch_size=4, device=DEV, overrides={"amp": False, "nbs": 8}, graphs=False)
    assert t.flat is not None
    assert t.accumulate == 2
    b = batch(1)
    t.train_step(dict(b))
    gs = t.model.__dict__["_sy11_grads"]
    g1 = gs.flat.clone()
    assert g1.abs().sum() > 0 and t.last_opt_step == -1           # no optimizer step yet: gradients kept
    t.train_step(dict(b))
    assert t.last_opt_step == 1
    assert gs.flat.abs().sum() == 0                                # gradients cleared after the optimizer step


@pytest.mark.parametrize("opt", [{}, {"optimizer": "auto", "iterations": 500}, {"optimizer": "Adam", "lr0": 1e-4, "weight_decay": 0.05}])
def test_flat_state_matches_per_tensor_optimizer(opt):
    """Flat-slice optimizer/EMA (3 tensors, fused kernels) and the reference-style per-tensor optimizer/ModelEMA give the same
    weights — for SGD-nesterov and for the AdamW that `optimizer=auto` picks on short runs (trainer.py:778-786)."""
    from sy11.engine.trainer import DetectionTrainer
    from sy11.nn.tasks import DetectionModel

    def mk(flat):
        m = DetectionModel("yolo11n.yaml", nc=80, verbose=False)
        m.load_state_dict(R.seeded_state_dict(R.empty_state_dict(R.resolve_graph("n", nc=80)), seed=4))
        t = DetectionTrainer(m, batch_size=4, device=DEV, overrides={"amp": False, "nbs": 4, **opt}, graphs=False, flat=flat)
        if opt.get("optimizer") == "auto":
            assert type(t.optimizer).__name__ == "AdamW" and abs(t.args.lr0 - round(0.002 * 5 / 84, 6)) < 1e-12
        if opt.get("optimizer") == "Adam":
            # torch.optim.Adam = L2 decay added to the gradient BEFORE the moments (the reference's build_optimizer, trainer.py:806-807);
            # the flat step's kind 2 — r03 routed it through the AdamW rule.  A large decay makes the two rules differ visibly.
            assert type(t.optimizer).__name__ == "Adam" and (not flat or t._opt_kind == 2)
        return t
    ta, tb = mk(True), mk(False)
    for i in range(3):
        b = batch(10 + i)
        ta.train_step(dict(b))
        tb.train_step(dict(b))
    sa, sb = ta.model.state_dict(), tb.model.state_dict()
    assert list(sa) == list(sb)
    # AdamW moves every element by ~lr per step whatever the gradient's size: elements whose exact gradient is ~0 (filters in
    # front of a BatchNorm are scale-invariant) get a sign decided by summation-order noise -> bound = steps * lr, not 2e-4
    atol = 2e-4 if not opt else 3 * ta.args.lr0 * 1.5
    for k in sa:
        if sa[k].dtype.is_floating_point:
            assert torch.allclose(sa[k], sb[k], rtol=2e-3, atol=atol), k
    ea, eb = ta.ema.ema.state_dict(), tb.ema.ema.state_dict()
    for k in ea:
        if ea[k].dtype.is_floating_point:
            assert torch.allclose(ea[k], eb[k], rtol=2e-3, atol=atol), k


def test_eager_half_eval_of_the_flat_ema_sees_updated_weights():
    """Round-1 defect: the f16 working-filter cache of an eager eval forward was keyed on the parameter's own version counter,
    which in-place updates of the FLAT buffer never bump — every later eager half eval of the EMA (short last batch, rect-val
    shapes beyond the forward-graph budget) ran with the weights of the first one.  Eval (half) -> train -> eval with a new
    batch shape must equal a forward of a fresh model loaded with the current EMA weights."""
    from sy11.nn.tasks import DetectionModel
    t = make_trainer(graphs=True, amp=True)
    ema = t.ema.ema
    ema._sy11_dtype = torch.float16
    x0 = torch.rand(2, 3, 96, 96, device=DEV)
    with torch.no_grad():
        y_before = ema.eval()(x0)[0].float().clone()          # fills every Conv's working-filter cache (eager: no graph cfg yet)
    for i in range(6):
        t.train_step(dict(batch(20 + i)))
    with torch.no_grad():
        t.ema.ema_state.flat.mul_(1.02)                         # make the change unmistakable (on top of the 6 EMA updates)
    x1 = torch.rand(3, 3, 64, 96, device=DEV)                  # a shape the EMA has never seen: eager path again
    with torch.no_grad():
        got = ema.eval()(x1)[0].float()
    fresh = DetectionModel("yolo11n.yaml", nc=80, verbose=False)
    fresh.load_state_dict({k: v.clone() for k, v in ema.state_dict().items()})
    fresh = fresh.to(DEV).eval()
    fresh._sy11_dtype = torch.float16
    with torch.no_grad():
        want = fresh(x1)[0].float()
        stale_probe = ema(x0)[0].float()
    assert torch.equal(got, want), (got - want).abs().max().item()
    assert not torch.allclose(stale_probe, y_before)           # and the first shape re-casts too


def test_resume_restores_best_fitness_and_closes_mosaic_late():
    """engine/trainer.py:727-756: a resumed run keeps the pre-resume best fitness (a worse first epoch must not replace
    best.pt) and closes mosaic at once when it restarts inside the last `close_mosaic` epochs."""
    t = make_trainer(graphs=False)
    start = t.resume_training({"epoch": 7, "best_fitness": 0.42, "optimizer": None, "ema": None, "updates": 3})
    assert start == 8 and abs(t.best_fitness - 0.42) < 1e-12

    class DS:
        hyp = None
        closed = 0

        def close_mosaic(self, hyp):
            self.closed += 1

    class Loader:
        dataset = DS()

        def __len__(self):
            return 1

        def __iter__(self):
            return iter([batch(5)])
    ld = Loader()
    hist = t.fit(ld, epochs=10, val_batches=None, save_dir=None, close_mosaic=4, start_epoch=8)      # 8 >= 10 - 4: already past
    assert ld.dataset.closed == 1 and len(hist) == 2
    assert abs(t.best_fitness - 0.42) < 1e-12                   # no validation ran: the restored value stands


@pytest.mark.parametrize("tag", ["sgd", "auto"])
def test_flat_trainer_update_rule_matches_the_reference_trainer(tag):
    """SURVEY §8(a) row 13 pinned to the reference: three optimizer steps (clip 10 -> fused SGD-nesterov / AdamW on the flat
    buffers -> FlatEMA) on the MI355X against tests/golden/trainer.npz = the reference's build_optimizer + optimizer_step +
    ModelEMA on the same model, gradients and BN-buffer updates (oracle/gen_golden_trainer.py): weights, EMA, momentum at 1e-5."""
    from tests._trainer_parity import compare_with_reference, product_trainer, run_three_steps
    tr = run_three_steps(product_trainer(tag, DEV, flat=True))
    torch.cuda.synchronize()
    compare_with_reference(tr, tag, rtol=2e-5)
