"""Shared body: the product trainer's update rule (sy11.engine.trainer.DetectionTrainer.optimizer_step on flat buffers +
FlatEMA) against tests/golden/trainer.npz = the reference's own BaseTrainer.build_optimizer / optimizer_step / ModelEMA run
on the same tiny model, gradients and buffer perturbations (oracle/gen_golden_trainer.py).  Device-agnostic host logic +
torch optimizers: runs on CPU (tests/test_trainer_oracle_cpu.py) and on the MI355X with the fused kernels (test_trainer_gpu.py)."""
import numpy as np
import torch
import yaml

from oracle import trainer_ref as T, yolo11_ref as R
from tests._golden import check, load


def tiny_sd(nc=4):
    return R.seeded_state_dict(R.empty_state_dict(R.resolve_graph("t", nc=nc)), seed=0)


def product_trainer(tag, device, flat=True):
    from sy11.engine.trainer import DetectionTrainer
    from sy11.nn.tasks import CFG_DIR, DetectionModel
    d = yaml.safe_load(open(CFG_DIR / "11" / "yolo11.yaml"))
    d["scales"]["t"] = [0.5, 0.125, 1024]
    d["scale"] = "t"
    m = DetectionModel(d, ch=3, nc=4, verbose=False)
    m.load_state_dict(tiny_sd())
    ov = {"amp": False, "lr0": 0.01, "momentum": 0.937, "weight_decay": 5e-4, "nbs": 64}
    ov.update({"optimizer": "SGD"} if tag == "sgd" else {"optimizer": "auto", "iterations": 300})
    tr = DetectionTrainer(m, batch_size=64, device=device, overrides=ov, graphs=False, flat=flat)
    tr.ema.updates = T.EMA_START_UPDATES
    return tr


def run_three_steps(tr):
    params = [(k, p) for k, p in tr.model.named_parameters() if p.requires_grad]
    for step in range(3):
        for k, p in params:
            g = T.synthetic_grad(k, p.shape, step).to(p.device)
            if tr.flat is not None:
                tr.grad_store.views[id(p)].copy_(g)            # what the kernels do: write into the flat gradient buffer's views
            else:
                p.grad = g
        T.perturb_buffers(tr.model.state_dict(), step)
        tr.optimizer_step()
    return tr


def compare_with_reference(tr, tag, rtol=1e-5):
    gold = load("trainer.npz")
    assert type(tr.optimizer).__name__ == str(gold[f"{tag}.optimizer"])
    assert abs(tr.args.warmup_bias_lr - float(gold[f"{tag}.warmup_bias_lr"])) < 1e-12
    assert tr.ema.updates == int(gold[f"{tag}.ema_updates"])
    # group hyper-parameters: the reference's groups are [bias, decay, norm]
    for gi, g in enumerate(tr.optimizer.param_groups):
        lr, mom, wd = gold[f"{tag}.group{gi}.hyper"]
        assert abs(g["lr"] - lr) < 1e-12 and abs(g["weight_decay"] - wd) < 1e-12
        assert abs(g.get("momentum", g.get("betas", (0, 0))[0]) - mom) < 1e-12
    for which, sd in (("model", tr.model.state_dict()), ("ema", tr.ema.ema.state_dict())):
        names = [str(n) for n in gold[f"{tag}.{which}.names"]]
        assert names == [k for k, v in sd.items() if v.dtype.is_floating_point]
        for k, (nrm, sm) in zip(names, gold[f"{tag}.{which}.norm_sum"]):
            v = sd[k].double()
            assert abs(v.norm().item() - nrm) <= rtol * max(nrm, 1e-6), (which, k, v.norm().item(), nrm)
            assert abs(v.sum().item() - sm) <= rtol * 10 * max(float(v.abs().sum()), 1e-6), (which, k)
        for k in ("model.0.conv.weight", "model.0.bn.weight", "model.0.bn.bias", "model.0.bn.running_mean", "model.0.bn.running_var",
                  "model.10.m.0.attn.qkv.conv.weight", "model.22.cv2.bn.weight", "model.23.cv2.0.2.bias", "model.23.cv3.2.2.weight",
                  "model.23.cv3.1.2.bias"):
            check(gold, f"{tag}.{which}.{k}", sd[k], rtol=rtol, atol=1e-7, what=which + " ")
    # optimizer state (momentum buffer / first Adam moment), per parameter
    key = str(gold[f"{tag}.opt.key"])
    want = dict(zip([str(n) for n in gold[f"{tag}.opt.names"]], gold[f"{tag}.opt.norm_sum"]))
    named = dict(tr.model.named_parameters())
    got = {}
    if tr.flat is not None:
        from sy11.engine.flat import _view_like
        for gi, (a, b) in enumerate(tr.flat.group_slices):
            flat_p = tr.flat_params[gi]
            st = tr.optimizer.state[flat_p][key]
            for k, p in named.items():
                off = tr.flat.offsets.get(id(p))
                if off is not None and a <= off < b:
                    got[k] = _view_like(st[off - a:off - a + p.numel()], p)
    else:
        got = {k: tr.optimizer.state[p][key] for k, p in named.items() if p in tr.optimizer.state}
    assert set(got) == set(want)
    for k, (nrm, sm) in want.items():
        v = got[k].double()
        assert abs(v.norm().item() - nrm) <= rtol * max(nrm, 1e-6), ("opt", k, v.norm().item(), nrm)
        if f"{tag}.opt.{k}.shape" in gold:
            check(gold, f"{tag}.opt.{k}", got[k], rtol=rtol, atol=1e-7, what="opt ")
