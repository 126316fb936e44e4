"""ORACLE — TEST INFRASTRUCTURE ONLY.  Golden generator for the trainer's update rule (dev container only).

Runs the REFERENCE's own `BaseTrainer.build_optimizer` (engine/trainer.py:758-819), `BaseTrainer.optimizer_step` (:585-593) and
`ModelEMA` (utils/torch_utils.py:495-531) on the tiny scale-`t` model for three optimizer steps with closed-form gradients
(oracle/trainer_ref.py: synthetic_grad / perturb_buffers), for explicit SGD-nesterov and for `optimizer=auto` on a short run
(-> AdamW), and stores the resulting weights, EMA and optimizer state in tests/golden/trainer.npz.
Run:  python -m oracle.gen_golden_trainer"""
from __future__ import annotations

import sys
from types import SimpleNamespace

import numpy as np
import torch

from .gen_golden import OUT, ROOT, import_reference, summarize

CASES = {"sgd": dict(name="SGD", lr=0.01, momentum=0.937, decay=5e-4, iterations=1e5),
         "auto": dict(name="auto", lr=0.01, momentum=0.937, decay=5e-4, iterations=300)}
SAMPLED = ("model.0.conv.weight", "model.0.bn.weight", "model.0.bn.bias", "model.0.bn.running_mean", "model.0.bn.running_var",
           "model.10.m.0.attn.qkv.conv.weight", "model.22.cv2.bn.weight", "model.23.cv2.0.2.bias", "model.23.cv3.2.2.weight",
           "model.23.cv3.1.2.bias")


def main():
    import_reference()
    sys.path.insert(0, str(ROOT))
    from oracle import trainer_ref as T
    from oracle.yolo11_ref import empty_state_dict, resolve_graph, seeded_state_dict
    from ultralytics.engine.trainer import BaseTrainer
    from ultralytics.nn.tasks import DetectionModel, yaml_model_load
    from ultralytics.utils.torch_utils import ModelEMA

    torch.set_num_threads(4)
    nc = 4
    d = yaml_model_load("yolo11n.yaml")
    d["scales"]["t"] = [0.5, 0.125, 1024]
    d["scale"] = "t"
    store = {}
    for tag, kw in CASES.items():
        model = DetectionModel(d, ch=3, nc=nc, verbose=False)
        model.load_state_dict(seeded_state_dict(empty_state_dict(resolve_graph("t", nc=nc)), seed=0))
        for k, v in model.named_parameters():                       # engine/trainer.py:246-252: always freeze the DFL conv
            if ".dfl" in k:
                v.requires_grad = False
        model.train()
        me = SimpleNamespace(args=SimpleNamespace(lr0=kw["lr"], momentum=kw["momentum"], warmup_bias_lr=0.1), data={"nc": nc}, model=model,
                             scaler=torch.amp.GradScaler("cpu", enabled=False))
        me.ema = ModelEMA(model, updates=T.EMA_START_UPDATES)
        me.optimizer = BaseTrainer.build_optimizer(me, model, **kw)
        store[f"{tag}.optimizer"] = np.asarray(type(me.optimizer).__name__)
        store[f"{tag}.warmup_bias_lr"] = np.asarray(me.args.warmup_bias_lr, dtype=np.float64)
        names = {id(p): k for k, p in model.named_parameters()}
        for gi, g in enumerate(me.optimizer.param_groups):
            store[f"{tag}.group{gi}.names"] = np.asarray([names[id(p)] for p in g["params"]])
            store[f"{tag}.group{gi}.hyper"] = np.asarray([g["lr"], g.get("momentum", g.get("betas", (0, 0))[0]), g["weight_decay"]], dtype=np.float64)
        for step in range(3):
            for k, p in model.named_parameters():
                if p.requires_grad:
                    p.grad = T.synthetic_grad(k, p.shape, step)
            T.perturb_buffers(model.state_dict(), step)
            BaseTrainer.optimizer_step(me)
        assert all(p.grad is None for p in model.parameters())
        store[f"{tag}.ema_updates"] = np.asarray(me.ema.updates)
        for which, sd in (("model", model.state_dict()), ("ema", me.ema.ema.state_dict())):
            keys = [k for k, v in sd.items() if v.dtype.is_floating_point]
            store[f"{tag}.{which}.names"] = np.asarray(keys)
            store[f"{tag}.{which}.norm_sum"] = np.asarray([[sd[k].double().norm().item(), sd[k].double().sum().item()] for k in keys], dtype=np.float64)
            for k in SAMPLED:
                summarize(store, f"{tag}.{which}.{k}", sd[k])
        state_key = "momentum_buffer" if type(me.optimizer).__name__ == "SGD" else "exp_avg"
        rows = []
        for k, p in model.named_parameters():
            st = me.optimizer.state.get(p)
            if st and state_key in st:
                rows.append((k, st[state_key].double().norm().item(), st[state_key].double().sum().item()))
                if k in SAMPLED:
                    summarize(store, f"{tag}.opt.{k}", st[state_key])
        store[f"{tag}.opt.names"] = np.asarray([r[0] for r in rows])
        store[f"{tag}.opt.norm_sum"] = np.asarray([[r[1], r[2]] for r in rows], dtype=np.float64)
        store[f"{tag}.opt.key"] = np.asarray(state_key)
    np.savez_compressed(OUT / "trainer.npz", **store)
    print("trainer.npz", (OUT / "trainer.npz").stat().st_size)


if __name__ == "__main__":
    main()
