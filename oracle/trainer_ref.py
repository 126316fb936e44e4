"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product package).

CPU restatement of the reference trainer's update rule, i.e. what one `optimizer_step` does to the weights:
  engine/trainer.py:758-819  build_optimizer: three parameter groups by name / type — any parameter whose full name contains
                             "bias" (no decay), weights of nn.*Norm* modules (no decay), the rest (weight decay); `auto` picks
                             SGD(0.01, 0.9, nesterov) for > 10 000 iterations, else AdamW(lr = round(0.002 * 5 / (4 + nc), 6),
                             betas (0.9, 0.999)) and sets warmup_bias_lr = 0;
  engine/trainer.py:585-593  optimizer_step: unscale -> clip_grad_norm_(10.0) over ALL parameters -> step -> zero_grad -> EMA;
  utils/torch_utils.py:495-531 ModelEMA.update: every floating-point state_dict entry (BN running statistics included),
                             v = d * v + (1 - d) * model, d = 0.9999 * (1 - exp(-updates / 2000)).
torch.optim's SGD / AdamW arithmetic (third party, `torch` — installed) is written out explicitly below.

Parity status: PINNED — tests/golden/trainer.npz holds the weights / EMA / optimizer state the reference's own
`BaseTrainer.build_optimizer`, `BaseTrainer.optimizer_step` and `ModelEMA` produce (oracle/gen_golden_trainer.py), and
tests/test_trainer_oracle_cpu.py checks this restatement against it.
"""
from __future__ import annotations

import math

import torch

from .yolo11_ref import closed_form

GRAD_AMPS = (1.0, 0.01, 0.2)          # total gradient norm ~ 500 / 5 / 100 on the tiny model: clipped, not clipped, clipped
EMA_START_UPDATES = 3000               # decay ~ 0.78 instead of ~ 5e-4: the running average visibly lags the model


def synthetic_grad(name: str, shape, step: int) -> torch.Tensor:
    """The gradient every test / generator assigns to parameter ``name`` at optimizer step ``step`` (closed form)."""
    return closed_form(f"tg{step}.{name}", tuple(shape), "signed") * GRAD_AMPS[step % len(GRAD_AMPS)]


def perturb_buffers(state_dict: dict, step: int):
    """What a train-mode forward would do to the BatchNorm buffers, as a closed form (in place)."""
    with torch.no_grad():
        for k, v in state_dict.items():
            if k.endswith("running_mean"):
                v.add_(0.05 * closed_form(f"tb{step}.{k}", tuple(v.shape), "signed").to(v.device))
            elif k.endswith("running_var"):
                v.mul_(1.0 + 0.1 * closed_form(f"tb{step}.{k}", tuple(v.shape), "signed").abs().to(v.device))
            elif k.endswith("num_batches_tracked"):
                v.add_(1)


def param_groups(named_parameters, norm_weight_names):
    """-> (decay, norm, bias) name lists in the reference's iteration order (trainer.py:789-797)."""
    g = [], [], []
    for name, _ in named_parameters:
        if "bias" in name:
            g[2].append(name)
        elif name in norm_weight_names:
            g[1].append(name)
        else:
            g[0].append(name)
    return g


def auto_optimizer(nc: int, iterations: float):
    """trainer.py:778-786 -> (name, lr, momentum)."""
    return ("SGD", 0.01, 0.9) if iterations > 10000 else ("AdamW", round(0.002 * 5 / (4 + nc), 6), 0.9)


class RefTrainerState:
    """Weights + optimizer state + EMA of one model, updated by the written-out rule."""

    def __init__(self, state_dict: dict, trainable: list, norm_weight_names: set, name="SGD", lr=0.01, momentum=0.937, decay=5e-4,
                 ema_updates=0):
        self.sd = {k: v.clone() for k, v in state_dict.items()}
        self.trainable = list(trainable)
        decay_g, norm_g, bias_g = param_groups([(k, None) for k in self.trainable], norm_weight_names)
        self.wd = {**{k: decay for k in decay_g}, **{k: 0.0 for k in norm_g}, **{k: 0.0 for k in bias_g}}
        self.name, self.lr, self.momentum = name, lr, momentum
        self.buf = {}                                   # SGD momentum buffers / Adam moments
        self.t = 0
        self.ema = {k: v.clone() for k, v in state_dict.items()}
        self.ema_updates = ema_updates

    def optimizer_step(self, grads: dict, lr=None):
        lr = self.lr if lr is None else lr
        total = math.sqrt(sum(float(g.double().pow(2).sum()) for g in grads.values()))
        coef = min(10.0 / (total + 1e-6), 1.0)           # clip_grad_norm_(max_norm=10): ONE factor for all parameters
        self.t += 1
        for k in self.trainable:
            if k not in grads:
                continue
            g = grads[k] * coef
            p = self.sd[k]
            if self.name == "SGD":                       # torch.optim.SGD, nesterov, dampening 0
                if self.wd[k]:
                    g = g + self.wd[k] * p
                b = self.buf.get(k)
                b = g.clone() if b is None else b * self.momentum + g
                self.buf[k] = b
                p.sub_(lr * (g + self.momentum * b))
            else:                                        # torch.optim.AdamW: decoupled decay, bias-corrected moments
                p.mul_(1.0 - lr * self.wd[k])
                m, v = self.buf.get(k, (torch.zeros_like(p), torch.zeros_like(p)))
                m = m * self.momentum + (1 - self.momentum) * g
                v = v * 0.999 + (1 - 0.999) * g * g
                self.buf[k] = (m, v)
                bc1, bc2 = 1 - self.momentum ** self.t, 1 - 0.999 ** self.t
                p.sub_((lr / bc1) * m / ((v / bc2).sqrt() + 1e-8))
        self.ema_updates += 1                             # ModelEMA.update
        d = 0.9999 * (1 - math.exp(-self.ema_updates / 2000))
        for k, v in self.ema.items():
            if v.dtype.is_floating_point:
                v.mul_(d).add_((1 - d) * self.sd[k])
