"""Helpers with the reference's names (ultralytics/utils/torch_utils.py: fuse_conv_and_bn :238-265,
initialize_weights :410-420, init_seeds :474-492, ModelEMA :495-531, intersect_dicts, de_parallel)."""
from __future__ import annotations

import math
import random
from copy import deepcopy

import numpy as np
import torch
import torch.nn as nn


def fuse_conv_and_bn(conv: nn.Conv2d, bn: nn.BatchNorm2d) -> nn.Conv2d:
    """W' = diag(gamma / sqrt(var + eps)) W,  b' = beta - gamma * mean / sqrt(var + eps) (+ scaled conv bias)."""
    fused = nn.Conv2d(conv.in_channels, conv.out_channels, kernel_size=conv.kernel_size, stride=conv.stride,
                      padding=conv.padding, dilation=conv.dilation, groups=conv.groups, bias=True
                      ).requires_grad_(False).to(conv.weight.device)
    scale = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
    w = conv.weight.detach() * scale.view(-1, 1, 1, 1)
    fused.weight.data = w.contiguous(memory_format=torch.channels_last)
    b_conv = torch.zeros(conv.out_channels, device=conv.weight.device) if conv.bias is None else conv.bias.detach()
    fused.bias.copy_(b_conv * scale + bn.bias.detach() - bn.running_mean * scale)
    return fused


def initialize_weights(model):
    """BN eps = 1e-3, momentum = 0.03; activations in place (torch_utils.py:410-420)."""
    for m in model.modules():
        t = type(m)
        if t is nn.BatchNorm2d:
            m.eps = 1e-3
            m.momentum = 0.03
        elif t in {nn.Hardswish, nn.LeakyReLU, nn.ReLU, nn.ReLU6, nn.SiLU}:
            m.inplace = True


def intersect_dicts(da, db, exclude=()):
    """Entries of ``da`` that ``db`` also has under the same key with the same shape, minus keys containing an ``exclude`` fragment."""
    common = {}
    for key, value in da.items():
        twin = db.get(key)
        if twin is None or twin.shape != value.shape or any(frag in key for frag in exclude):
            continue
        common[key] = value
    return common


def de_parallel(model):
    """The wrapped module of a DataParallel / DDP container, else the model itself."""
    inner = getattr(model, "module", None)
    return inner if isinstance(inner, nn.Module) else model


def init_seeds(seed=0, deterministic=False):
    """Seed every generator (torch_utils.py:474-492).  ``deterministic`` (cfg/default.yaml:29) switches libsy11 to ordered
    reductions and pins the tile choice: see ``set_deterministic``."""
    for seeder in (random.seed, np.random.seed, torch.manual_seed):
        seeder(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    set_deterministic(bool(deterministic))


def set_deterministic(on: bool = True):
    """The reference's `deterministic: True` (torch.use_deterministic_algorithms, torch_utils.py:483-489) for the HIP path: every sum
    over workgroups (BatchNorm statistics and backward sums, filter gradients, loss partials, bias gradients) is taken in a fixed
    order (libsy11 option "deterministic", csrc/det.h), and the tile autotuner stops measuring (the heuristic or an imported pick
    table decides) — a different tile is a different summation order.  Two runs on the same inputs are then bit-identical; the
    price is the fold launches and slower filter gradients.  Turning it off restores the measuring tuner only if SY11_TUNE allows."""
    import os

    from .. import _lib
    _lib.set_option("deterministic", 1 if on else 0)
    if on:
        _lib.set_option("tune", 0)
    elif os.environ.get("SY11_TUNE", "1") != "0":
        _lib.set_option("tune", 1)


class ModelEMA:
    """Exponential moving average of every float entry of the state_dict, decay 0.9999 * (1 - exp(-updates / tau))."""

    def __init__(self, model, decay=0.9999, tau=2000, updates=0):
        self.ema = deepcopy(de_parallel(model)).eval()
        self.updates = updates
        self.decay = lambda x: decay * (1 - math.exp(-x / tau))
        for p in self.ema.parameters():
            p.requires_grad_(False)
        self.enabled = True

    def update(self, model):
        if not self.enabled:
            return
        self.updates += 1
        d = self.decay(self.updates)
        msd = de_parallel(model).state_dict()
        ek, ev, mv = [], [], []
        for k, v in self.ema.state_dict().items():
            if v.dtype.is_floating_point:
                ev.append(v)
                mv.append(msd[k].detach())
        torch._foreach_mul_(ev, d)
        torch._foreach_add_(ev, mv, alpha=1 - d)

    def update_attr(self, model, include=(), exclude=("process_group", "reducer")):
        """Mirror the model's public plain attributes (names, nc, args, ...) onto the averaged copy."""
        if not self.enabled:
            return
        for name, value in vars(model).items():
            private = name.startswith("_")
            if private or name in exclude or (include and name not in include):
                continue
            setattr(self.ema, name, value)
