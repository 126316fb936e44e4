"""Graph builder and model classes with the reference's surface (ultralytics/nn/tasks.py: BaseModel :122-327,
DetectionModel :329-418, parse_model :963-1168, yaml_model_load :1171-1184, guess_model_scale :1187-1203).

The whole layer graph executes inside ONE engine pass (``EngineFn``): activations stay NHWC on the MI355X, the
``y[m.f]`` skip bookkeeping of ``_predict_once`` is done on engine activations, and backward is the engine's tape.
"""
from __future__ import annotations

import ast
import os
import math
import re
from copy import deepcopy
from pathlib import Path

import torch
import torch.nn as nn
import yaml

from .. import ops
from ..engine import Act, Ctx, run_module
from ..utils.torch_utils import fuse_conv_and_bn, initialize_weights, intersect_dicts
from .modules import (C2PSA, C3, SPPF, Bottleneck, C2f, C3k, C3k2, Concat, Conv, DDWConv, Detect, DWConv, Fusion)

CFG_DIR = Path(__file__).resolve().parents[1] / "cfg" / "models"


def make_divisible(x, divisor):
    if isinstance(divisor, torch.Tensor):
        divisor = int(divisor.max())
    return math.ceil(x / divisor) * divisor


class BaseModel(nn.Module):
    """Base class: dict input -> loss, tensor input -> predictions (tasks.py:125-141)."""

    def forward(self, x, *args, **kwargs):
        if isinstance(x, dict):
            return self.loss(x, *args, **kwargs)
        return self.predict(x, *args, **kwargs)

    def predict(self, x, profile=False, visualize=False, augment=False, embed=None):
        if augment or visualize or embed or profile:
            raise NotImplementedError("augment / visualize / embed / profile are outside the hot path")
        return self._predict_once(x)

    def _predict_once(self, x, profile=False, visualize=False, embed=None):
        """One engine pass over the whole graph."""
        outs = run_module(self, x)
        det = self.model[-1]
        if isinstance(det, Detect):
            if self.training:
                return list(outs)
            return outs[0], list(outs[1:])
        return outs[0]

    def _concat_plan(self):
        """layer index -> (concat layer index, channel offset): producers write straight into the concat buffer of their
        (single) consuming Concat, so the model-level torch.cat (conv.py:1821) costs no copy and no extra backward pass."""
        plan = self.__dict__.get("_sy11_cat_plan")
        if plan is not None:
            return plan
        plan, widths = {}, {}
        layers = list(self.model)
        for m in layers:
            if isinstance(m, Concat) and m.d == 1 and not isinstance(m.f, int):
                srcs = [(m.i - 1 if j == -1 else j) for j in m.f]
                chans = [getattr(layers[s], "c_out", None) for s in srcs]
                ok = all(c is not None for c in chans) and len(set(srcs)) == len(srcs) and not any(s in plan for s in srcs) \
                    and all(isinstance(layers[s], (Conv, C2f, C3, SPPF, C2PSA, nn.Upsample)) for s in srcs)
                if ok:
                    off = 0
                    for s_, c in zip(srcs, chans):
                        plan[s_] = (m.i, off)
                        off += c
                    widths[m.i] = off
        self.__dict__["_sy11_cat_plan"] = plan
        self.__dict__["_sy11_cat_width"] = widths
        return plan

    def _run(self, ec: Ctx, x: Act):
        """tasks.py:174-188 on engine activations; Upsample / Concat run as 'write into the concat buffer'."""
        plan = self._concat_plan()
        widths = self.__dict__["_sy11_cat_width"]
        cats = {}
        y = []
        split = self.__dict__.get("_sy11_bucket_layer")     # data parallel: gradients of layers >= split form the first bucket
        for m in self.model:
            if m.i == split and ec.record:
                ec.marks["bucket"] = len(ec.tape)
            if m.f != -1:
                x = y[m.f] if isinstance(m.f, int) else [x if j == -1 else y[j] for j in m.f]
            out = None
            if m.i in plan:
                j, off = plan[m.i]
                B, H, W, _ = x.shape
                if isinstance(m, nn.Upsample):
                    H, W = 2 * H, 2 * W
                elif isinstance(m, Conv):
                    cv = m.conv
                    H, W = ops.conv_out_hw(H, W, cv.kernel_size[0], cv.stride[0], cv.padding[0], cv.dilation[0])
                if j not in cats:
                    cats[j] = Act(ec.empty(B, H, W, widths[j]))
                out = cats[j].slice(off, off + m.c_out)
            if isinstance(m, nn.Upsample):
                x = _upsample_run(ec, m, x, out)
            elif isinstance(m, Concat) and m.i in cats:
                x = cats[m.i]                       # every source already lives in this buffer
            elif out is not None:
                x = m._run(ec, x, out=out)
            else:
                x = m._run(ec, x)
            y.append(x if m.i in self.save else None)
        return x

    def fuse(self, verbose=True):
        """Fold every BatchNorm into its conv (tasks.py:223-251) -> single conv+bias+SiLU kernels."""
        if not self.is_fused():
            for m in self.model.modules():
                if isinstance(m, Conv) and hasattr(m, "bn"):
                    m.conv = fuse_conv_and_bn(m.conv, m.bn)
                    delattr(m, "bn")
                    m.forward = m.forward_fuse
        return self

    def is_fused(self, thresh=10):
        bn = tuple(v for k, v in nn.__dict__.items() if "Norm" in k)
        return sum(isinstance(v, bn) for v in self.modules()) < thresh

    def info(self, detailed=False, verbose=True, imgsz=640):
        n_p = sum(x.numel() for x in self.parameters())
        n_g = sum(x.numel() for x in self.parameters() if x.requires_grad)
        n_l = len(list(self.modules()))
        return n_l, n_p, n_g

    def _apply(self, fn):
        self = super()._apply(fn)
        m = self.model[-1]
        if isinstance(m, Detect):
            m.stride = fn(m.stride)
            m.anchors = fn(m.anchors)
            m.strides = fn(m.strides)
        return self

    def load(self, weights, verbose=True):
        model = weights["model"] if isinstance(weights, dict) else weights
        csd = model.float().state_dict()
        csd = intersect_dicts(csd, self.state_dict())
        self.load_state_dict(csd, strict=False)

    def loss(self, batch, preds=None):
        if getattr(self, "criterion", None) is None:
            self.criterion = self.init_criterion()
        preds = self.forward(batch["img"]) if preds is None else preds
        if self.training and os.environ.get("SY11_LOSS_INPLACE", "1") != "0":       # under graph replay the criterion writes d(loss)/d(maps) straight into the backward graph's inputs
            from ..engine import graph_static_gout
            self.criterion.grad_out = graph_static_gout(self)
        return self.criterion(preds, batch)

    def init_criterion(self):
        raise NotImplementedError("compute_loss() needs to be implemented by task heads")


def _upsample_run(ec: Ctx, m: nn.Upsample, x: Act, out: Act = None) -> Act:
    if m.mode != "nearest" or float(m.scale_factor) != 2.0:
        raise ops._lib.Sy11Error("only nn.Upsample(None, 2, 'nearest') has a HIP kernel")
    B, H, W, Cn = x.shape
    if out is None:
        out = Act(ec.empty(B, 2 * H, 2 * W, Cn))
    ops.upsample2x_fwd(x.data, out.data)
    if ec.record:
        def bw():
            if x.req:
                g, acc = x.grad_for_write()
                ops.upsample2x_bwd(out.grad_read(), g, accumulate=acc)
        ec.tape.append(bw)
    return out


class DetectionModel(BaseModel):
    """YOLO detection model."""

    def __init__(self, cfg="yolo11n.yaml", ch=3, nc=None, verbose=True):
        super().__init__()
        self.yaml = cfg if isinstance(cfg, dict) else yaml_model_load(cfg)
        ch = self.yaml["ch"] = self.yaml.get("ch", ch)
        if nc and nc != self.yaml["nc"]:
            self.yaml["nc"] = nc
        self.model, self.save = parse_model(deepcopy(self.yaml), ch=ch, verbose=verbose)
        self.names = {i: f"{i}" for i in range(self.yaml["nc"])}
        self.inplace = self.yaml.get("inplace", True)
        self.end2end = getattr(self.model[-1], "end2end", False)
        m = self.model[-1]
        if isinstance(m, Detect):
            m.inplace = self.inplace
            # The reference discovers the strides with a train-mode dry run on zeros(1, ch, 256, 256) (tasks.py:359-367).
            # There is no CPU execution path here, so they are derived from the graph (product of conv strides /
            # upsample factors along each Detect input) — same values.  The dry run's side effect is reproduced in closed
            # form: on a zero image every bias-free conv yields zeros, every BatchNorm (beta = 0 at construction) sees batch
            # mean 0 / variance 0 and maps to zeros again, so each of them takes ONE running-statistics update with the
            # constructor's momentum 0.1: running_mean 0, running_var 0.9 * 1 + 0.1 * 0, num_batches_tracked 1 — checked
            # against the reference's fresh models (yolo11n, the fusion variant: all 81 / 88 BatchNorms, r02).
            m.stride = torch.tensor(_graph_strides(self.model, self.save))
            self.stride = m.stride
            m.bias_init()
            for mod in self.modules():
                if type(mod) is nn.BatchNorm2d and mod.track_running_stats:
                    mod.running_mean.zero_()
                    mod.running_var.fill_(0.9)
                    mod.num_batches_tracked.fill_(1)
        else:
            self.stride = torch.Tensor([32])
        initialize_weights(self)

    def init_criterion(self):
        from ..utils.loss import v8DetectionLoss
        return v8DetectionLoss(self)


def _graph_strides(model, save):
    """Down-sampling factor of every Detect input, from the layer graph."""
    s = []
    for m in model:
        prev = s[-1] if s else 1.0
        f = m.f
        src = prev if f == -1 else (s[f] if isinstance(f, int) else None)
        if isinstance(m, Detect):
            return [float(s[j]) for j in f]
        if isinstance(m, Conv):
            s.append(src * m.conv.stride[0])
        elif isinstance(m, DDWConv):
            s.append(src * m.conv1.conv.stride[0])
        elif isinstance(m, Fusion):
            vals = [prev if j == -1 else s[j] for j in f]
            assert len(set(vals)) == 1, "Fusion inputs have different strides"
            s.append(vals[0])
        elif isinstance(m, nn.Upsample):
            s.append(src / float(m.scale_factor))
        elif isinstance(m, Concat):
            vals = [prev if j == -1 else s[j] for j in f]
            assert len(set(vals)) == 1, "Concat inputs have different strides"
            s.append(vals[0])
        else:
            s.append(src)
    raise ValueError("no Detect layer")


_MODULES = {"Conv": Conv, "DWConv": DWConv, "Bottleneck": Bottleneck, "C2f": C2f, "C3": C3, "C3k": C3k, "C3k2": C3k2,
            "SPPF": SPPF, "C2PSA": C2PSA, "Concat": Concat, "Detect": Detect, "DDWConv": DDWConv, "Fusion": Fusion}
_WIDTH_SCALED = {Conv, DWConv, DDWConv, Bottleneck, C2f, C3, C3k, C3k2, SPPF, C2PSA}       # (c1, c2, ...) constructors: c2 follows the width multiple
_TAKES_REPEATS = {C2f, C3, C3k, C3k2, C2PSA}                                               # the repeat count is a constructor argument


def _model_scale(d):
    """(depth multiple, width multiple, channel cap, scale letter) of a model dict (tasks.py:968-981)."""
    scales, letter = d.get("scales"), d.get("scale")
    if scales:
        letter = letter or next(iter(scales))
        depth, width, cap = scales[letter]
        return depth, width, cap, letter
    return d.get("depth_multiple", 1.0), d.get("width_multiple", 1.0), float("inf"), letter


def _yaml_value(token, names):
    """A YAML argument: a name the parser knows (`nc`), a Python literal ('None', '[1, 2]'), or the string itself ('nearest')."""
    if not isinstance(token, str):
        return token
    if token in names:
        return names[token]
    try:
        return ast.literal_eval(token)
    except (ValueError, SyntaxError):
        return token


def _module_class(name):
    if not isinstance(name, str):
        return name
    if name.startswith("nn."):
        return getattr(torch.nn, name[3:])
    if name not in _MODULES:
        raise ops._lib.Sy11Error(f"module '{name}' is outside the MI355X hot path (SURVEY.md §2.1): no HIP kernel")
    return _MODULES[name]


def parse_model(d, ch, verbose=True):
    """Model dict (YAML) -> (nn.Sequential of layers, sorted list of layer indices whose outputs later layers read).
    One pass over `backbone + head` rows [from, repeats, module, args]; the channel / depth arithmetic of tasks.py:1085-1141:
    width-scaled modules get (c_in, round-up-to-8(min(c2, cap) * width), ...), repeat counts scale with depth, Concat sums its
    inputs, Fusion keeps its first input's width, Detect receives the list of its input widths.  Every layer carries
    .i / .f / .type / .np (index, sources, class path, parameter count) and .c_out (for the concat planner)."""
    depth, width, cap, letter = _model_scale(d)
    nc = d.get("nc")
    if d.get("activation"):
        Conv.default_act = eval(d["activation"])  # noqa: S307 — same contract as the reference's YAML `activation:` key
    known = {"nc": nc}
    out_ch = []                                    # output width of every layer built so far
    width_of = lambda j: out_ch[j] if out_ch else ch          # noqa: E731  (layer 0 reads the image)
    layers, reads, legacy = [], set(), True
    c_out = ch
    for i, (src, repeats, name, raw_args) in enumerate(d["backbone"] + d["head"]):
        cls = _module_class(name)
        args = [_yaml_value(a, known) for a in raw_args]
        n = max(round(repeats * depth), 1) if repeats > 1 else repeats
        if cls in _WIDTH_SCALED:
            c_out = args[0]
            if c_out != nc:
                c_out = make_divisible(min(c_out, cap) * width, 8)
            args = [width_of(src), c_out, *args[1:]]
            if cls in _TAKES_REPEATS:
                args.insert(2, n)
                n = 1
            if cls is C3k2:                        # the YOLO11 family: non-legacy Detect head; m / l / x always use C3k inside
                legacy = False
                if letter in "mlx":
                    args[3] = True
        elif cls is Concat:
            c_out = sum(width_of(j) for j in src)
        elif cls is Fusion:                        # tasks.py:1132-1135: the parser overrides the YAML's fusion type
            args = [[width_of(j) for j in src], "ESChannel"]
            c_out = width_of(src[0])
        elif cls is Detect:
            args.append([width_of(j) for j in src])
            cls.legacy = legacy
        else:                                      # nn.Upsample & co: width passes through
            c_out = width_of(src)
        layer = cls(*args) if n == 1 else nn.Sequential(*[cls(*args) for _ in range(n)])
        layer.i, layer.f, layer.type = i, src, f"{cls.__module__}.{cls.__qualname__}"
        layer.np = sum(p.numel() for p in layer.parameters())
        layer.c_out = c_out
        reads.update(j % i for j in ([src] if isinstance(src, int) else src) if j != -1)
        layers.append(layer)
        out_ch.append(c_out)
    return nn.Sequential(*layers), sorted(reads)


def guess_model_scale(model_path):
    try:
        return re.search(r"yolo[v]?\d+([nslmx])", Path(model_path).stem).group(1)
    except AttributeError:
        return ""


def yaml_model_load(path):
    """'yolo11s.yaml' -> dict of cfg/models/11/yolo11.yaml with scale 's' (tasks.py:1171-1184)."""
    path = Path(path)
    unified = re.sub(r"(\d+)([nslmx])(.+)?$", r"\1\3", str(path.name))
    for cand in (path, Path(unified)):
        for base in (Path("."), CFG_DIR / "11", CFG_DIR):
            f = base / cand
            if f.is_file():
                with open(f) as fh:
                    d = yaml.safe_load(fh)
                d["scale"] = guess_model_scale(path)
                d["yaml_file"] = str(path)
                return d
    raise FileNotFoundError(f"model yaml '{path}' not found (searched {CFG_DIR})")
